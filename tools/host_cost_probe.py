"""Host cost of one operator call on the GPU box: bare ctypes call, ops.* wrapper, HIP launch, torch.empty."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "p2i-gan-benchmark_amd"))
import torch
from p2igan_bench import _hip, ops
lib = _hip.load()
dev = "cuda"
t = torch.zeros(1024, device=dev)
u = torch.zeros(1024, device=dev)
s = torch.cuda.current_stream().cuda_stream
N = 20000
def tm(name, fn):
    for _ in range(100): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(N): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    print("%-52s %.2f us/call" % (name, (t1 - t0) / N * 1e6), flush=True)
tp, up = t.data_ptr(), u.data_ptr()
tm("ctypes p2i_zero(n=0): no launch", lambda: lib.p2i_zero(tp, 0, s))
tm("ctypes p2i_axpy(n=1024): one launch", lambda: lib.p2i_axpy(tp, up, 1.0, 1024, s))
with ops.step_stream():
    tm("ops.axpy_ wrapper (pinned stream)", lambda: ops.axpy_(t, u, 1.0))
tm("ops.axpy_ wrapper", lambda: ops.axpy_(t, u, 1.0))
tm("torch.empty((8,64,128,128))", lambda: torch.empty((8, 64, 128, 128), device=dev))
tm("torch.cuda.current_stream().cuda_stream", lambda: torch.cuda.current_stream().cuda_stream)
spec = ops.ConvSpec(64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1))
x = torch.randn(1, 64, 32, 32, device=dev)
wp_f, wp_d = ops.weight_pack(torch.randn(64, 64, 9, device=dev) * 0.05)
N = 5000
with ops.step_stream():
    tm("ops.conv_fwd small (wrapper + plan + launch)", lambda: ops.conv_fwd(spec, x, wp_f))
    tm("ops.conv_wgrad small", lambda: ops.conv_wgrad(spec, x, x))
