"""Round-4 experiment: the 32-channel-tile split-pipe kernels fed from PRE-SPLIT bf16 source planes instead of
splitting every source element in the kernel once per channel tile.  Per layer: the result against the in-kernel split (must be
bit-equal: the same three planes reach the same MFMAs), the launch time of both forms, and the cost of making the planes with a
standalone pass (an epilogue that writes them would replace it).
usage: python tools/presplit_probe.py [B=8] [iters=20]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "p2i-gan-benchmark_amd"))
import torch
from p2igan_bench import _hip, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
lib = _hip.load()
S2 = ops.ConvSpec
LAYERS = [("G l2 3x3 256 @32", S2(256, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1)), (256, 1, 32, 32)),
          ("G l3 3x3 512 @16", S2(512, 512, (1, 3, 3), (1, 1, 1), (0, 1, 1)), (512, 1, 16, 16)),
          ("D3 128->128 s211", S2(128, 128, (3, 3, 3), (2, 1, 1), (1, 1, 1)), (128, 16, 16, 16))]


def timed(f):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for name, spec, (c, t, h, w) in LAYERS:
    is3d = spec.k[0] > 1
    xs = (B, c, t, h, w) if is3d else (B, c, h, w)
    x = torch.randn(xs, device="cuda")
    wp_f, wp_d = ops.weight_pack(torch.randn(spec.cout, spec.cin, spec.ntaps, device="cuda") * 0.05)
    ops.x6_presplit([wp_f, wp_d], [True, True])
    res = torch.randn((B, spec.cout) + tuple(spec.out_dims(t, h, w)[(0 if is3d else 1):]), device="cuda")
    P = t * h * w
    planes = torch.empty(3 * B * (c // 8) * P * 8, device="cuda", dtype=torch.int16)
    stream = lambda: torch.cuda.current_stream().cuda_stream
    split = lambda: _hip.check(lib.p2i_x6_split_planes(x.data_ptr(), planes.data_ptr(), B, c, P, stream()), "split")
    plain = lambda: ops.conv_fwd(spec, x, wp_f, residual=res, act=ops.ACT_NONE)

    def pre():
        lib.p2i_x6_next_source_planes(planes.data_ptr())
        return ops.conv_fwd(spec, x, wp_f, residual=res, act=ops.ACT_NONE)

    split()
    y0, y1 = plain(), pre()
    import ctypes as C
    plan = (C.c_int * 6)()
    lib.p2i_conv_last_plan(plan)
    same = bool(torch.equal(y0, y1))
    t_plain, t_pre, t_split = timed(plain), timed(pre), timed(split)
    print(f"B={B} {name:18s} plan={tuple(plan)} bit-equal={same} max|diff|={float((y0 - y1).abs().max()):.3e}  in-kernel split {t_plain:6.1f} us | "
          f"pre-split source {t_pre:6.1f} us | standalone split pass {t_split:5.1f} us", flush=True)
