"""Producer / consumer kernel (patch_gemm_x6p_kernel): where the waves of a workgroup wait (diagnostic build, -DP2I_STAMP).
usage: P2I_HIP_LIB=build/ab/libp2i_hip_stamp.so python tools/stamp_x6p.py [B=8]
consumer rows: cycles waiting for own LDS reads / parked at the stage barrier / loop total;
producer rows: wait for patch loads / staging work (split, load issue, DMA issue) / wait for the next stage's weights / parked at the barrier."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "p2i-gan-benchmark_amd"))
import torch
from p2igan_bench import _hip, ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
lib = _hip.load()
lib.p2i_debug_set_stamp.argtypes = [ctypes.c_void_p]
buf = torch.zeros(8 * 8 * 65536, dtype=torch.int64, device="cuda")
for name, C, S in (("l1 128@64", 128, 64), ("l2 256@32", 256, 32)):
    spec = ops.ConvSpec(C, C, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    x = torch.randn(B, C, S, S, device="cuda")
    wp_f, wp_d = ops.weight_pack(torch.randn(C, C, 9, device="cuda") * 0.05)
    f = lambda: ops.conv_fwd(spec, x, wp_f, act=ops.ACT_RELU)
    f(); f(); torch.cuda.synchronize(); buf.zero_()
    lib.p2i_debug_set_stamp(ctypes.c_void_p(buf.data_ptr())); f(); torch.cuda.synchronize(); lib.p2i_debug_set_stamp(ctypes.c_void_p(0))
    r = buf.view(-1, 8, 8).cpu().double()
    r = r[r[:, 0, 4] > 0]
    cons, prod = r[:, :4].reshape(-1, 8), r[:, 4:].reshape(-1, 8)
    mc, mp = cons.median(0).values, prod.median(0).values
    print(f"{name}: workgroups {r.shape[0]}")
    print(f"  consumer: prologue {mc[5]:8.0f} | loop {mc[4]:9.0f} cyc | after the loop (last tap, exchange, epilogue) {mc[7]:8.0f} | lds wait {mc[0]/mc[4]:5.1%} | barrier {mc[1]/mc[4]:5.1%}")
    print(f"  producer: loop {mp[4]:9.0f} cyc | patch wait {mp[0]/mp[4]:5.1%} | split pass {mp[2]/mp[4]:5.1%} | load issue {mp[6]/mp[4]:5.1%} | "
          f"weight wait {mp[3]/mp[4]:5.1%} | barrier {mp[1]/mp[4]:5.1%} | prologue {mp[5]:8.0f} | after loop {mp[7]:8.0f}", flush=True)
