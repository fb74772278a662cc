"""Where a fwd/dgrad chunk iteration spends its cycles (diagnostic build: tools/build_stamp.sh, -DP2I_STAMP).
usage: P2I_HIP_LIB=build/ab/libp2i_hip_stamp.so python tools/stamp_conv.py [B=8]
Per layer: median over waves of the per-wave cycle sums {vmcnt wait, barrier, DMA issue, LDS reads + MFMA} and the loop total.
Read SHARES, not times (the stamps' fences forbid overlaps the real kernel has)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "p2i-gan-benchmark_amd"))
import torch
from p2igan_bench import _hip, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
lib = _hip.load()
lib.p2i_debug_set_stamp.argtypes = [ctypes.c_void_p]
buf = torch.zeros(8 * 8 * 65536, dtype=torch.int64, device="cuda")
dev = "cuda"
for name, C, S in (("l0 64@128", 64, 128), ("l1 128@64", 128, 64), ("l2 256@32", 256, 32), ("l3 512@16", 512, 16)):
    spec = ops.ConvSpec(C, C, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    x = torch.randn(B, C, S, S, device=dev)
    wp_f, wp_d = ops.weight_pack(torch.randn(C, C, 9, device=dev) * 0.05)
    for kind in ("fwd", "dgrad"):
        f = (lambda: ops.conv_fwd(spec, x, wp_f, act=ops.ACT_RELU)) if kind == "fwd" else (lambda: ops.conv_dgrad(spec, x, wp_d, tuple(x.shape), add=x))
        f(); f()
        torch.cuda.synchronize()
        buf.zero_()
        lib.p2i_debug_set_stamp(ctypes.c_void_p(buf.data_ptr()))
        f()
        torch.cuda.synchronize()
        lib.p2i_debug_set_stamp(ctypes.c_void_p(0))
        plan = (ctypes.c_int * 6)()
        lib.p2i_conv_last_plan(plan)
        r = buf.view(-1, 8).cpu()
        r = r[r[:, 4] > 0].double()
        med = r.median(0).values
        tot = med[4]
        t0, t1 = r[:, 5].min(), (r[:, 5] + r[:, 4]).max()
        print(f"{name:10s} {kind:5s} plan={tuple(plan)} waves={r.shape[0]:5d} loop cycles {tot:9.0f}  wait {med[0]/tot:5.1%} barrier {med[1]/tot:5.1%} "
              f"issue {med[2]/tot:5.1%} reads+mfma {med[3]/tot:5.1%}   prologue {med[6]:8.0f} cyc  epilogue(kg0 waves) {r[:,7][r[:,7]>0].median():8.0f} cyc", flush=True)
