"""wgrad_x6_kernel: where a workgroup's time goes (diagnostic build, -DP2I_STAMP) and what bounds its tile loop (knock-out variants).
usage: P2I_HIP_LIB=build/ab/libp2i_hip_stamp.so python tools/stamp_wgrad_x6.py [B=8]
Per layer: median shader cycles of prologue (first tile staged) / tile loop / epilogue (centre-tap exchange + partial-tile store), the
share of the loop spent at the tile barrier, and the launch time (HIP events, us) of the full kernel and of its knock-out variants
(P2I_WGRAD_DIAG bits: 1 no global loads, 2 no split / LDS writes, 4 no MFMAs, 8 no operand reads; results are wrong by design)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "p2i-gan-benchmark_amd"))
import torch
from p2igan_bench import _hip, ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
lib = _hip.load()
lib.p2i_debug_set_stamp.argtypes = [ctypes.c_void_p]
buf = torch.zeros(8 * 8 * 65536, dtype=torch.int64, device="cuda")


def timed(f, n=20):
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


for name, C, S in (("l0 64@128", 64, 128), ("l1 128@64", 128, 64), ("l2 256@32", 256, 32), ("l3 512@16", 512, 16)):
    spec = ops.ConvSpec(C, C, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    x = torch.randn(B, C, S, S, device="cuda")
    dy = torch.randn(B, C, S, S, device="cuda")
    f = lambda: ops.conv_wgrad(spec, x, dy)
    os.environ.pop("P2I_WGRAD_DIAG", None)
    f(); f(); torch.cuda.synchronize(); buf.zero_()
    lib.p2i_debug_set_stamp(ctypes.c_void_p(buf.data_ptr())); f(); torch.cuda.synchronize(); lib.p2i_debug_set_stamp(ctypes.c_void_p(0))
    r = buf.view(-1, 8, 8).cpu().double()
    r = r[r[:, 0, 1] > 0].reshape(-1, 8)
    m = r.median(0).values
    span = (r[:, 5].max() - r[:, 4].min())
    print(f"{name}: waves {r.shape[0]}  prologue {m[0]:8.0f} | loop {m[1]:9.0f} | epilogue {m[2]:8.0f} cyc | barrier {m[3] / m[1]:5.1%} of the loop | "
          f"first entry -> last exit {span:9.0f} cyc (100 MHz s_memtime ticks)", flush=True)
    base = timed(f)
    line = f"   launch incl. reduce: full {base:6.1f} us"
    for dg, what in ((1, "no loads"), (2, "no split/writes"), (3, "no loads+split"), (4, "no MFMA"), (8, "no reads"), (12, "no MFMA+reads"), (11, "MFMA only"), (7, "reads only")):
        os.environ["P2I_WGRAD_DIAG"] = str(dg)
        line += f" | {what} {timed(f):6.1f}"
    os.environ.pop("P2I_WGRAD_DIAG", None)
    print(line, flush=True)
