"""16-B patch DMA (P2I_CONV_V4) on vs off: forward and stride-1 dgrad must agree bit for bit (same MFMA order)."""
import itertools, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "p2i-gan-benchmark_amd"))
from p2igan_bench import ops
torch.manual_seed(0)
bad = 0
cases = []
for B, C, S in [(2, 64, 32), (2, 128, 16), (2, 256, 8), (2, 512, 4), (3, 64, 8), (1, 32, 12), (5, 48, 20)]:
    cases.append((B, C, C, (1, S, S), (1, 3, 3), (1, 1, 1), (0, 1, 1)))
cases += [(2, 128, 64, (1, 32, 32), (1, 1, 1), (1, 1, 1), (0, 0, 0)), (2, 16, 64, (1, 32, 32), (1, 3, 3), (1, 1, 1), (0, 1, 1)),
          (2, 64, 16, (1, 32, 32), (1, 1, 1), (1, 1, 1), (0, 0, 0)), (2, 128, 128, (16, 4, 4), (3, 3, 3), (2, 1, 1), (1, 1, 1)),
          (2, 128, 1, (8, 4, 4), (1, 1, 1), (1, 1, 1), (0, 0, 0)), (2, 256, 256, (1, 8, 8), (1, 3, 3), (1, 1, 1), (0, 1, 1)),
          (2, 256, 1, (1, 8, 8), (1, 3, 3), (1, 1, 1), (0, 1, 1)), (2, 16, 64, (1, 32, 32), (1, 3, 3), (1, 1, 1), (0, 1, 1)),
          (2, 32, 64, (16, 16, 16), (3, 3, 3), (1, 2, 2), (1, 1, 1)), (2, 16, 32, (1, 24, 28), (1, 3, 3), (1, 1, 1), (0, 1, 1))]
for (B, Cin, Cout, sp, k, st, pd) in cases:
    spec = ops.ConvSpec(Cin, Cout, k, st, pd)
    x = torch.randn(B, Cin, *sp, device="cuda")
    w = torch.randn(Cout, Cin, k[0] * k[1] * k[2], device="cuda") * 0.05
    wp_f, wp_d = ops.weight_pack(w)
    res = {}
    for v in ("1", "0"):
        os.environ["P2I_CONV_V4"] = v
        y = ops.conv_fwd(spec, x, wp_f)
        dx = ops.conv_dgrad(spec, torch.ones_like(y) * 0.5 + y, wp_d, tuple(x.shape))
        import ctypes
        pl = (ctypes.c_int * 6)(); ops._hip.load().p2i_conv_last_plan(pl)
        res[v] = (y.clone(), dx.clone(), tuple(pl))
    dy = float((res["1"][0] - res["0"][0]).abs().max())
    ddx = float((res["1"][1] - res["0"][1]).abs().max())
    flag = "" if dy == 0 and ddx == 0 else "  <-- MISMATCH"
    bad += flag != ""
    print(B, Cin, Cout, sp, k, st, "fwd diff", dy, "dgrad diff", ddx, "max|y|", float(res["0"][0].abs().max()), res["1"][2], res["0"][2], flag, flush=True)
print("mismatching cases:", bad)
