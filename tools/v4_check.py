"""16-B patch DMA (P2I_CONV_V4, read once per process) on vs off: forward and dgrad must agree bit for bit (same MFMA order).
Runs itself once per setting in a child process and compares the saved outputs."""
import os, subprocess, sys, tempfile
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CASES = [(B, C, C, (1, S, S), (1, 3, 3), (1, 1, 1), (0, 1, 1)) for B, C, S in [(2, 64, 32), (2, 128, 16), (2, 256, 8), (2, 512, 4), (3, 64, 8), (1, 32, 12), (5, 48, 20)]]
CASES += [(2, 128, 64, (1, 32, 32), (1, 1, 1), (1, 1, 1), (0, 0, 0)), (2, 16, 64, (1, 32, 32), (1, 3, 3), (1, 1, 1), (0, 1, 1)),
          (2, 64, 16, (1, 32, 32), (1, 1, 1), (1, 1, 1), (0, 0, 0)), (2, 128, 128, (16, 4, 4), (3, 3, 3), (2, 1, 1), (1, 1, 1)),
          (2, 128, 1, (8, 4, 4), (1, 1, 1), (1, 1, 1), (0, 0, 0)), (2, 256, 256, (1, 8, 8), (1, 3, 3), (1, 1, 1), (0, 1, 1)),
          (2, 256, 1, (1, 8, 8), (1, 3, 3), (1, 1, 1), (0, 1, 1)), (2, 32, 64, (16, 16, 16), (3, 3, 3), (1, 2, 2), (1, 1, 1)),
          (2, 16, 32, (1, 24, 28), (1, 3, 3), (1, 1, 1), (0, 1, 1))]


def child(path):
    sys.path.insert(0, os.path.join(ROOT, "p2i-gan-benchmark_amd"))
    from p2igan_bench import ops
    out = []
    for i, (B, Cin, Cout, sp, k, st, pd) in enumerate(CASES):
        g = torch.Generator().manual_seed(i)
        spec = ops.ConvSpec(Cin, Cout, k, st, pd)
        x = torch.randn(B, Cin, *sp, generator=g).cuda()
        w = (torch.randn(Cout, Cin, k[0] * k[1] * k[2], generator=g) * 0.05).cuda()
        wp_f, wp_d = ops.weight_pack(w)
        y = ops.conv_fwd(spec, x, wp_f)
        dx = ops.conv_dgrad(spec, torch.ones_like(y) * 0.5 + y, wp_d, tuple(x.shape))
        out.append((y.cpu(), dx.cpu()))
    torch.save(out, path)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
        sys.exit(0)
    with tempfile.TemporaryDirectory() as td:
        res = {}
        for v in ("1", "0"):
            p = os.path.join(td, v + ".pt")
            subprocess.check_call([sys.executable, os.path.abspath(__file__), p], env=dict(os.environ, P2I_CONV_V4=v))
            res[v] = torch.load(p)
        bad = 0
        for c, (a, b) in zip(CASES, zip(res["1"], res["0"])):
            dy, ddx = float((a[0] - b[0]).abs().max()), float((a[1] - b[1]).abs().max())
            bad += (dy != 0 or ddx != 0)
            print(c, "fwd diff", dy, "dgrad diff", ddx, "" if dy == 0 and ddx == 0 else "  <-- MISMATCH", flush=True)
        print("mismatching cases:", bad)
