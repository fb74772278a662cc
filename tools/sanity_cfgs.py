"""Full train steps at the other BASELINE.json configurations' per-GPU shapes (sanity: ms/step, frames/s, finite losses, peak memory).
usage: python tools/sanity_cfgs.py            (the table below)
       python tools/sanity_cfgs.py B H W [T]   (one case, e.g. under rocprofv3)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "p2i-gan-benchmark_amd"))
from p2igan_bench.engine import TrainEngine
from p2igan_bench.models import build_discriminator, build_generator
from p2igan_bench.utils import seeded


def run(B, H, W, T=16, steps=int(os.environ.get("SANITY_STEPS", "3")), note=""):
    cfg = {"seed": 1, "model": {"name": "p2igan", "in_channels": 1}, "data": {"train": {"h": H, "w": W, "sample_length": T}},
           "loss": {"use_gan": 1, "gan_loss": "hinge", "k1_weight": 0.05, "adversarial_weight": 0.01},
           "train": {"optimizer": {"lr": 1e-4, "beta1": 0.0, "beta2": 0.99}}}
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    G = build_generator(cfg).to(dev); D = build_discriminator(cfg).to(dev); eng = TrainEngine(G, D, cfg)
    f, k, m = [t.to(dev) for t in seeded.synthetic_batch(B, T, H, W, seeded.gauge_mask(H, W, 79 * H * W // 16384))]
    for _ in range(5):                       # (past TrainEngine.AUTO_GRAPH_AFTER when P2I_AUTO_GRAPH=1)
        out = eng.train_step(f, k, m)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        out = eng.train_step(f, k, m)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print(f"B={B} T={T} {H}x{W} {note}: {dt * 1e3:.1f} ms/step {B * T / dt:.0f} frames/s loss_g={float(out['loss_g']):.4f} "
          f"loss_d={float(out['loss_d']):.4f} launch={'graph' if getattr(eng, '_graph', None) is not None else 'eager'} finite={bool(torch.isfinite(out['preds']).all())} mem={torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
    del eng, G, D
    torch.cuda.empty_cache()


if len(sys.argv) >= 4:
    run(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), T=int(sys.argv[4]) if len(sys.argv) > 4 else 16, note="(command line)")
    sys.exit(0)
run(8, 128, 128, note="configs[1]/[2] per-GPU shape")
run(32, 128, 128, note="B=32 (north_star stack target batch)")
run(4, 256, 256, note="configs[3] per-GPU shape (B=16 over 4 GPUs)")
run(16, 256, 256, note="configs[3] on one GPU")
run(4, 128, 128, T=32, note="configs[4] per-GPU shape (B=32 over 8 GPUs; T=32 generalisation, parity unpinned)")
run(1, 128, 128, note="B=1 (launch-bound)")
run(2, 128, 128, note="B=2")
run(4, 128, 128, note="B=4")
run(2, 64, 96, note="non-square, non-power-of-two width")
run(2, 128, 160, note="non-power-of-two width")
run(2, 64, 64, note="small square (for comparison with 64x96)")
