"""Host enqueue time and step time of one train step: eager launches vs hipGraph replay vs launch-tape replay (p2i_tape_replay).
usage: python tools/tape_probe.py [B=1,2,4,8] [steps=40]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "p2i-gan-benchmark_amd"))
import torch
from p2igan_bench.models import build_generator, build_discriminator
from p2igan_bench.engine import TrainEngine
from p2igan_bench.utils import seeded

Bs = [int(b) for b in (sys.argv[1] if len(sys.argv) > 1 else "1,2,4,8").split(",")]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cfg = {"seed": 1, "model": {"name": "p2igan", "in_channels": 1}, "data": {"train": {"h": 128, "w": 128, "sample_length": 16}},
       "loss": {"use_gan": 1, "gan_loss": "hinge", "adversarial_weight": 0.01, "k1_weight": 0.01},
       "train": {"optimizer": {"lr": 1e-4, "beta1": 0.0, "beta2": 0.99}}}
dev = "cuda"
for B in Bs:
    fr, mk_, ms = [t.to(dev) for t in seeded.synthetic_batch(B, 16, 128, 128, seeded.gauge_mask(128, 128, 79))]
    for mode in ("eager", "graph", "tape"):
        torch.manual_seed(1)
        G, D = build_generator(cfg).to(dev), build_discriminator(cfg).to(dev)
        eng = TrainEngine(G, D, cfg)
        if mode == "eager":
            for _ in range(5):
                eng.train_step(fr, mk_, ms)
        else:
            eng.capture(fr, mk_, ms, warmup=3, mode=mode)
            for _ in range(2):
                eng.train_step(fr, mk_, ms)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            eng.train_step(fr, mk_, ms)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        info = f" tape ops (kernels, memsets, events, streams) {eng.tape_info()}" if mode == "tape" else ""
        print(f"B={B} {mode:5s}: host enqueue {1e3 * (t1 - t0) / n:6.2f} ms/step, step {1e3 * (t2 - t0) / n:6.2f} ms{info}", flush=True)
        del eng, G, D
        torch.cuda.empty_cache()
