"""Where the HOST time of a train step goes (cProfile over steps at a batch small enough to be launch-bound).
usage: python tools/host_profile.py [B=1] [steps=30]"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "p2i-gan-benchmark_amd"))
import torch
from p2igan_bench.models import build_generator, build_discriminator
from p2igan_bench.engine import TrainEngine
from p2igan_bench.utils import seeded

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
cfg = {"seed": 1, "model": {"name": "p2igan", "in_channels": 1}, "data": {"train": {"h": 128, "w": 128, "sample_length": 16}},
       "loss": {"use_gan": 1, "gan_loss": "hinge", "adversarial_weight": 0.01, "k1_weight": 0.01},
       "train": {"optimizer": {"lr": 1e-4, "beta1": 0.0, "beta2": 0.99}}}
dev = "cuda"
G, D = build_generator(cfg).to(dev), build_discriminator(cfg).to(dev)
eng = TrainEngine(G, D, cfg)
fr, mk_, ms = [t.to(dev) for t in seeded.synthetic_batch(B, 16, 128, 128, seeded.gauge_mask(128, 128, 79))]
for _ in range(5):
    eng.train_step(fr, mk_, ms)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    eng.train_step(fr, mk_, ms)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"B={B}: host enqueue {1e3 * (t1 - t0) / n:.2f} ms/step, with drain {1e3 * (t2 - t0) / n:.2f} ms/step", flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    eng.train_step(fr, mk_, ms)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
