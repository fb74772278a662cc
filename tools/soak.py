"""Soak: N train steps twice from the same state -- finite losses, bit-equal final parameters (the deterministic scratch ring wraps every
few steps, the event slots every ~6), stable device memory.   usage: python tools/soak.py [steps=300] [B=4] [hw=64]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "p2i-gan-benchmark_amd"))
import torch
from p2igan_bench.models import build_generator, build_discriminator
from p2igan_bench.engine import TrainEngine
from p2igan_bench.utils import seeded
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
hw = int(sys.argv[3]) if len(sys.argv) > 3 else 64
cfg = {"seed": 1, "model": {"name": "p2igan", "in_channels": 1}, "data": {"train": {"h": hw, "w": hw, "sample_length": 16}},
       "loss": {"use_gan": 1, "gan_loss": "hinge", "adversarial_weight": 0.01, "k1_weight": 0.01},
       "train": {"optimizer": {"lr": 1e-4, "beta1": 0.0, "beta2": 0.99}}}
dev = "cuda"
batches = [[t.to(dev) for t in seeded.synthetic_batch(B, 16, hw, hw, seeded.block_mask(hw, hw, 8, seed=s) if s % 2 else seeded.gauge_mask(hw, hw, 24, seed=s), seed=100 + s)] for s in range(6)]
finals = []
for rep in range(2):
    G, D = build_generator(cfg).to(dev), build_discriminator(cfg).to(dev)
    G.load_state_dict(seeded.seeded_generator_state(hw, hw)); D.load_state_dict(seeded.seeded_discriminator_state())
    eng = TrainEngine(G, D, cfg)
    mem0 = None
    for i in range(n):
        r = eng.train_step(*batches[i % len(batches)])
        if i == 20:
            torch.cuda.synchronize(); mem0 = torch.cuda.memory_allocated()
        if i % 100 == 99:
            print(f"rep {rep} step {i + 1}: loss_g {float(r['loss_g']):.5f} loss_d {float(r['loss_d']):.5f} finite {bool(torch.isfinite(r['preds']).all())}", flush=True)
    torch.cuda.synchronize()
    print(f"rep {rep}: memory allocated after 21 steps {mem0 / 2**20:.1f} MiB, after {n}: {torch.cuda.memory_allocated() / 2**20:.1f} MiB", flush=True)
    finals.append((eng.gp.flat.clone(), eng.dp.flat.clone()))
    assert bool(torch.isfinite(eng.gp.flat).all()) and bool(torch.isfinite(eng.dp.flat).all())
print("bit-equal final parameters after", n, "steps:", bool(torch.equal(finals[0][0], finals[1][0]) and torch.equal(finals[0][1], finals[1][1])),
      "max |diff| G", float((finals[0][0] - finals[1][0]).abs().max()), "D", float((finals[0][1] - finals[1][1]).abs().max()))
