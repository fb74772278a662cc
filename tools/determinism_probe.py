"""Which gradients of a train step differ bit-wise between two runs from the same state?  (float atomics: VERDICT r3 weak #11)
usage: python tools/determinism_probe.py [B=2] [hw=64] [steps=3]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "p2i-gan-benchmark_amd"))
import torch
from p2igan_bench.models import build_generator, build_discriminator
from p2igan_bench.engine import TrainEngine
from p2igan_bench.utils import seeded

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cfg = {"seed": 1, "model": {"name": "p2igan", "in_channels": 1}, "data": {"train": {"h": hw, "w": hw, "sample_length": 16}},
       "loss": {"use_gan": 1, "gan_loss": "hinge", "adversarial_weight": 0.01, "k1_weight": 0.01},
       "train": {"optimizer": {"lr": 1e-4, "beta1": 0.0, "beta2": 0.99}}}
dev = "cuda"
fr, mk_, ms = [t.to(dev) for t in seeded.synthetic_batch(B, 16, hw, hw, seeded.gauge_mask(hw, hw, max(8, 79 * hw * hw // 16384)))]
runs = []
for rep in range(2):
    gs, ds = seeded.seeded_generator_state(hw, hw), seeded.seeded_discriminator_state()
    G, D = build_generator(cfg).to(dev), build_discriminator(cfg).to(dev)
    G.load_state_dict(gs); D.load_state_dict(ds)
    eng = TrainEngine(G, D, cfg)
    snap = []
    for s in range(steps):
        r = eng.train_step(fr, mk_, ms)
        torch.cuda.synchronize()
        snap.append(({n: p.grad.clone() for n, p in list(G.named_parameters()) + list(D.named_parameters()) if p.grad is not None},
                     {k: r[k].clone() for k in ("loss_g", "loss_d", "pool", "reg", "adv", "preds", "logits_real", "logits_fake")},
                     {n: b.clone() for n, b in D.named_buffers()}))
    runs.append(snap)
for s in range(steps):
    ga, oa, ba = runs[0][s]
    gb, ob, bb = runs[1][s]
    bad = [(n, float((ga[n] - gb[n]).abs().max() / (ga[n].abs().max() + 1e-30))) for n in ga if not torch.equal(ga[n], gb[n])]
    bado = [k for k in oa if not torch.equal(oa[k], ob[k])]
    badb = [n for n in ba if not torch.equal(ba[n], bb[n])]
    print(f"step {s}: {len(bad)} of {len(ga)} gradient tensors differ; outputs differing: {bado}; D buffers (u, v) differing: {len(badb)}")
    if s == 0:
        for n, e in sorted(bad, key=lambda t: -t[1])[:60]:
            print(f"   {n:50s} rel diff {e:.2e}")
