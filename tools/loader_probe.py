"""Where does the host time of a loader-fed train step go?  (GPU box diagnostic)
usage: python tools/loader_probe.py"""
import cProfile
import os
import pstats
import sys
import tempfile
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "p2i-gan-benchmark_amd"))
import numpy as np
import torch

from p2igan_bench.data.dataloader import P2IDataModule
from p2igan_bench.data.synth_store import write_train_zarr

d = tempfile.mkdtemp()
root = os.path.join(d, "train.zarr")
t0 = time.time()
n = write_train_zarr(root, 64, 30, 128, 128)
print("store written:", n, "windows in", round(time.time() - t0, 2), "s at", d, flush=True)


def cfg_for(workers, pin):
    return {"seed": 11, "data": {"train": {"data_root": root, "w": 128, "h": 128, "sample_length": 16, "mask": {"type": "sti", "block_sizes": [10]}}},
            "train": {"batch_size": 8, "num_workers": workers, "device_assemble": True, "pin_memory": pin, "persistent_workers": workers > 0}}


ds = P2IDataModule(cfg_for(0, False)).train_dataset
for i in range(20):
    ds[i]
t0 = time.time()
for i in range(200):
    ds[i % len(ds)]
print("ds[i]: %.3f ms/sample" % ((time.time() - t0) / 200 * 1e3), flush=True)
pr = cProfile.Profile()
pr.enable()
for i in range(100):
    ds[i]
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(10)

for workers in (0, 4, 8):
    for pin in (False, True):
        l = P2IDataModule(cfg_for(workers, pin)).train_dataloader()
        it = iter(l)
        next(it)
        t0 = time.time()
        k = 0
        for b in it:
            k += 1
        dt = time.time() - t0
        print("workers %d pin %s: %.0f samples/s" % (workers, pin, k * 8 / dt), flush=True)
        del it, l

if torch.cuda.is_available():
    from p2igan_bench.engine import TrainEngine
    from p2igan_bench.models import build_discriminator, build_generator
    from p2igan_bench.utils import seeded
    H = W = 128
    cfg = {"seed": 1, "model": {"name": "p2igan", "in_channels": 1}, "data": {"train": {"h": H, "w": W, "sample_length": 16}},
           "loss": {"use_gan": 1, "gan_loss": "hinge", "k1_weight": 0.05, "adversarial_weight": 0.01},
           "train": {"optimizer": {"lr": 1e-4, "beta1": 0.0, "beta2": 0.99}}}
    dev = torch.device("cuda:0")
    for B in (8, 1):
        torch.manual_seed(0)
        os.environ["P2I_AUTO_GRAPH"] = "0"
        G = build_generator(cfg).to(dev); D = build_discriminator(cfg).to(dev); eng = TrainEngine(G, D, cfg)
        f, k, m = [t.to(dev) for t in seeded.synthetic_batch(B, 16, H, W, seeded.gauge_mask(H, W, 79))]
        for _ in range(3):
            eng.train_step(f, k, m)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            eng.train_step(f, k, m)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("B=%d eager: host enqueue %.2f ms/step, wall %.2f ms/step" % (B, (t1 - t0) / 10 * 1e3, (t2 - t0) / 10 * 1e3), flush=True)
        eng.capture(f, k, m, warmup=1)
        for _ in range(2):
            eng.train_step(f, k, m)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            eng.train_step(f, k, m)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("B=%d graph: host enqueue %.2f ms/step, wall %.2f ms/step" % (B, (t1 - t0) / 10 * 1e3, (t2 - t0) / 10 * 1e3), flush=True)
        del eng, G, D
