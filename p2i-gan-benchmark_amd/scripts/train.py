#!/usr/bin/env python3
"""Train the P2I-GAN on the MI355X path.  Same CLI flags and config keys as the reference's
scripts/train.py:26-64 (config surface: SURVEY.md §5).  Launch with
`python -m torch.distributed.run --nproc-per-node N scripts/train.py --config ...` for data-parallel."""
from __future__ import annotations

import argparse
import json
import logging
import math
import os
import random
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from p2igan_bench import ops, parallel  # noqa: E402
from p2igan_bench.data.dataloader import P2IDataModule  # noqa: E402
from p2igan_bench.data.prefetch import DevicePrefetcher  # noqa: E402
from p2igan_bench.engine import TrainEngine  # noqa: E402
from p2igan_bench.models import build_discriminator, build_generator  # noqa: E402


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Train P2I-GAN benchmark model")
    p.add_argument("--config", type=Path, default=Path("p2igan_bench/config/p2igan_baseline.json"))
    p.add_argument("--experiment-name", type=str, default=None)
    p.add_argument("--run-name", type=str, default=None)
    p.add_argument("--tracking-uri", type=str, default=None)
    p.add_argument("--log-level", type=str, default="INFO")
    p.add_argument("--run-validation", dest="run_validation", action="store_true")
    p.add_argument("--skip-validation", dest="run_validation", action="store_false")
    p.set_defaults(run_validation=None)
    p.add_argument("--run-test", dest="run_test", action="store_true")
    p.add_argument("--skip-test", dest="run_test", action="store_false")
    p.set_defaults(run_test=None)
    return p.parse_args(argv)


def load_config(path: Path):
    if not path.exists():
        raise FileNotFoundError(path)
    with path.open("r", encoding="utf-8") as f:
        if path.suffix in {".yaml", ".yml"}:
            import yaml
            return yaml.safe_load(f)
        return json.load(f)


def seed_everything(seed: int, rank: int = 0):
    """train.py:78-82.  Under data parallelism the mask / crop generators (create_mask draws from np.random, Dataset_ZarrTrain
    from random) get seed + rank so that the ranks of a step see different masks and crops; torch's generator keeps the common
    seed, so every rank builds identical initial weights (they are broadcast from rank 0 anyway)."""
    random.seed(seed + rank)
    np.random.seed(seed + rank)
    torch.manual_seed(seed)


class _Tracker:
    """MLflow when importable (train.py:185-221), otherwise plain logging."""

    def __init__(self, cfg, uri, enabled):
        self.ml = None
        if enabled:
            try:
                import mlflow
                self.ml = mlflow
                if uri or "MLFLOW_TRACKING_URI" in os.environ:
                    mlflow.set_tracking_uri(uri or os.environ["MLFLOW_TRACKING_URI"])
                if cfg.get("experiment_name"):
                    mlflow.set_experiment(cfg["experiment_name"])
                mlflow.start_run(run_name=cfg.get("run_name"))
            except ImportError:
                logging.info("mlflow not installed: metrics go to the log only")

    def metric(self, key, value, step):
        if self.ml:
            self.ml.log_metric(key, float(value), step=step)

    def artifact(self, path):
        if self.ml:
            self.ml.log_artifact(str(path))

    def close(self):
        if self.ml:
            self.ml.end_run()


class Trainer:
    def __init__(self, cfg, rank=0, world=1, local=0):
        self.cfg, self.rank, self.world = cfg, rank, world
        seed_everything(cfg.get("seed", 42), rank)
        if not torch.cuda.is_available():
            raise RuntimeError("the MI355X path needs a GPU: there is no CPU fallback")
        self.device = torch.device("cuda", local)
        torch.cuda.set_device(self.device)
        dm = P2IDataModule(cfg, rank, world)
        self.train_loader, self.val_loader = dm.train_dataloader(), dm.val_dataloader()
        self.test_loader = dm.test_dataloader()
        tr = cfg.get("train", {})
        self.run_validation = bool(tr.get("use_validation", True))
        # train.py:111,164: the reference stores use_test / test_interval and builds the test loader but never runs it; here the
        # flags drive a periodic pass over the test events (rec loss on the first 16 frames of each), every test_interval epochs
        self.run_test = bool(tr.get("use_test", True))
        self.test_interval = int(tr.get("test_interval", 20))
        logging.info("Data loaders ready | train=%s, val=%s, test=%s", len(self.train_loader),
                     len(self.val_loader) if self.val_loader is not None else 0, len(self.test_loader) if self.test_loader is not None else 0)
        self.generator = build_generator(cfg).to(self.device)
        self.discriminator = build_discriminator(cfg).to(self.device) if cfg["loss"].get("use_gan", 0) else None
        self.engine = TrainEngine(self.generator, self.discriminator, cfg, distributed=world > 1)
        self.save_dir = Path(cfg.get("save_dir", "weights"))
        if rank == 0:
            self.save_dir.mkdir(parents=True, exist_ok=True)
        self.log_every = int(tr.get("log_step", 100))
        self.global_step = 0
        n = max(1, len(self.train_loader))
        self.max_steps = tr.get("iterations")
        self.max_epochs = tr.get("max_epochs") or (math.ceil(self.max_steps / n) if self.max_steps else tr.get("niter", 1))
        if self.max_steps is None:
            self.max_steps = self.max_epochs * n
        self.best_val = float("inf")

    def _batch(self, batch):
        if len(batch) == 2:        # train.device_assemble: (uint8 frames (B,T,H,W), uint8 masks) -> fp32 triple on the device
            fr, mk = (t.to(self.device, non_blocking=True) for t in batch)
            return list(ops.assemble_batch(fr.contiguous(), mk.contiguous()))
        return [t.permute(0, 1, 4, 2, 3).contiguous().to(self.device, non_blocking=True) for t in batch]   # train.py:468-473

    def _evaluate_rec_loss(self, loader, max_frames=None):
        """train.py:369-382 on the HIP path.  One process (world == 1): the reference's own figure, the mean of the BATCH means
        (total_loss / batches) -- a short last batch weighs as much as a full one, and val_loss / the best.pt decision equal the
        reference's.  Data parallel (world > 1, no reference behaviour: its Trainer is single-process): the evaluation loaders shard
        the split without padding, so ranks see different batch counts and sizes; there (loss sum weighted by samples, sample count)
        are all-reduced and every rank (and the best.pt decision on rank 0) sees the mean over the whole split with every sample
        counted once.  Returns (mean loss, number of samples)."""
        tot = torch.zeros(2, dtype=torch.float64, device=self.device)
        bsum, nb = torch.zeros((), dtype=torch.float64, device=self.device), 0
        skipped = 0
        for batch in loader:
            fr, mk, ms = self._batch(batch)
            if max_frames is not None and fr.shape[1] > max_frames:      # test events are full-length: first window only
                fr, mk, ms = (t[:, :max_frames].contiguous() for t in (fr, mk, ms))
            if fr.shape[1] != self.generator.length:
                skipped += fr.shape[0]
                continue
            bl = self.engine.eval_rec_loss(fr, mk, ms).double()
            bsum, nb = bsum + bl, nb + 1
            tot[0] += bl * fr.shape[0]                                                   # batch mean -> sample-weighted sum
            tot[1] += fr.shape[0]
        if skipped:
            logging.warning("evaluation skipped %d sample(s) whose length differs from the generator's %d frames", skipped, self.generator.length)
        if self.world > 1:
            torch.distributed.all_reduce(tot)
        ns = int(tot[1])
        if self.world == 1:
            return (float(bsum) / nb if nb else float("nan")), ns
        return (float(tot[0]) / ns if ns else float("nan")), ns

    def train(self, tracker):
        for epoch in range(1, self.max_epochs + 1):
            if hasattr(self.train_loader.sampler, "set_epoch"):
                self.train_loader.sampler.set_epoch(epoch)
            run, steps = None, 0
            # batches arrive on the device one step ahead (copy stream, pageable source -- pinned staging measured 17 ms slower per batch
            # on this platform: data/prefetch.py); train.prefetch=false
            # keeps the reference's in-line hand-over (train.py:468-473)
            feed = DevicePrefetcher(self.train_loader, self.device) if self.cfg.get("train", {}).get("prefetch", True) else \
                (self._batch(b) for b in self.train_loader)
            for batch in feed:
                out = self.engine.train_step(*batch)
                # stays on the device; cloned because a graph-replayed step returns the same output tensors every call
                run = out["loss_g"].clone() if run is None else run + out["loss_g"]
                steps += 1
                self.global_step += 1
                if self.global_step % self.log_every == 0 and self.rank == 0:       # the only host sync of the loop
                    vals = {k: float(out[k]) for k in ("loss_g", "rec", "pool", "reg", "adv", "loss_d") if k in out}
                    for k, v in vals.items():
                        tracker.metric(f"train/{k}", v, self.global_step)
                    logging.info("Epoch %d | step %d/%d | %s", epoch, self.global_step, self.max_steps,
                                 " ".join(f"{k}={v:.4f}" for k, v in vals.items()))
                if self.global_step >= self.max_steps:
                    break
            train_loss = float(run) / max(1, steps)
            logging.info("Epoch %d completed | train_loss=%.4f | global_step=%d", epoch, train_loss, self.global_step)
            val_loss = None
            if self.run_validation and self.val_loader is not None:
                vl, nb = self._evaluate_rec_loss(self.val_loader)
                if nb:                                      # an empty split must not pin best.pt at 0.0
                    val_loss = vl
                    tracker.metric("val/loss", val_loss, self.global_step)
                    logging.info("Validation done | val_loss=%.4f (%d samples)", val_loss, nb)
            if self.run_test and self.test_loader is not None and self.test_interval > 0 and epoch % self.test_interval == 0:
                tl, nb = self._evaluate_rec_loss(self.test_loader, max_frames=self.generator.length)
                if nb:
                    tracker.metric("test/loss", tl, self.global_step)
                    logging.info("Test pass | test_loss=%.4f (%d events)", tl, nb)
            if self.rank == 0:
                ck = self.engine.checkpoint(epoch, self.global_step)
                torch.save(ck, self.save_dir / "latest.pt")
                tracker.artifact(self.save_dir / "latest.pt")
                if val_loss is not None and val_loss < self.best_val:      # (reference reads an unbound val_loss here: train.py:215)
                    self.best_val = val_loss
                    torch.save(ck, self.save_dir / "best.pt")
            if self.global_step >= self.max_steps:
                break


def main(argv=None):
    args = parse_args(argv)
    logging.basicConfig(level=getattr(logging, args.log_level.upper(), logging.INFO), format="%(asctime)s | %(levelname)s | %(message)s")
    cfg = load_config(args.config)
    tr = cfg.setdefault("train", {})
    if args.experiment_name:
        cfg["experiment_name"] = args.experiment_name
    if args.run_name:
        cfg["run_name"] = args.run_name
    if args.run_validation is not None:
        tr["use_validation"] = bool(args.run_validation)
    if args.run_test is not None:
        tr["use_test"] = bool(args.run_test)
    rank, world, local = parallel.init_distributed()
    tracker = _Tracker(cfg, args.tracking_uri, enabled=rank == 0)
    try:
        Trainer(cfg, rank, world, local).train(tracker)
    finally:
        tracker.close()


if __name__ == "__main__":
    main()
