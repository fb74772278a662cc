"""P2IDataModule with the reference's config handling (dataloader.py:11-170): train/valid/test splits,
train.zarr window dataset + seeded 80/20 split, test batch size 1 with sample_length dropped,
variable-length collate; plus rank sharding for data-parallel runs."""
from __future__ import annotations

from copy import deepcopy

import torch
from torch.utils.data import DataLoader, Subset

from ..parallel import ShardedSampler
from .sti_dataset import Dataset, Dataset_ZarrTrain


class P2IDataModule:
    def __init__(self, cfg, rank: int = 0, world: int = 1):
        self.cfg, self.rank, self.world = cfg, rank, world
        data_cfg, tr = cfg["data"], cfg["train"]
        self.num_workers = tr.get("num_workers", 0)
        self.pin_memory = tr.get("pin_memory", True)
        self.persistent_workers = tr.get("persistent_workers", True)
        self.prefetch_factor = tr.get("prefetch_factor", 2)
        self.train_args = self._args(data_cfg["train"])
        # this build's extension: loaders hand over uint8 frames + uint8 masks, ops.assemble_batch finishes on the device
        self.train_args["device_assemble"] = bool(tr.get("device_assemble", False))
        shared = {k: deepcopy(self.train_args[k]) for k in ("w", "h", "sample_length", "mask") if k in self.train_args}
        self.valid_dataset = self.test_dataset = None
        self.valid_shuffle = self.test_shuffle = False
        if str(self.train_args.get("data_root", "")).endswith("train.zarr"):
            base = Dataset_ZarrTrain(self.train_args)
            self.train_dataset, self.valid_dataset = self._split(base, cfg.get("seed", 42))
        else:
            self.train_dataset = Dataset(self.train_args)
            if data_cfg.get("valid"):
                self.valid_shuffle = bool(data_cfg["valid"].get("shuffle", False))
                self.valid_dataset = Dataset(self._args(data_cfg["valid"], shared))
        if data_cfg.get("test"):
            d = deepcopy(shared)
            d.pop("sample_length", None)
            self.test_shuffle = bool(data_cfg["test"].get("shuffle", False))
            self.test_dataset = Dataset(self._args(data_cfg["test"], d))

    @staticmethod
    def _args(split, defaults=None):
        defaults = defaults or {}
        a = {}
        for k in ("w", "h", "sample_length"):
            if k in defaults:
                a[k] = defaults[k]
            if k in split:
                if split[k] is None:
                    a.pop(k, None)
                else:
                    a[k] = split[k]
        m = deepcopy(defaults.get("mask", {}))
        m.update(split.get("mask") or {})
        if m:
            a["mask"] = m
        if "synthetic_length" in split:
            a["synthetic_length"] = split["synthetic_length"]
        if "data_root" in split:
            a["data_root"] = split["data_root"]
        elif "data_root1" in split:
            a["data_root"] = split["data_root1"]
        else:
            raise KeyError("Dataset config requires 'data_root'.")
        return a

    @staticmethod
    def _split(ds, seed, ratio=0.8):
        n = len(ds)
        if n <= 1:
            return ds, None
        nv = min(max(int(n * (1 - ratio)), 1), n - 1)
        idx = torch.randperm(n, generator=torch.Generator().manual_seed(seed)).tolist()
        return Subset(ds, idx[:n - nv]), Subset(ds, idx[n - nv:])

    def _loader(self, ds, shuffle, bs, drop_last=None):
        if ds is None:
            return None
        base = ds.dataset if isinstance(ds, Subset) else ds
        collate = _collate_variable if getattr(base, "is_zarr", False) and getattr(base, "sample_length", None) is None else None
        # evaluation loaders (drop_last=False) shard without truncation or padding: every sample is seen exactly once
        sampler = ShardedSampler(len(ds), self.rank, self.world, shuffle, self.cfg.get("seed", 42), even=drop_last is not False) \
            if self.world > 1 else None
        return DataLoader(ds, batch_size=bs, shuffle=shuffle and sampler is None, sampler=sampler, num_workers=self.num_workers,
                          pin_memory=self.pin_memory, persistent_workers=self.num_workers > 0 and self.persistent_workers,
                          prefetch_factor=self.prefetch_factor if self.num_workers > 0 else None, collate_fn=collate,
                          drop_last=(self.world > 1) if drop_last is None else drop_last,
                          worker_init_fn=_WorkerSeed(self.cfg.get("seed", 42), self.rank) if self.num_workers > 0 else None)

    def train_dataloader(self):
        return self._loader(self.train_dataset, True, self.cfg["train"]["batch_size"])

    def val_dataloader(self):
        # evaluation splits keep their tail batch on every rank (the trainer all-reduces (sum, sample count))
        return self._loader(self.valid_dataset, self.valid_shuffle, self.cfg["train"]["batch_size"], drop_last=False)

    def test_dataloader(self):
        return self._loader(self.test_dataset, self.test_shuffle, 1, drop_last=False)


class _WorkerSeed:
    """worker_init_fn: numpy's and python's generators (mask draws, crop offsets) seeded per (epoch, rank, worker).
    torch has already seeded the worker with base_seed + worker_id by the time this runs, and base_seed is drawn anew every
    time the loader's workers are (re)started -- every epoch unless persistent_workers -- so deriving from torch.initial_seed()
    keeps the epoch-to-epoch variation of the reference (torch's default worker seeding) and adds the rank term on top."""

    def __init__(self, seed, rank):
        self.seed, self.rank = seed, rank

    def __call__(self, worker_id):
        import random
        import numpy as np
        s = (torch.initial_seed() + 1000003 * self.rank) % (2 ** 32)
        random.seed(s)
        np.random.seed(s)


def _collate_variable(batch):
    vids, masked, masks = zip(*batch)
    L = max(v.shape[0] for v in vids)

    def pad(s):
        return s if s.shape[0] == L else torch.cat([s, s[-1:].repeat(L - s.shape[0], 1, 1, 1)], dim=0)
    return tuple(torch.stack([pad(s) for s in seq], dim=0) for seq in (vids, masked, masks))
