"""Datasets producing the (video, masked_video, mask) triples of the reference (sti_dataset.py:18-325),
each (T,H,W,1) float32 in [0,1].  Sources: a directory of events (.npy always; .h5 when h5py is
importable), a Zarr-v2 store of flat event arrays, the windowed train.zarr layout, or the synthetic
generator of SURVEY.md §8d (`synthetic://<n_events>`).  Mask semantics follow create_mask."""
from __future__ import annotations

import os
import random
import re
from typing import List, Optional

import numpy as np
import torch

from ..utils import seeded
from . import zarr_lite


def _sti_matrix(H: int, W: int, block: int) -> np.ndarray:
    """One random pixel per block x block cell, drawn from numpy's global RNG in the reference's order (row of
    blocks outer, randint(h) then randint(w): sti_dataset.py:44-58).  One vectorised randint over the interleaved
    (h, w) bounds consumes the legacy RandomState stream exactly like the reference's per-cell scalar calls
    (tests/test_data_cpu.py checks both the matrix and the RNG state), at 1/30 of the Python cost."""
    hh, ww = np.meshgrid(np.arange(0, H, block), np.arange(0, W, block), indexing="ij")
    lo = np.stack([hh.ravel(), ww.ravel()], 1).ravel()
    hi = np.stack([np.minimum(hh + block, H).ravel(), np.minimum(ww + block, W).ravel()], 1).ravel()
    r = np.random.randint(lo, hi)
    m = np.zeros((H, W), dtype=np.float32)
    m[r[0::2], r[1::2]] = 1.0
    return m


def create_mask(video_tensor, mask_type="sti", mask_file=None, block_sizes=(4,), keep=4, interval=(2, 5)):
    """Mask (T,H,W,C) float32, 1 = observed.  Types: sti / fi / nowcasting / stin / stis (sti_dataset.py:18-122)."""
    T, H, W, C = video_tensor.shape
    if mask_type == "sti":
        block = np.random.choice(list(block_sizes))
        mat = torch.from_numpy(_sti_matrix(H, W, int(block)))
        return mat.reshape(1, H, W, 1).repeat(T, 1, 1, C)
    if mask_type == "fi":
        mask = torch.zeros((T, H, W, C), dtype=torch.float32)
        step = int(np.random.choice(list(interval)))
        mask[0:T:step + 1] = 1.0
        return mask
    if mask_type == "nowcasting":
        mask = torch.ones((T, H, W, C), dtype=torch.float32)
        mask[keep:] = 0.0
        return mask
    if mask_type == "stin":
        mask = torch.ones((T, H, W, C), dtype=torch.float32)
        for _ in range(keep, T):                      # the reference redraws per frame and keeps the LAST draw (:84-99)
            block = np.random.choice(list(block_sizes))
            mat = torch.from_numpy(_sti_matrix(H, W, int(block)))
            mask = mat.reshape(1, H, W, 1).repeat(T, 1, 1, C)
        mask[:keep] = 1.0
        return mask
    if mask_type == "stis" and mask_file is not None:
        mat = torch.tensor(np.loadtxt(mask_file), dtype=torch.bool)
        if tuple(mat.shape) != (H, W):
            raise ValueError(f"Mask matrix in {mask_file} does not match video spatial dimensions {H}x{W}")
        return mat.float().reshape(1, H, W, 1).repeat(T, 1, 1, C)
    raise ValueError("Invalid mask type or mask file not provided for 'selfdefine' mask.")


def _number(name: str) -> int:
    m = re.search(r"\d+", name)
    return int(m.group()) if m else -1


class Dataset(torch.utils.data.Dataset):
    """Whole-event dataset (sti_dataset.py:128-239)."""

    def __init__(self, args):
        root = str(args["data_root"])
        self.root = root
        self.synthetic = root.startswith("synthetic://")
        self.is_zarr = root.endswith(".zarr")
        self.zroot = None
        if self.synthetic:
            self.n_syn = int(root.split("://")[1] or 8)
            self.video_files: List = list(range(self.n_syn))
        elif self.is_zarr:
            self.zroot = zarr_lite.open_group(root, "r")
            self.video_files = sorted(self.zroot.array_keys())
        else:
            self.video_files = sorted([os.path.join(root, f) for f in os.listdir(root) if f.endswith((".npy", ".h5"))],
                                      key=lambda f: _number(os.path.basename(f)))
        m = args.get("mask", {})
        self.mask_type, self.mask_file = m.get("type", "sti"), m.get("file")
        self.block_sizes, self.mask_keep, self.mask_interval = m.get("block_sizes", [4]), m.get("keep", 4), m.get("interval", [2, 5])
        self.width, self.height = args["w"], args["h"]
        self.sample_length = args.get("sample_length")
        self.syn_len = args.get("synthetic_length", 16)
        self.raw_u8 = bool(args.get("device_assemble", False))

    def __len__(self):
        return len(self.video_files)

    def _load(self, item) -> np.ndarray:
        if self.synthetic:
            return seeded.synthetic_event(self.sample_length or self.syn_len, self.height, self.width, seed=2024 + item).numpy()
        if self.is_zarr:
            return np.asarray(self.zroot[item][:])
        if item.endswith(".npy"):
            return np.load(item)
        import h5py  # optional dependency
        with h5py.File(item, "r") as f:
            return f["frames"][:]

    def __getitem__(self, idx):
        v = self._load(self.video_files[idx])
        if v.ndim == 3:
            v = v[..., None]
        elif v.shape[-1] != 1:
            v = v.mean(axis=-1, keepdims=True)
        if self.sample_length is not None:
            v = v[:min(self.sample_length, v.shape[0])]
        if self.raw_u8 and v.dtype == np.uint8:
            # device-side assembly (ops.assemble_batch): ship uint8 frames + uint8 mask, same RNG consumption
            mask = create_mask(torch.empty((v.shape[0], v.shape[1], v.shape[2], 1), dtype=torch.uint8), self.mask_type, self.mask_file, self.block_sizes,
                               self.mask_keep, self.mask_interval)
            return self._crop(torch.from_numpy(np.ascontiguousarray(v)))[..., 0].contiguous(), \
                self._crop(mask)[..., 0].to(torch.uint8).contiguous()
        video = torch.from_numpy(v.astype(np.float32) / 255.0)
        mask = create_mask(video, self.mask_type, self.mask_file, self.block_sizes, self.mask_keep, self.mask_interval)
        masked = video * mask
        return self._crop(video), self._crop(masked), self._crop(mask)

    def _crop(self, d):
        if d.shape[1] == self.height and d.shape[2] == self.width:
            return d
        y0 = max((d.shape[1] - self.height) // 2, 0)
        x0 = max((d.shape[2] - self.width) // 2, 0)
        return d[:, y0:y0 + self.height, x0:x0 + self.width, :]


class Dataset_ZarrTrain(torch.utils.data.Dataset):
    """Window dataset over events/<key>/frames + index/windows (sti_dataset.py:245-325).  As in the
    reference the window LENGTH comes from the index (the model needs 16: SURVEY.md C5)."""

    def __init__(self, args):
        self.z = zarr_lite.open_group(args["data_root"], "r")
        self.events = self.z["events"]
        self.index = np.asarray(self.z["index"]["windows"][:])
        self.keys = sorted(self.events.keys())
        self._frames = {}            # event key -> opened `frames` array (metadata parsed once, chunk cache kept: zarr_lite.Array)
        self.crop_h, self.crop_w = args["h"], args["w"]
        m = args.get("mask", {})
        self.mask_type, self.mask_file = m.get("type", "sti"), m.get("file")
        self.block_sizes, self.mask_keep, self.mask_interval = m.get("block_sizes", [4]), m.get("keep", 4), m.get("interval", [2, 5])
        self.raw_u8 = bool(args.get("device_assemble", False))

    def __len__(self):
        return self.index.shape[0]

    def __getitem__(self, idx):
        ev, t0, L = (int(x) for x in self.index[idx])
        fr = self._frames.get(ev)
        if fr is None:
            fr = self._frames[ev] = self.events[self.keys[ev]]["frames"]
        T, H, W = fr.shape
        y0 = 0 if H == self.crop_h else random.randint(0, H - self.crop_h)
        x0 = 0 if W == self.crop_w else random.randint(0, W - self.crop_w)
        v = np.asarray(fr[t0:t0 + L, y0:y0 + self.crop_h, x0:x0 + self.crop_w])
        if self.raw_u8 and v.dtype == np.uint8:
            mask = create_mask(torch.empty((v.shape[0], v.shape[1], v.shape[2], 1), dtype=torch.uint8), self.mask_type, self.mask_file,
                               self.block_sizes, self.mask_keep, self.mask_interval)
            return torch.from_numpy(np.ascontiguousarray(v)), mask[..., 0].to(torch.uint8).contiguous()
        video = torch.from_numpy(v.astype(np.float32) / 255.0).unsqueeze(-1)
        mask = create_mask(video, self.mask_type, self.mask_file, self.block_sizes, self.mask_keep, self.mask_interval)
        return video, video * mask, mask
