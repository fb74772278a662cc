"""Synthetic train.zarr in the reference's windowed layout (preprocess.py:195-225: `events/<key>/frames` uint8 with
chunks (20, 128, 128), `index/windows` int32 (N, 3) rows [event, t0, length], `.zattrs suggested_window`), written with the
dependency-free zarr_lite store.  Feeds configs[2] (BASELINE.json: "synthetic Zarr train.zarr windows") in tests and in
`bench.py --with-loader`; events are the SURVEY 8d rain-like fields of utils.seeded.synthetic_event."""
from __future__ import annotations

import os

import numpy as np

from ..utils import seeded
from . import zarr_lite


def write_train_zarr(root: str, n_events: int = 4, frames_per_event: int = 30, h: int = 128, w: int = 128, window: int = 16,
                     stride: int = 2, seed: int = 2024, compress: bool = False) -> int:
    """Creates `root` (must end in train.zarr: dataloader.py:89-92 keys on the name).  Returns the number of windows."""
    if not str(root).rstrip("/").endswith("train.zarr"):
        raise ValueError("the windowed training store must be named train.zarr")
    g = zarr_lite.Group(str(root), "w")
    ev = g.require_group("events")
    rows = []
    for e in range(n_events):
        key = "2018%02d%02d0000" % (1 + e // 28, 1 + e % 28)
        fr = seeded.synthetic_event(frames_per_event, h, w, seed=seed + e).numpy()
        ev.require_group(key).create_dataset("frames", fr, chunks=(20, min(128, h), min(128, w)), compress=compress)
        rows += [[e, t0, window] for t0 in range(0, frames_per_event - window + 1, stride)]
    g.require_group("index").create_dataset("windows", np.asarray(rows, dtype=np.int32))
    g.attrs.update(suggested_window=window)
    return len(rows)
