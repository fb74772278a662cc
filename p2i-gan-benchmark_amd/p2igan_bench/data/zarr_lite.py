"""Dependency-free Zarr-v2 directory store, just enough for the on-disk formats of the reference:
test stores (flat `event_%02d` float32 arrays, tozarr.py:105-111), the train store
(`events/<key>/frames` uint8 + `index/windows` int32, preprocess.py:195-225) and the inference
output group (infer.py:168-180,250-257).  Compressors: none or zlib (the reference's Blosc-zstd
chunks need the `zarr`/`numcodecs` packages, which are used instead when importable)."""
from __future__ import annotations

import itertools
import json
import os
import zlib
from typing import Dict, Optional, Tuple

import numpy as np


class Array:
    def __init__(self, path: str):
        self.path = path
        meta = json.load(open(os.path.join(path, ".zarray")))
        if meta.get("zarr_format") != 2 or meta.get("order", "C") != "C":
            raise ValueError(f"{path}: only zarr v2 C-order arrays are supported")
        comp = meta.get("compressor")
        if comp is not None and comp.get("id") != "zlib":
            raise ValueError(f"{path}: compressor {comp.get('id')} needs the zarr/numcodecs packages")
        if meta.get("filters"):
            raise ValueError(f"{path}: filters are not supported")
        self.shape: Tuple[int, ...] = tuple(meta["shape"])
        self.chunks: Tuple[int, ...] = tuple(meta["chunks"])
        self.dtype = np.dtype(meta["dtype"])
        self.fill = meta.get("fill_value") or 0
        self.zlib = comp is not None
        self.sep = meta.get("dimension_separator", ".")


    # Decoded chunks of read-only arrays, ONE least-recently-used budget per process shared by every Array (a training window
    # re-reads the same one or two chunks for every overlapping window of an event; on the GPU box an uncached window read cost
    # 2 ms of stat / open / read).  Round 3 kept up to 64 MB per ARRAY and a dataset keeps every opened event's Array: host memory
    # grew with the number of events x DataLoader workers.  P2I_ZARR_CACHE_MB sets the budget (default 256 MB per process).
    CACHE_BYTES = int(os.environ.get("P2I_ZARR_CACHE_MB", "256")) << 20
    _lru: "Dict[tuple, np.ndarray]" = {}          # (array path, chunk index) -> decoded chunk, most recently used last
    _lru_bytes = 0

    @classmethod
    def cache_clear(cls):
        Array._lru.clear()
        Array._lru_bytes = 0

    def _chunk(self, idx) -> np.ndarray:
        key = (self.path, idx)
        hit = Array._lru.pop(key, None)
        if hit is not None:
            Array._lru[key] = hit                  # most recently used last
            return hit
        f = os.path.join(self.path, self.sep.join(map(str, idx)) if idx else "0")
        try:
            with open(f, "rb") as fh:
                raw = fh.read()
        except FileNotFoundError:
            return np.full(self.chunks, self.fill, dtype=self.dtype)
        if self.zlib:
            raw = zlib.decompress(raw)
        ch = np.frombuffer(raw, dtype=self.dtype).reshape(self.chunks)
        if ch.nbytes <= Array.CACHE_BYTES:
            while Array._lru and Array._lru_bytes + ch.nbytes > Array.CACHE_BYTES:
                Array._lru_bytes -= Array._lru.pop(next(iter(Array._lru))).nbytes
            Array._lru[key] = ch
            Array._lru_bytes += ch.nbytes
        return ch

    def __getitem__(self, key) -> np.ndarray:
        if not isinstance(key, tuple):
            key = (key,)
        key = key + (slice(None),) * (len(self.shape) - len(key))
        sl = [k if isinstance(k, slice) else slice(k, k + 1) for k in key]
        rng = [s.indices(n)[:2] for s, n in zip(sl, self.shape)]
        out = np.empty([b - a for a, b in rng], dtype=self.dtype)
        grid = [range(a // c, (max(b, a + 1) - 1) // c + 1) for (a, b), c in zip(rng, self.chunks)]
        for idx in itertools.product(*grid):
            ch = self._chunk(idx)
            src, dst = [], []
            for i, (a, b), c in zip(idx, rng, self.chunks):
                lo, hi = max(a, i * c), min(b, (i + 1) * c)
                src.append(slice(lo - i * c, hi - i * c))
                dst.append(slice(lo - a, hi - a))
            out[tuple(dst)] = ch[tuple(src)]
        squeeze = tuple(i for i, k in enumerate(key) if not isinstance(k, slice))
        return out.squeeze(axis=squeeze) if squeeze else out

    def __len__(self):
        return self.shape[0]


class Group:
    def __init__(self, path: str, mode: str = "r"):
        self.path = path
        if mode == "w":
            os.makedirs(path, exist_ok=True)
            json.dump({"zarr_format": 2}, open(os.path.join(path, ".zgroup"), "w"))
        elif not os.path.exists(os.path.join(path, ".zgroup")):
            raise FileNotFoundError(f"{path} is not a zarr v2 group")
        self.attrs = _Attrs(path)

    def _children(self, marker):
        return sorted(d for d in os.listdir(self.path) if os.path.exists(os.path.join(self.path, d, marker)))

    def array_keys(self):
        return self._children(".zarray")

    def keys(self):
        return sorted(set(self._children(".zarray")) | set(self._children(".zgroup")))

    def __getitem__(self, key: str):
        p = os.path.join(self.path, key)
        if os.path.exists(os.path.join(p, ".zarray")):
            return Array(p)
        if os.path.exists(os.path.join(p, ".zgroup")):
            return Group(p)
        raise KeyError(key)

    def require_group(self, key: str) -> "Group":
        return Group(os.path.join(self.path, key), mode="w" if not os.path.exists(os.path.join(self.path, key, ".zgroup")) else "r")

    def create_dataset(self, name: str, data: np.ndarray, chunks: Optional[Tuple[int, ...]] = None, compress: bool = False):
        p = os.path.join(self.path, name)
        os.makedirs(p, exist_ok=True)
        data = np.ascontiguousarray(data)
        chunks = tuple(chunks or data.shape)
        meta = {"zarr_format": 2, "shape": list(data.shape), "chunks": list(chunks), "dtype": data.dtype.str,
                "compressor": {"id": "zlib", "level": 1} if compress else None, "fill_value": 0, "order": "C", "filters": None}
        json.dump(meta, open(os.path.join(p, ".zarray"), "w"))
        grid = [range((n + c - 1) // c) for n, c in zip(data.shape, chunks)]
        for idx in itertools.product(*grid):
            block = np.zeros(chunks, dtype=data.dtype)
            sl = tuple(slice(i * c, min((i + 1) * c, n)) for i, c, n in zip(idx, chunks, data.shape))
            block[tuple(slice(0, s.stop - s.start) for s in sl)] = data[sl]
            raw = block.tobytes()
            if compress:
                raw = zlib.compress(raw, 1)
            open(os.path.join(p, ".".join(map(str, idx)) if idx else "0"), "wb").write(raw)
        return Array(p)


class _Attrs(dict):
    def __init__(self, path):
        self._f = os.path.join(path, ".zattrs")
        super().__init__(json.load(open(self._f)) if os.path.exists(self._f) else {})

    def update(self, *a, **k):
        super().update(*a, **k)
        json.dump(dict(self), open(self._f, "w"))

    def __setitem__(self, k, v):
        super().__setitem__(k, v)
        json.dump(dict(self), open(self._f, "w"))


def open_group(path: str, mode: str = "r"):
    """zarr.open_group equivalent; defers to the real `zarr` package when it is importable."""
    try:
        import zarr  # type: ignore
        return zarr.open_group(path, mode=mode)
    except ImportError:
        return Group(path, mode)
