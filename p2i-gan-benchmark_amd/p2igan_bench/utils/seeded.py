"""Portable, name-keyed parameter recipe (no torch RNG involved).

Used to give the reference model, the CPU oracle and the HIP path bit-identical weights
without shipping checkpoints: every tensor is drawn from a Philox stream keyed by
(seed, crc32(state_dict key)).  Scales follow the reference's init:
DO-Conv ``W`` ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (deconv_pytorch.py:58), conv weights
~ N(0, 2/fan_in) (layer.py:31, p2igan.py:150-155).  ``mode="test"`` additionally makes the
tensors the reference initialises to zero (``D``, biases, ``pos``, ``alpha*``) non-trivial so
that parity tests exercise every term; ``mode="init"`` keeps the reference's zeros.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Tuple

import numpy as np
import torch


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.Generator(np.random.Philox(key=[seed & 0xFFFFFFFF, zlib.crc32(name.encode())]))


def _normal(seed, name, shape, std):
    return torch.from_numpy((_rng(seed, name).standard_normal(shape) * std).astype(np.float32))


def _uniform(seed, name, shape, bound):
    return torch.from_numpy(_rng(seed, name).uniform(-bound, bound, shape).astype(np.float32))


def _unit(seed, name, n):
    v = _rng(seed, name).standard_normal(n).astype(np.float64)
    v /= max(np.linalg.norm(v), 1e-12)
    return torch.from_numpy(v.astype(np.float32))


def generator_shapes(h: int, w: int, t: int = 16, base: int = 0) -> Dict[str, Tuple[int, ...]]:
    """state_dict keys/shapes of P2IGenerator (p2igan.py:44-67) in the reference's order (base = 4t: 64 at T=16)."""
    base = base or 4 * t
    s: Dict[str, Tuple[int, ...]] = {}
    for i in range(2):
        s[f"input.layers.{i}.conv.weight"] = (t, t, 1)
        s[f"input.layers.{i}.conv.bias"] = (t,)
    for lvl in range(4):
        c = base << lvl
        for r in range(4):
            for j in range(2):
                pre = f"Decoder.{lvl}.layers.{r}.main.{j}.main.0"
                s[pre + ".W"] = (c, c, 9)
                s[pre + ".D"] = (c, 9, 9)
                s[pre + ".D_diag"] = (c, 9, 9)
    s["ConvsOut.0.main.0.W"] = (t, base // 4, 1)
    for i in range(3):
        cin = base << (i + 1)
        s[f"UP.{i}.pos"] = (1, 1, h >> i, w >> i)
        s[f"UP.{i}.proj.weight"] = (cin // 2, cin, 1, 1)
        s[f"UP.{i}.proj.bias"] = (cin // 2,)
    s["Convsin.0.main.0.W"] = (base, t // 4, 9)
    s["Convsin.0.main.0.D"] = (t, 9, 9)
    s["Convsin.0.main.0.D_diag"] = (t, 9, 9)
    return s


D2D_SPEC = [(0, 16, 64), (2, 64, 128), (4, 128, 256), (6, 256, 256), (8, 256, 1)]
D3D_SPEC = [(0, 1, 32, 3), (2, 32, 64, 3), (4, 64, 128, 3), (6, 128, 128, 3), (8, 128, 1, 1)]


def discriminator_shapes(t: int = 16) -> Dict[str, Tuple[int, ...]]:
    """state_dict keys/shapes of P2IDiscriminator (p2igan.py:120-145)."""
    s: Dict[str, Tuple[int, ...]] = {"alpha2d": (), "alpha3d": ()}
    for i, cin, cout in D2D_SPEC:
        cin = t if i == 0 else cin
        s[f"d2d.{i}.bias"] = (cout,)
        s[f"d2d.{i}.weight_orig"] = (cout, cin, 3, 3)
        s[f"d2d.{i}.weight_u"] = (cout,)
        s[f"d2d.{i}.weight_v"] = (cin * 9,)
    for i, cin, cout, k in D3D_SPEC:
        s[f"d3d.{i}.bias"] = (cout,)
        s[f"d3d.{i}.weight_orig"] = (cout, cin, k, k, k)
        s[f"d3d.{i}.weight_u"] = (cout,)
        s[f"d3d.{i}.weight_v"] = (cin * k ** 3,)
    return s


def seeded_generator_state(h: int, w: int, seed: int = 2024, mode: str = "test", t: int = 16) -> Dict[str, torch.Tensor]:
    test = mode == "test"
    out = {}
    for k, shp in generator_shapes(h, w, t).items():
        if k.endswith("D_diag"):
            out[k] = torch.eye(9).reshape(1, 9, 9).repeat(shp[0], 1, 1)
        elif k.endswith(".D"):
            out[k] = _normal(seed, k, shp, 0.05) if test else torch.zeros(shp)
        elif k.endswith(".W"):
            out[k] = _uniform(seed, k, shp, 1.0 / math.sqrt(shp[1] * shp[2]))
        elif k.endswith("pos"):
            out[k] = _normal(seed, k, shp, 0.5) if test else torch.zeros(shp)
        elif k.endswith("bias"):
            out[k] = _normal(seed, k, shp, 0.05) if test else torch.zeros(shp)
        else:  # conv weights, kaiming normal fan_in
            fan_in = int(np.prod(shp[1:]))
            out[k] = _normal(seed, k, shp, math.sqrt(2.0 / fan_in))
    return out


def seeded_discriminator_state(seed: int = 2024, mode: str = "test", t: int = 16) -> Dict[str, torch.Tensor]:
    test = mode == "test"
    out = {}
    for k, shp in discriminator_shapes(t).items():
        if k.startswith("alpha"):
            out[k] = torch.tensor(0.3 if (test and k == "alpha2d") else 0.0)
        elif k.endswith("bias"):
            out[k] = _normal(seed, k, shp, 0.05) if test else torch.zeros(shp)
        elif k.endswith("weight_orig"):
            fan_in = int(np.prod(shp[1:]))
            gain = math.sqrt(2.0 / (1 + 0.2 ** 2))
            out[k] = _normal(seed, k, shp, gain / math.sqrt(fan_in))
        else:  # weight_u / weight_v: unit vectors like torch's normalize(randn)
            out[k] = _unit(seed, k, shp[0])
    return out


def synthetic_event(t: int, h: int, w: int, seed: int = 2024) -> torch.Tensor:
    """SURVEY §8d synthetic rain-like event: uint8 noise smoothed by a 5x5 box blur, (T,H,W) uint8."""
    raw = _rng(seed, f"event{t}x{h}x{w}").integers(0, 256, (t, h + 4, w + 4)).astype(np.float32)
    cs = np.cumsum(np.cumsum(np.pad(raw, ((0, 0), (1, 0), (1, 0))), axis=1), axis=2)
    box = (cs[:, 5:, 5:] - cs[:, :-5, 5:] - cs[:, 5:, :-5] + cs[:, :-5, :-5]) / 25.0
    # stretch contrast so values span most of [0,255] and vary in time
    box = (box - box.min()) / max(box.max() - box.min(), 1e-6) * 255.0
    return torch.from_numpy(np.clip(np.rint(box), 0, 255).astype(np.uint8))


def gauge_mask(h: int, w: int, n_points: int, seed: int = 2024) -> torch.Tensor:
    """'stis'-style mask (sti_dataset.py:104-117): (H,W) float 0/1 with exactly n_points ones."""
    perm = _rng(seed, f"gauge{h}x{w}").permutation(h * w)[:n_points]
    m = np.zeros(h * w, dtype=np.float32)
    m[perm] = 1.0
    return torch.from_numpy(m.reshape(h, w))


def block_mask(h: int, w: int, block: int, seed: int = 2024) -> torch.Tensor:
    """'sti' mask semantics (sti_dataset.py:37-62): one random pixel per block x block cell."""
    rng = _rng(seed, f"block{h}x{w}x{block}")
    m = np.zeros((h, w), dtype=np.float32)
    for y0 in range(0, h, block):
        for x0 in range(0, w, block):
            m[rng.integers(y0, min(y0 + block, h)), rng.integers(x0, min(x0 + block, w))] = 1.0
    return torch.from_numpy(m)


def synthetic_batch(b: int, t: int, h: int, w: int, mask_hw: torch.Tensor, seed: int = 2024):
    """(frames, masked, masks) each (B,T,1,H,W) fp32 as Trainer._prepare_batch delivers (train.py:468-473)."""
    fr = torch.stack([synthetic_event(t, h, w, seed + i).float() / 255.0 for i in range(b)])
    frames = fr.unsqueeze(2)
    masks = mask_hw.reshape(1, 1, 1, h, w).expand(b, t, 1, h, w).contiguous()
    return frames.contiguous(), (frames * masks).contiguous(), masks


def metric_fields(seed: int, n: int = 2, t: int = 16, h: int = 32, w: int = 32):
    """(preds, target), each (n,t,1,h,w): smooth fields on the 0..64 normalised rain scale, so that the rain-rate
    transform 10^(x/16)*0.036 of metrics/metric.py crosses its 0.5 / 2 / 4 / 8 mm/h thresholds."""
    g = torch.Generator().manual_seed(seed)
    base = torch.nn.functional.avg_pool2d(torch.rand(n * t, 1, h + 4, w + 4, generator=g), 5, 1) * 110.0 - 25.0
    noise = torch.randn(n * t, 1, h, w, generator=g) * 4.0
    target = base.clamp(min=0.0)
    preds = (base + noise).clamp(min=0.0)
    return preds.reshape(n, t, 1, h, w), target.reshape(n, t, 1, h, w)
