"""Evaluation suite of the reference (p2igan_bench/metrics/metric.py:16-229) on the HIP path: same classes, constructor
arguments, update/compute/reset/to surface and result keys.  State lives in device tensors and is only read back in
compute(); one pass over (preds, target) feeds the regression sums, the contingency tables and the bit plane FSS
box-sums.  Not built: SSIM (torchmetrics' StructuralSimilarityIndexMeasure is a third-party algorithm that is absent
here, so there is nothing to pin it against): compute() has no "ssim" key."""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Dict, Optional, Sequence

import torch

from .. import _hip

EPS = 1e-10


def transform(output):
    """Normalised value -> rainfall intensity (metric.py:16-20)."""
    if isinstance(output, torch.Tensor):
        return torch.pow(10.0, output * 0.0625) * 0.036
    return (10.0 ** (output * 0.0625)) * 0.036


def _prep(preds, target):
    if preds.shape != target.shape or preds.dim() < 2:
        raise RuntimeError(f"metrics: preds {tuple(preds.shape)} and target {tuple(target.shape)} must match (..., H, W)")
    if not preds.is_cuda:
        raise RuntimeError("metrics run on the HIP path (no CPU fallback)")
    from .. import ops
    ops.det_ready(preds.device)          # fixed-order sums (run-to-run reproducible MAE / RMSE / FSS): include/p2i_hip.h, p2i_det_workspace
    return preds.detach().float().contiguous(), target.detach().float().contiguous()


class _DeviceState:
    device: Optional[torch.device] = None

    def to(self, device):
        self.device = torch.device(device)
        self.reset()
        return self


class RegressionMetrics(_DeviceState):
    """MAE / RMSE (metric.py:28-75)."""

    def __init__(self, apply_transform: bool = True, data_range: float = 1.0):
        self.apply_transform = apply_transform
        self.sums = None
        self.n_obs = 0

    def reset(self):
        self.sums = torch.zeros(2, device=self.device) if self.device is not None else None
        self.n_obs = 0

    def update(self, preds, target, _bits=None):
        p, t = _prep(preds, target)
        if self.sums is None:
            self.to(p.device)
        lib = _hip.load()
        _hip.check(lib.p2i_metrics_pointwise(p.data_ptr(), t.data_ptr(), p.numel(), None, 0, int(self.apply_transform),
                                             self.sums.data_ptr(), None, None, torch.cuda.current_stream().cuda_stream),
                   "p2i_metrics_pointwise")
        self.n_obs += p.numel()

    def compute(self) -> Dict[str, float]:
        s = self.sums.double().cpu()
        n = max(float(self.n_obs), 1.0)
        return {"mae": float(s[0] / n), "rmse": float((s[1] / n).sqrt())}


class CategoricalMetrics(_DeviceState):
    """POD / FAR / CSI / HSS per threshold (metric.py:78-136)."""

    def __init__(self, thresholds: Sequence[float]):
        self.thresholds = [float(t) for t in thresholds]
        self.counts = None

    def reset(self):
        self.counts = torch.zeros(len(self.thresholds) * 4, device=self.device, dtype=torch.int64) if self.device is not None else None

    def update(self, preds, target, bits: Optional[torch.Tensor] = None):
        p, t = _prep(preds, target)
        if self.counts is None:
            self.to(p.device)
        lib = _hip.load()
        thr = (ctypes.c_float * len(self.thresholds))(*self.thresholds)
        scratch = torch.zeros(2, device=p.device)
        _hip.check(lib.p2i_metrics_pointwise(p.data_ptr(), t.data_ptr(), p.numel(), thr, len(self.thresholds), 1, scratch.data_ptr(),
                                             self.counts.data_ptr(), None if bits is None else bits.data_ptr(),
                                             torch.cuda.current_stream().cuda_stream), "p2i_metrics_pointwise")

    def compute(self) -> Dict[str, float]:
        c = self.counts.double().cpu().view(-1, 4)
        out: Dict[str, float] = {}
        for thr, (hits, misses, false, correct) in zip(self.thresholds, c):
            pod = hits / (hits + misses + EPS)
            far = false / (hits + false + EPS)
            csi = hits / (hits + misses + false + EPS)
            denom = (misses + false) * (false + correct) + (hits + misses) * (misses + correct)
            hss = 2 * (hits * correct - misses * false) / (denom + EPS)
            prefix = f"cat_thr{thr:.2f}"
            out[f"{prefix}/pod"], out[f"{prefix}/far"], out[f"{prefix}/csi"], out[f"{prefix}/hss"] = float(pod), float(far), float(csi), float(hss)
        return out


class FractionalSkillScoreMetric(_DeviceState):
    """FSS per threshold and scale, averaged over update() calls (metric.py:139-187)."""

    def __init__(self, thresholds: Sequence[float], scales: Sequence[int]):
        self.thresholds = [float(t) for t in thresholds]
        self.scales = [int(s) for s in scales]
        self.score_sum = None
        self.counts = 0

    def reset(self):
        self.score_sum = torch.zeros(len(self.thresholds), len(self.scales), device=self.device) if self.device is not None else None
        self.counts = 0

    def update(self, preds, target, bits: Optional[torch.Tensor] = None):
        p, t = _prep(preds, target)
        if self.score_sum is None:
            self.to(p.device)
        lib = _hip.load()
        H, W = p.shape[-2], p.shape[-1]
        N = p.numel() // (H * W)
        nt, ns = len(self.thresholds), len(self.scales)
        stream = torch.cuda.current_stream().cuda_stream
        if bits is None:
            bits = torch.empty(p.numel(), device=p.device, dtype=torch.uint8)
            thr = (ctypes.c_float * nt)(*self.thresholds)
            scratch = torch.zeros(2, device=p.device)
            cnt = torch.zeros(nt * 4, device=p.device, dtype=torch.int64)
            _hip.check(lib.p2i_metrics_pointwise(p.data_ptr(), t.data_ptr(), p.numel(), thr, nt, 1, scratch.data_ptr(), cnt.data_ptr(),
                                                 bits.data_ptr(), stream), "p2i_metrics_pointwise")
        nd = torch.zeros(2, nt, ns, device=p.device)
        sc = (ctypes.c_int * ns)(*self.scales)
        _hip.check(lib.p2i_metrics_fss(bits.data_ptr(), N, H, W, nt, sc, ns, nd[0].data_ptr(), nd[1].data_ptr(), stream), "p2i_metrics_fss")
        elems = torch.tensor([float(N * (H + 2 * (s // 2) - s + 1) * (W + 2 * (s // 2) - s + 1)) for s in self.scales], device=p.device)
        self.score_sum += 1.0 - (nd[0] / elems) / (nd[1] / elems + EPS)
        self.counts += 1

    def compute(self) -> Dict[str, float]:
        out: Dict[str, float] = {}
        if self.counts == 0:
            return out
        s = (self.score_sum / self.counts).cpu()
        for ti, thr in enumerate(self.thresholds):
            for si, scale in enumerate(self.scales):
                out[f"fss_thr{thr:.2f}_s{scale}"] = float(s[ti, si])
        return out


@dataclass
class MetricConfig:
    thresholds: Sequence[float] = (0.5, 2.0, 4.0, 8.0)
    scales: Sequence[int] = (1, 2, 4, 8)
    apply_transform: bool = True
    data_range: float = 1.0


class RainfallMetricSuite:
    """Bundle of the three metric groups (metric.py:199-229); the categorical pass also emits the threshold bit plane
    that the FSS pass consumes, so preds/target are read twice per update (regression + categorical), not three times."""

    def __init__(self, config: Optional[MetricConfig] = None):
        cfg = config or MetricConfig()
        self.regression = RegressionMetrics(apply_transform=cfg.apply_transform, data_range=cfg.data_range)
        self.categorical = CategoricalMetrics(cfg.thresholds)
        self.fss = FractionalSkillScoreMetric(cfg.thresholds, cfg.scales)
        self.device: Optional[torch.device] = None

    def to(self, device):
        self.device = torch.device(device)
        for m in (self.regression, self.categorical, self.fss):
            m.to(self.device)
        return self

    def update(self, preds, target) -> None:
        p, t = _prep(preds, target)
        if self.device is None:
            self.to(p.device)
        bits = torch.empty(p.numel(), device=p.device, dtype=torch.uint8)
        self.regression.update(p, t)
        self.categorical.update(p, t, bits=bits)
        self.fss.update(p, t, bits=bits)

    def compute(self) -> Dict[str, float]:
        out: Dict[str, float] = {}
        out.update(self.regression.compute())
        out.update(self.categorical.compute())
        out.update(self.fss.compute())
        return out

    def reset(self) -> None:
        for m in (self.regression, self.categorical, self.fss):
            m.reset()


__all__ = ["transform", "RegressionMetrics", "CategoricalMetrics", "FractionalSkillScoreMetric", "RainfallMetricSuite", "MetricConfig"]
