"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's evaluation metrics (p2igan_bench/metrics/metric.py).

Only tests/ may import this file.  Each function cites the reference lines it follows.  Pinned by
tests/golden/metrics_32.npz, captured by tests/golden/make_golden.py from the genuine reference classes
(RegressionMetrics / CategoricalMetrics / FractionalSkillScoreMetric; torchmetrics' Metric base and SSIM are stubbed
there because torchmetrics is absent: SSIM is therefore "parity unpinned" and not restated)."""
import torch
import torch.nn.functional as F

EPS = 1e-10


def transform(x):                                     # metric.py:16-20
    return torch.pow(10.0, x * 0.0625) * 0.036


def regression(preds, target, apply_transform=True):  # metric.py:42-52, 66-69
    p, t = preds.float(), target.float()
    if apply_transform:
        p, t = transform(p), transform(t)
    d = p - t
    n = max(d.numel(), 1)
    return {"mae": float(d.abs().sum() / n), "rmse": float(torch.sqrt((d ** 2).sum() / n))}


def contingency(preds, target, thresholds):           # metric.py:92-111  -> (nt, 4) hits, misses, false alarms, correct negatives
    p, t = transform(preds.float()).reshape(1, -1), transform(target.float()).reshape(1, -1)
    thr = torch.tensor(thresholds, dtype=torch.float32).view(-1, 1)
    tp, tt = p >= thr, t >= thr
    return torch.stack([(tp & tt).sum(1), (~tp & tt).sum(1), (tp & ~tt).sum(1), (~tp & ~tt).sum(1)], dim=1)


def categorical_scores(table, thresholds):            # metric.py:113-136
    out = {}
    for thr, (hits, misses, false, correct) in zip(thresholds, table.double()):
        denom = (misses + false) * (false + correct) + (hits + misses) * (misses + correct)
        pre = f"cat_thr{float(thr):.2f}"
        out[f"{pre}/pod"] = float(hits / (hits + misses + EPS))
        out[f"{pre}/far"] = float(false / (hits + false + EPS))
        out[f"{pre}/csi"] = float(hits / (hits + misses + false + EPS))
        out[f"{pre}/hss"] = float(2 * (hits * correct - misses * false) / (denom + EPS))
    return out


def fss(preds, target, thresholds, scales):           # metric.py:152-175  -> (nt, ns) score of ONE update call
    h, w = preds.shape[-2], preds.shape[-1]
    p = transform(preds.float()).reshape(-1, 1, h, w)
    t = transform(target.float()).reshape(-1, 1, h, w)
    out = torch.zeros(len(thresholds), len(scales))
    for ti, thr in enumerate(thresholds):
        pm, tm = (p >= thr).float(), (t >= thr).float()
        for si, s in enumerate(scales):
            fp = F.avg_pool2d(pm, kernel_size=int(s), stride=1, padding=int(s) // 2)
            ft = F.avg_pool2d(tm, kernel_size=int(s), stride=1, padding=int(s) // 2)
            out[ti, si] = 1.0 - torch.mean((fp - ft) ** 2) / (torch.mean(fp ** 2 + ft ** 2) + EPS)
    return out
