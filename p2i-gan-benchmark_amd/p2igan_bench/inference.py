"""Sliding-window inference of infer.py:188-262 on the device: window `stride` (16) every
`stride - overlap` (4) frames, the last windows padded by repeating the final frame, overlap-averaged,
scaled and clipped at 0.  All windows of an event go through the generator as ONE batch (the
reference runs them one by one at B=1) and the overlap-average is accumulated on the device."""
from __future__ import annotations

import torch


@torch.no_grad()
def infer_event(generator, masked: torch.Tensor, masks: torch.Tensor, stride: int = 16, overlap: int = 12,
                output_scale: float = 255.0, max_windows_per_batch: int = 32) -> torch.Tensor:
    """masked, masks: (1, L, 1, H, W) on the generator's device -> (L, 1, H, W) float32.
    HIP tensors (H*W a multiple of 4): windows are gathered and the overlap average / scale / clip is taken by two bandwidth kernels
    (p2i_window_gather, p2i_window_mean) -- no index tensors, no boolean-mask indexing (a device-to-host sync in ATen).  CPU tensors
    (drop-in users, tests): the same arithmetic in torch."""
    L = masked.shape[1]
    step = max(1, stride - overlap)
    nwin = len(range(0, L, step))
    hw = masked.shape[-2] * masked.shape[-1]
    if masked.is_cuda and hw % 4 == 0:
        from . import ops
        m0, k0 = masked[0].contiguous().float(), masks[0].contiguous().float()
        preds = torch.empty((nwin, stride) + tuple(masked.shape[2:]), device=masked.device, dtype=torch.float32)
        for s in range(0, nwin, max_windows_per_batch):
            nw = min(max_windows_per_batch, nwin - s)
            wm, wk = ops.window_gather(m0, k0, L, s, nw, stride, step)
            preds[s:s + nw] = generator(wm, wk)                  # (nw, stride, 1, H, W)
        return ops.window_mean(preds, L, stride, step, output_scale)
    starts = list(range(0, L, step))
    idx = torch.arange(stride, device=masked.device).unsqueeze(0) + torch.tensor(starts, device=masked.device).unsqueeze(1)
    valid = idx < L                                             # (nwin, stride)
    idx = idx.clamp(max=L - 1)                                  # repeat the last frame (infer.py:219-227)
    acc = torch.zeros(L, *masked.shape[2:], device=masked.device)
    cnt = torch.zeros(L, device=masked.device)
    for s in range(0, len(starts), max_windows_per_batch):
        ii = idx[s:s + max_windows_per_batch]
        out = generator(masked[0][ii].contiguous(), masks[0][ii].contiguous())      # (nw, stride, 1, H, W)
        vv = valid[s:s + max_windows_per_batch]
        acc.index_add_(0, ii[vv], out[vv])
        cnt.index_add_(0, ii[vv], torch.ones_like(ii[vv], dtype=torch.float32))
    comp = acc / cnt.clamp(min=1e-5).view(L, 1, 1, 1) * float(output_scale)
    return comp.clamp(min=0.0)
