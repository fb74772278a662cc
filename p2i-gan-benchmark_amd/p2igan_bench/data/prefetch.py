"""Host -> HBM hand-over of training batches, one batch ahead of the train step -- on the launching thread.

At 8.2 k frames/s a rank consumes 514 samples/s and the step keeps its launching thread busy for ~9 of its 15 ms (~800 kernel
launches).  The per-sample work (window read, crop, mask draw) therefore lives in DataLoader WORKER PROCESSES (the reference's
num_workers = 4); what is left for the training process is taking the collated uint8 batch out of the worker queue and
enqueueing the H2D copies plus the `p2i_assemble_batch` kernel (/255, video * mask, channel permute:
sti_dataset.py:209,223-224 + train.py:468-473) on a copy stream, one batch AHEAD of the step that consumes it, so the transfer
runs beside the previous step's kernels and the training stream only waits on an event.

No pinned staging: hipHostMalloc'ed memory is fine-grained (uncached for the CPU) on this platform, and filling a pinned buffer
cost 17 ms per 4 MB batch on the GPU box (tools/feed_probe.py: prefetcher with pinned staging 18.3 ms per batch, raw loader 1.0 ms;
DataLoader(pin_memory=True) showed the same: 4 671 vs 7 201 samples/s); the pageable copy is staged by the runtime in ~0.3 ms.
No helper thread: measured on the GPU box (tools/feed_probe.py, profiles/README.md), a Python thread that merely drains the
DataLoader beside the launching thread DOUBLES the step (15.1 -> 29 ms: every one of the ~800 ctypes launches re-acquires the GIL
against it), while the same hand-over in line costs nothing measurable (15.09 ms loader-fed vs 15.09 ms resident).
"""
from __future__ import annotations

from typing import Iterable, Iterator

import torch

from .. import ops


class DevicePrefetcher:
    """for frames, masked, masks in DevicePrefetcher(loader, device): ...   (each (B,T,1,H,W) fp32 on `device`).
    `loader` yields (uint8 frames (B,T,H,W), uint8 masks) pairs (train.device_assemble) or the reference's fp32
    (video, masked, mask) triples of shape (B,T,H,W,1), which are permuted on the device."""

    def __init__(self, loader: Iterable, device: torch.device):
        self.loader, self.device = loader, device
        self.stream = torch.cuda.Stream(device)

    def __len__(self):
        return len(self.loader)

    def _enqueue(self, batch):
        """host batch -> (device tensors, ready event) on the copy stream."""
        with torch.cuda.stream(self.stream):
            dev = [t.to(self.device, non_blocking=True) for t in batch]
            if len(dev) == 2:
                out = list(ops.assemble_batch(dev[0].contiguous(), dev[1].contiguous()))
            else:
                out = [t.permute(0, 1, 4, 2, 3).contiguous() for t in dev]
            ready = torch.cuda.Event()
            ready.record()
        return out, ready

    def __iter__(self) -> Iterator[list]:
        pending = None
        for batch in self.loader:
            nxt = self._enqueue(batch)                       # batch k+1 goes out before step k is launched
            if pending is not None:
                yield self._hand_over(*pending)
            pending = nxt
        if pending is not None:
            yield self._hand_over(*pending)

    def _hand_over(self, out, ready):
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ready)
        for t in out:
            t.record_stream(cur)                             # allocated on the copy stream, consumed on the training stream
        return out
