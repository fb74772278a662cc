"""Host -> HBM hand-over of training batches, one batch ahead of the train step -- on the launching thread.

At 8.2 k frames/s a rank consumes 514 samples/s and the step keeps its launching thread busy for ~9 of its 15 ms (~800 kernel
launches).  The per-sample work (window read, crop, mask draw) therefore lives in DataLoader WORKER PROCESSES (the reference's
num_workers = 4); what is left for the training process is taking the collated uint8 batch out of the worker queue, staging it
in PERSISTENT pinned buffers (a fresh pinned allocation per batch, which DataLoader(pin_memory=True) does, costs more than the
copy), and enqueueing the H2D copies plus the `p2i_assemble_batch` kernel (/255, video * mask, channel permute:
sti_dataset.py:209,223-224 + train.py:468-473) on a copy stream, one batch AHEAD of the step that consumes it, so the transfer
runs beside the previous step's kernels and the training stream only waits on an event.

No helper thread: measured on the GPU box (tools/feed_probe.py, profiles/README.md), a Python thread that merely drains the
DataLoader beside the launching thread DOUBLES the step (15.1 -> 29 ms: every one of the ~800 ctypes launches re-acquires the GIL
against it), while the same hand-over in line costs nothing measurable (15.09 ms loader-fed vs 15.09 ms resident).
"""
from __future__ import annotations

from typing import Iterable, Iterator, List, Optional

import torch

from .. import ops


class DevicePrefetcher:
    """for frames, masked, masks in DevicePrefetcher(loader, device): ...   (each (B,T,1,H,W) fp32 on `device`).
    `loader` yields (uint8 frames (B,T,H,W), uint8 masks) pairs (train.device_assemble) or the reference's fp32
    (video, masked, mask) triples of shape (B,T,H,W,1), which are permuted on the device."""

    SLOTS = 3            # pinned staging sets: one being filled, one in flight, one whose copy is surely finished

    def __init__(self, loader: Iterable, device: torch.device):
        self.loader, self.device = loader, device
        self.stream = torch.cuda.Stream(device)
        self._stage: List[Optional[list]] = [None] * self.SLOTS
        self._copied: List[Optional[torch.cuda.Event]] = [None] * self.SLOTS
        self._slot = 0

    def __len__(self):
        return len(self.loader)

    def _enqueue(self, batch):
        """host batch -> (device tensors, ready event); everything asynchronous on the copy stream."""
        slot = self._slot
        self._slot = (slot + 1) % self.SLOTS
        batch = list(batch)
        bufs = self._stage[slot]
        if bufs is None or len(bufs) != len(batch) or any(b.shape != t.shape or b.dtype != t.dtype for b, t in zip(bufs, batch)):
            bufs = self._stage[slot] = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in batch]
        if self._copied[slot] is not None:
            self._copied[slot].synchronize()                 # the H2D copy that last read these buffers (3 batches ago) has finished
        for b, t in zip(bufs, batch):
            b.copy_(t)
        with torch.cuda.stream(self.stream):
            dev = [t.to(self.device, non_blocking=True) for t in bufs]
            ev = torch.cuda.Event()
            ev.record()
            self._copied[slot] = ev
            if len(dev) == 2:
                out = list(ops.assemble_batch(dev[0], dev[1]))
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
