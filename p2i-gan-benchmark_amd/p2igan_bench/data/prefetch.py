"""Host -> HBM hand-over of training batches, one batch ahead of the train step.

At 8.2 k frames/s a rank consumes 514 samples/s; the step itself keeps the launching thread busy for ~9 ms of its 15.5 ms
(~800 kernel launches), so everything else has to happen beside it: the per-sample work (window read, crop, mask draw) runs in
DataLoader worker processes, and this prefetcher's helper thread takes the collated uint8 batch, stages it in PERSISTENT pinned
buffers (a fresh cudaHostAlloc per batch, which DataLoader(pin_memory=True) does, costs more than the copy), issues the H2D copies
and the `p2i_assemble_batch` kernel (/255, video * mask, channel permute: sti_dataset.py:209,223-224 + train.py:468-473) on a copy
stream and hands the three fp32 tensors over with an event.  The training stream never waits on the host.
"""
from __future__ import annotations

import queue
import threading
from typing import Iterable, Iterator, List, Optional

import torch

from .. import ops


class DevicePrefetcher:
    """for frames, masked, masks in DevicePrefetcher(loader, device): ...   (each (B,T,1,H,W) fp32 on `device`).
    `loader` yields (uint8 frames (B,T,H,W), uint8 masks) pairs (train.device_assemble) or the reference's fp32
    (video, masked, mask) triples of shape (B,T,H,W,1), which are permuted on the device."""

    def __init__(self, loader: Iterable, device: torch.device, depth: int = 2):
        self.loader, self.device, self.depth = loader, device, max(1, depth)
        self.stream = torch.cuda.Stream(device)
        self._stage: List[Optional[list]] = [None] * (self.depth + 1)          # pinned staging buffers per slot
        self._copied: List[Optional[torch.cuda.Event]] = [None] * (self.depth + 1)

    def __len__(self):
        return len(self.loader)

    def _pinned(self, slot: int, batch) -> list:
        bufs = self._stage[slot]
        if bufs is None or len(bufs) != len(batch) or any(b.shape != t.shape or b.dtype != t.dtype for b, t in zip(bufs, batch)):
            bufs = self._stage[slot] = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in batch]
        if self._copied[slot] is not None:
            self._copied[slot].synchronize()                 # the H2D copy that last read these buffers has finished
        for b, t in zip(bufs, batch):
            b.copy_(t)
        return bufs

    def _produce(self, q: "queue.Queue", stop: threading.Event):
        try:
            torch.cuda.set_device(self.device)
            slot = 0
            for batch in self.loader:
                if stop.is_set():
                    break
                host = self._pinned(slot, list(batch))
                with torch.cuda.stream(self.stream):
                    dev = [t.to(self.device, non_blocking=True) for t in host]
                    ev = torch.cuda.Event()
                    ev.record()
                    self._copied[slot] = ev
                    if len(dev) == 2:
                        out = list(ops.assemble_batch(dev[0].contiguous(), dev[1].contiguous()))
                    else:
                        out = [t.permute(0, 1, 4, 2, 3).contiguous() for t in dev]
                    ready = torch.cuda.Event()
                    ready.record()
                q.put((out, ready))
                slot = (slot + 1) % len(self._stage)
            q.put(None)
        except BaseException as e:                            # surfaces in the consumer
            q.put(e)

    def __iter__(self) -> Iterator[list]:
        q: "queue.Queue" = queue.Queue(maxsize=self.depth)
        stop = threading.Event()
        th = threading.Thread(target=self._produce, args=(q, stop), daemon=True, name="p2i-prefetch")
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                out, ready = item
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ready)
                for t in out:
                    t.record_stream(cur)                     # allocated on the copy stream, consumed on the training stream
                yield out
        finally:
            stop.set()
            while th.is_alive():                             # unblock a producer waiting on the full queue
                try:
                    q.get_nowait()
                except queue.Empty:
                    pass
                th.join(timeout=0.05)
