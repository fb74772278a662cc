"""Data-parallel plumbing: one process per GPU, batch sharded by rank, one flat-bucket
gradient all-reduce per network per optimiser step over RCCL/xGMI (backend "nccl" on ROCm; "gloo" in the
CPU tests).  No data-path collective exists besides that exchange: samples are independent (no BatchNorm
on the P2I path, IDW is per-sample, losses are batch means)."""
from __future__ import annotations

import os
from typing import Iterator, List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn


def init_distributed(backend: Optional[str] = None) -> tuple:
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torch.distributed.run contract). Returns (rank, world, local)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


class FlatParams:
    """Re-homes the trainable parameters of `module` into ONE contiguous fp32 buffer (+ a flat gradient
    buffer whose views are installed as .grad), so the optimiser is one fused launch and the DP exchange
    one all-reduce."""

    def __init__(self, module: nn.Module):
        self.module = module
        self.params: List[nn.Parameter] = [p for p in module.parameters() if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.empty(n, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(n, device=dev, dtype=torch.float32)
        off = 0
        for p in self.params:
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view(p.shape)
            p.grad = self.grad[off:off + k].view(p.shape)
            off += k
        self.n = n

    def zero_grad(self):
        if self.grad.is_cuda:
            from . import ops           # p2i_zero (hipMemsetAsync): no ATen fill kernel on the step
            ops.zero_(self.grad)
        else:
            self.grad.zero_()
        off = 0
        for p in self.params:       # autograd accumulates in place; re-attach if something replaced .grad
            k = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * off:
                p.grad = self.grad[off:off + k].view(p.shape)
            off += k


def allreduce_mean_(buf: torch.Tensor, world: int, scale_fn=None):
    """Sum over ranks then 1/world (losses are per-rank batch means => averaged gradients)."""
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    if scale_fn is not None:
        scale_fn(buf, 1.0 / world)
    else:
        buf.mul_(1.0 / world)


class BucketedAllReduce:
    """The flat gradient buffer exchanged as contiguous BUCKETS, each launched asynchronously the moment the backward pass has
    finished writing it, so that the exchange of the early buckets runs beside the rest of the backward (the generator's backward
    completes Decoder level 0 first and level 3 -- 74 % of the bytes -- last).  launch(lo, hi) is called from the stream that wrote
    grad[lo:hi] (the collective is ordered behind everything enqueued there so far: RCCL waits on an event of the current
    stream and runs on its own stream); finish() launches what was not covered yet, waits for every bucket and applies the 1/world
    mean.  Two ranks: bit-identical to one flat all-reduce (a sum of two addends has no order); more ranks: equal to rounding."""

    def __init__(self, grad: torch.Tensor, world: int, scale_fn=None):
        self.grad, self.world, self.scale_fn = grad, world, scale_fn
        self.works, self.done = [], []

    def launch(self, lo: int, hi: int):
        if hi > lo:
            self.works.append(dist.all_reduce(self.grad[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
            self.done.append((lo, hi))

    def finish(self):
        pos = 0
        for lo, hi in sorted(self.done):            # the gaps between the launched buckets
            if lo > pos:
                self.launch(pos, lo)
            pos = max(pos, hi)
        if pos < self.grad.numel():
            self.launch(pos, self.grad.numel())
        for w in self.works:
            w.wait()                                # the current stream waits for the collective's stream
        self.works, self.done = [], []
        if self.scale_fn is not None:
            self.scale_fn(self.grad, 1.0 / self.world)
        else:
            self.grad.mul_(1.0 / self.world)


def broadcast_module_state(module: nn.Module, flat: Optional[FlatParams] = None, src: int = 0):
    """Rank `src`'s weights, buffers (spectral-norm u/v) and frozen tensors to every rank, once: afterwards
    the deterministic updates keep ranks bit-identical."""
    for t in list(module.buffers()) + [p.data for p in module.parameters() if not p.requires_grad]:
        dist.broadcast(t, src)
    if flat is not None:
        dist.broadcast(flat.flat, src)
    else:
        for p in module.parameters():
            if p.requires_grad:
                dist.broadcast(p.data, src)


class ShardedSampler:
    """Rank-sharded index stream: every rank draws the SAME seeded permutation per epoch and takes the
    slice rank::world of it.  even=True (training): truncated so that all ranks see the same number of samples (every
    step holds a collective).  even=False (evaluation): nothing is dropped and nothing is repeated -- ranks may differ
    by one sample, and the caller all-reduces (sum, sample count)."""

    def __init__(self, n: int, rank: int, world: int, shuffle: bool = True, seed: int = 0, even: bool = True):
        self.n, self.rank, self.world, self.shuffle, self.seed, self.epoch, self.even = n, rank, world, shuffle, seed, 0, even

    def set_epoch(self, epoch: int):
        self.epoch = epoch

    def __len__(self):
        if self.even:
            return self.n // self.world
        return len(range(self.rank, self.n, self.world))

    def __iter__(self) -> Iterator[int]:
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            idx = torch.randperm(self.n, generator=g).tolist()
        else:
            idx = list(range(self.n))
        if not self.even:
            return iter(idx[self.rank::self.world])
        per = self.n // self.world
        return iter(idx[self.rank:per * self.world:self.world])
