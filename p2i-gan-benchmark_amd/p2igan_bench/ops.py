"""Raw (non-autograd) operators over the C ABI of libp2i_hip.so.

Every function takes/returns contiguous fp32 CUDA tensors, validates shapes on the host BEFORE
any launch (a faulting kernel can reset the node), enqueues on torch's current stream and
raises RuntimeError on failure.  No CPU path exists here by design.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import _hip
from ._hip import ACT_LEAKY, ACT_NONE, ACT_RELU, ACT_TANH, ConvDesc  # noqa: F401


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


# HIP stream handle of the calling step.  torch.cuda.current_stream() costs ~1.5 us and a train step asks ~800 times; TrainEngine
# pins the handle for the duration of a step (step_stream), SideStream.run swaps in its own.  None: ask torch (every other caller).
_CUR_STREAM = None


def _stream():
    return _CUR_STREAM if _CUR_STREAM is not None else torch.cuda.current_stream().cuda_stream


class step_stream:
    """with ops.step_stream(): ...  -- every op inside enqueues on the stream that is current at entry."""

    def __enter__(self):
        global _CUR_STREAM
        self.prev = _CUR_STREAM
        _CUR_STREAM = torch.cuda.current_stream().cuda_stream
        return self

    def __exit__(self, *exc):
        global _CUR_STREAM
        _CUR_STREAM = self.prev
        return False


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not (t.is_cuda and t.is_contiguous()):
            raise RuntimeError("p2i ops need contiguous CUDA tensors (the HIP path has no CPU fallback)")
        if t.dtype not in (torch.float32, torch.int32, torch.uint8, torch.int16):
            raise RuntimeError(f"unsupported dtype {t.dtype}")


class KernelProfile:
    """Optional live timing of the conv-engine launches with HIP events on the launch stream
    (torch's current stream).  Used by bench.py's instrumented pass for the roofline figure; off
    by default so that the measured throughput is unperturbed."""

    def __init__(self):
        self.records = []            # (kernel key, algorithmic flops, start event, end event)

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for key, flops, e0, e1 in self.records:
            a = agg.setdefault(key, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += e0.elapsed_time(e1) * 1e-3
            a[2] += flops
        return {k: {"launches": v[0], "seconds": v[1], "flops": v[2]} for k, v in agg.items()}


PROFILE: Optional[KernelProfile] = None


def _prof_begin():
    if PROFILE is None:
        return None
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def _prof_end(e0, kind, spec, desc, stride1):
    if e0 is None:
        return
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record()
    flops = 2.0 * desc.B * desc.Cout * desc.Cin * spec.ntaps * desc.To * desc.Ho * desc.Wo
    if kind == "wgrad":
        import ctypes
        wp = (ctypes.c_int * 4)()
        _hip.load().p2i_wgrad_last_plan(wp)
        key = {0: "wgrad_kernel<64>", 2: "c1_wgrad_kernel<32>", 3: "wgrad_x6_kernel (bf16-split x6)"}.get(wp[0]) or \
            "wgrad_dma_kernel<64, %d, %s, %d>%s" % (wp[1], "true" if wp[2] else "false", wp[3], "" if spec.k[0] == 1 else " (x%d kt slices)" % spec.k[0])
    else:
        import ctypes
        plan = (ctypes.c_int * 6)()
        _hip.load().p2i_conv_last_plan(plan)
        if plan[5] > 10:
            key = "patch_gemm_fused_kernel<%d, %d, %d, %d, %d> (strided dgrad, %d parity classes per workgroup)" % (plan[0], plan[1], plan[2], plan[3], plan[5] - 10, plan[5] - 10)
        elif plan[5] == 7 and plan[3] == 4:        # producer / consumer wave roles
            key = "patch_gemm_x6p_kernel<%d, %d, false> (%d x %d tile, %d taps per stage, producer / consumer waves, bf16-split x6)" % (
                plan[0] // 32, plan[4], plan[0], plan[1], plan[4])
        elif plan[5] == 7:
            key = "patch_gemm_x6c_kernel<%d, %d, false, %d> (%d x %d tile, %d taps per stage, bf16-split x6)" % (
                plan[1] // 32, plan[0] // 32, plan[4], plan[0], plan[1], plan[4])
        elif plan[5] == 3:
            key = "o1_fwd_kernel (single output channel, bandwidth-bound)"
        elif plan[5] == 8 and plan[3] == 4:
            key = "patch_gemm_x6p_kernel<1, %d, true> (strided dgrad, 4 parity classes per workgroup, producer / consumer waves, bf16-split x6)" % plan[4]
        elif plan[5] == 8:
            key = "patch_gemm_x6c_kernel<8, 1, true, %d> (strided dgrad, 4 parity classes per workgroup, bf16-split x6)" % plan[4]
        else:
            key = "patch_gemm_dma_kernel<%d, %d, %d, %d, %d, %d>" % tuple(plan)
    PROFILE.records.append((key, flops, e0, e1))


# Convolution engine selection (forward and data gradient).  "auto" (default): the exact-fp32 bf16-split kernel of csrc/conv_x6c.hip
# (3-way bf16 split, six v_mfma_f32_32x32x16_bf16 products) on the layers where it is the faster one (3x3 stride-1 2-D layers with
# >= 200 workgroups: the generator's 64- and 128-channel levels at B=8; measured 63 vs 100 us on the 128-channel level), the
# v_mfma_f32_32x32x2_f32 kernels everywhere else; the choice is made inside the library (x6c_would_take).  "f32": f32 kernels only.
# ("x6" is accepted as an alias of "auto".)  Both are HIP paths.
import os as _os

CONV_ENGINE = _os.environ.get("P2I_CONV_ENGINE", "auto")
_X6_SCRATCH = {}


def _x6_scratch(numel: int, device):
    """uint16 scratch for the split weights of ONE conv call, reused by every call on the same stream (the split
    kernel and its consumer are stream-ordered, so the next call may overwrite it)."""
    if CONV_ENGINE not in ("auto", "x6"):
        return None
    key = (device.index, _stream())
    buf = _X6_SCRATCH.get(key)
    if buf is None or buf.numel() < numel:
        buf = torch.empty(max(numel, 1 << 22), device=device, dtype=torch.int16)
        _X6_SCRATCH[key] = buf
    return buf


class X6Stack:
    """bf16 split of a stack of same-shape packed weights (n, taps, K, Mpad), made ONCE per weight update and on first need: the
    conv calls of its n layers then skip their per-call split (60 launches of ~5 us per step before).  Attached to the packed
    tensors as `wp._x6s = (stack, layer index)` by the batched fold / pack helpers."""

    def __init__(self, wstack):
        self.w = wstack
        self.wb = None
        self.n, self.taps, self.k, self.mpad = wstack.shape

    def layer_ptr(self, i):
        if self.wb is None:                 # first layer that the bf16-split kernel takes: split the whole stack in one launch
            self.wb = torch.empty(3 * self.w.numel(), device=self.w.device, dtype=torch.int16)
            _hip.check(_hip.load().p2i_x6_split(_ptr(self.w), _ptr(self.wb), self.n * self.taps, self.k, self.mpad, _stream()), "p2i_x6_split")
        return self.wb.data_ptr() + 2 * i * self.taps * self.k * self.mpad

    @staticmethod
    def attach(wstack, views):
        """views[i] is the caller's view wstack[i] (the attribute lives on the view OBJECT that is handed to the conv calls)."""
        if CONV_ENGINE in ("auto", "x6") and wstack is not None and wstack.shape[2] % 16 == 0 and wstack.shape[3] % 32 == 0:
            st = X6Stack(wstack)
            for i, v in enumerate(views):
                v._x6s = (st, i)


class _X6One:
    """One tensor's pre-split image (see x6_presplit): quacks like an X6Stack of a single layer."""

    def __init__(self, ptr, taps):
        self.ptr, self.n, self.taps = ptr, 1, taps

    def layer_ptr(self, i):
        return self.ptr


def x6_presplit(tensors, wants, buf=None):
    """bf16 split of several packed weight tensors of DIFFERENT shapes in ONE launch (p2i_x6_split_batched) -- the discriminators'
    layers right after weight_pack_batched; `wants[i]` says whether any conv call of the coming pass would use tensor i's split
    (p2i_x6c_would_take).  Attaches `_x6s` to the tensors like X6Stack.attach does, so that conv_fwd / conv_dgrad skip their
    per-call split (18 launches of ~7 us per train step before).  buf: caller-owned int16 buffer to reuse.  Returns the buffer."""
    if CONV_ENGINE not in ("auto", "x6"):
        return buf
    sel = [t for t, w in zip(tensors, wants) if w and t is not None and t.shape[1] % 16 == 0]
    if not sel:
        return buf
    import ctypes
    need = sum(3 * t.numel() for t in sel)
    if buf is None or buf.numel() < need:
        buf = torch.empty(need, device=sel[0].device, dtype=torch.int16)
    n = len(sel)
    ptrs, off = [], 0
    for t in sel:
        ptrs.append(buf.data_ptr() + 2 * off)
        off += 3 * t.numel()
    arr_i = ctypes.c_int * n
    _hip.check(_hip.load().p2i_x6_split_batched(_ptr_array(sel), (ctypes.c_void_p * n)(*ptrs), arr_i(*[t.shape[0] for t in sel]),
                                                arr_i(*[t.shape[1] for t in sel]), arr_i(*[t.shape[2] for t in sel]), n, _stream()),
               "p2i_x6_split_batched")
    for t, p_ in zip(sel, ptrs):
        t._x6s = (_X6One(p_, t.shape[0]), 0)
        t._x6buf = buf                       # keeps the image alive as long as the packed tensor
    return buf


def _x6s_of(wp, d, dgrad, act=ACT_NONE):
    """(layer pointer, total taps) of a pre-split stack when this call would run on the bf16-split kernel, else None."""
    h = getattr(wp, "_x6s", None)
    if h is None or CONV_ENGINE not in ("auto", "x6") or not _hip.load().p2i_x6c_would_take(d, 1 if dgrad else 0, act):
        return None
    st, i = h
    return st.layer_ptr(i), st.n * st.taps


# Weight-gradient partial tiles: stored to a per-stream scratch and summed by a second kernel (deterministic, and faster than
# 256-way float atomics); P2I_WGRAD_SLICES=0 keeps the atomic path.  256 slices * 9 taps * 64 * 64 floats cover every layer.
WGRAD_SLICES = _os.environ.get("P2I_WGRAD_SLICES", "1") != "0"
_WGRAD_WS = {}


def _wgrad_scratch(device):
    if not WGRAD_SLICES:
        return None
    key = (device.index, _stream())
    buf = _WGRAD_WS.get(key)
    if buf is None:
        buf = torch.empty(256 * (9 * 64 * 64 + 512) + 1024, device=device, dtype=torch.float32)      # slices + their bias rows
        _WGRAD_WS[key] = buf
    return buf


# Depth of SideStream.run calls on the Python stack.  A side stream is only ever forked from the step's ORIGIN stream: forking a
# stream from a stream that is itself a fork (a SideStream used inside another SideStream.run) is legal in the CUDA model but ends
# hipGraph capture in a segmentation fault on ROCm 7.2 -- reproduced with torch tensors only by tools/graph_nested_fork.py (cases
# `nested*`: SIGSEGV; `flat*`: fine; gpurun_out/r4a/nested_fork.log).  models/p2igan.py::_side_of therefore hands out no side stream
# while another one is running (depth > 0): the nested work runs in line on the current side stream, eager and captured alike.
_SIDE_DEPTH = 0


def side_depth() -> int:
    return _SIDE_DEPTH


class SideStream:
    """A second HIP stream for work that nothing on the main stream waits for until a join: the weight-gradient kernels of a
    backward pass (each needs only tensors the data-gradient chain has already produced, and its result is consumed once per
    level).  Run beside the data-gradient chain they fill each other's kernel tails (single-round grids: ~10 us of ramp / drain /
    launch gap per kernel) and hide the wgrad slice-reduce kernels.  Dependencies are HIP events; tensors handed to the side
    stream are kept alive until the join (the caching allocator would otherwise hand their memory to the main stream)."""

    def __init__(self, device):
        self.stream = torch.cuda.Stream(device)
        self.keep = []
        self.pending = False

    def run(self, fn, *tensors, after=None):
        # everything enqueued on the forking stream so far (producers of `tensors`) -> this side stream.  after: a token of
        # mark_stream() instead -- the side work then depends on what the forking stream held at THAT point only (TrainEngine enqueues
        # the step's head on the main stream first and lets the side work start beside it, not behind it)
        if after is None:
            fork(_stream(), self.stream.cuda_stream)
        else:
            _hip.check(_hip.load().p2i_event_wait(after, self.stream.cuda_stream), "p2i_event_wait")
        global _CUR_STREAM, _SIDE_DEPTH
        prev, _CUR_STREAM = _CUR_STREAM, self.stream.cuda_stream
        _SIDE_DEPTH += 1
        try:
            with torch.cuda.stream(self.stream):
                out = fn()
        finally:
            _CUR_STREAM = prev
            _SIDE_DEPTH -= 1
        self.keep.extend(tensors)
        self.pending = True
        return out

    def mark(self):
        """A wait token after everything enqueued on the side stream so far (the main stream can wait for PART of the side work):
        call it -- token() -- on the thread of control whose current stream shall wait."""
        slot = _next_slot()
        _hip.check(_hip.load().p2i_event_record(slot, self.stream.cuda_stream), "p2i_event_record")
        return lambda: _hip.check(_hip.load().p2i_event_wait(slot, _stream()), "p2i_event_wait")

    def join(self):
        if self.pending:
            fork(self.stream.cuda_stream, _stream())
            self.pending = False
        self.keep.clear()


# Cross-stream dependencies go through the library's event table (p2i_event_record / p2i_event_wait: 256 slots of events without
# timing, include/p2i_hip.h) instead of torch.cuda.Event objects: a launch tape (p2i_tape_*) sees them, and a fork costs two ctypes
# calls instead of an event allocation + record + wait through torch.  Slots are handed out round-robin; a step uses ~40.
_SLOT = 0


def _next_slot() -> int:
    global _SLOT
    _SLOT = (_SLOT + 1) % 256
    return _SLOT


def mark_stream() -> int:
    """A token for "everything enqueued on the current launch stream so far" (SideStream.run(..., after=token))."""
    slot = _next_slot()
    _hip.check(_hip.load().p2i_event_record(slot, _stream()), "p2i_event_record")
    return slot


def fork(src_stream: int, dst_stream: int):
    """dst_stream waits for everything enqueued on src_stream so far (raw HIP stream handles)."""
    lib = _hip.load()
    slot = _next_slot()
    _hip.check(lib.p2i_event_record(slot, src_stream), "p2i_event_record")
    _hip.check(lib.p2i_event_wait(slot, dst_stream), "p2i_event_wait")


# Run-to-run reproducibility: the scratch of the library's deterministic reductions (include/p2i_hip.h, p2i_det_workspace), registered
# once per process and device at the first backward op.  P2I_DETERMINISTIC=0 keeps the float atomics (A/B runs).
_DET = {}


def det_ready(device):
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key in _DET:
        return
    if _os.environ.get("P2I_DETERMINISTIC", "1") == "0":
        _DET[key] = None
        return
    mb = int(_os.environ.get("P2I_DET_SCRATCH_MB", "128"))
    part = torch.empty(mb << 18, device=device, dtype=torch.float32)
    cnt = torch.zeros(8192, device=device, dtype=torch.int32)
    torch.cuda.current_stream(device).synchronize()          # the counters are zero before any kernel of any stream can use them
    _hip.check(_hip.load().p2i_det_workspace(part.data_ptr(), part.numel(), cnt.data_ptr(), cnt.numel()), "p2i_det_workspace")
    _DET[key] = (part, cnt)


def pad32(n: int) -> int:
    return (n + 31) // 32 * 32


@dataclass(frozen=True)
class ConvSpec:
    """Static description of one convolution layer (2-D layers have kt = st = 1, pt = 0)."""
    cin: int
    cout: int
    k: Tuple[int, int, int]
    stride: Tuple[int, int, int] = (1, 1, 1)
    pad: Tuple[int, int, int] = (0, 0, 0)

    @property
    def ntaps(self) -> int:
        return self.k[0] * self.k[1] * self.k[2]

    def out_dims(self, t, h, w):
        return tuple((d + 2 * p - k) // s + 1 for d, p, k, s in zip((t, h, w), self.pad, self.k, self.stride))

    def desc(self, b, t, h, w) -> ConvDesc:
        return _conv_desc(self, b, t, h, w)

    def wp_f_shape(self):
        return (self.ntaps, self.cin, pad32(self.cout))

    def wp_d_shape(self):
        return (self.ntaps, self.cout, pad32(self.cin))


import functools


@functools.lru_cache(maxsize=512)
def _conv_desc(spec: ConvSpec, b, t, h, w) -> ConvDesc:
    """The (read-only) descriptor of one layer at one input size: built once, ~3 us of ctypes work per call otherwise."""
    to, ho, wo = spec.out_dims(t, h, w)
    return ConvDesc(b, spec.cin, spec.cout, t, h, w, to, ho, wo, *spec.k, *spec.stride, *spec.pad)


def _dims5(x: torch.Tensor):
    if x.dim() == 4:
        b, c, h, w = x.shape
        return b, c, 1, h, w
    b, c, t, h, w = x.shape
    return b, c, t, h, w


def conv_fwd(spec: ConvSpec, x, wp_f, bias=None, residual=None, act=ACT_NONE, out=None):
    lib = _hip.load()
    b, c, t, h, w = _dims5(x)
    if c != spec.cin or tuple(wp_f.shape) != spec.wp_f_shape():
        raise RuntimeError(f"conv_fwd: shape mismatch x={tuple(x.shape)} wp={tuple(wp_f.shape)} spec={spec}")
    to, ho, wo = spec.out_dims(t, h, w)
    oshape = (b, spec.cout, ho, wo) if x.dim() == 4 else (b, spec.cout, to, ho, wo)
    y = out if out is not None else torch.empty(oshape, device=x.device, dtype=torch.float32)
    if tuple(y.shape) != oshape or (residual is not None and tuple(residual.shape) != oshape):
        raise RuntimeError("conv_fwd: output/residual shape mismatch")
    if bias is not None and bias.numel() != spec.cout:
        raise RuntimeError("conv_fwd: bias size mismatch")
    _chk(x, wp_f, bias, residual, y)
    d = spec.desc(b, t, h, w)
    e0 = _prof_begin()
    pre = _x6s_of(wp_f, d, False, act)
    ws = _x6_scratch(3 * wp_f.numel(), x.device) if (pre is None and spec.cin % 16 == 0) else None
    if pre is not None:
        _hip.check(lib.p2i_conv_fwd_x6s(d, _ptr(x), _ptr(wp_f), pre[0], pre[1], _ptr(bias), _ptr(residual), _ptr(y), act, _stream()),
                   "p2i_conv_fwd_x6s")
    elif ws is not None:
        _hip.check(lib.p2i_conv_fwd_x6(d, _ptr(x), _ptr(wp_f), _ptr(ws), _ptr(bias), _ptr(residual), _ptr(y), act, _stream()),
                   "p2i_conv_fwd_x6")
    else:
        _hip.check(lib.p2i_conv_fwd(d, _ptr(x), _ptr(wp_f), _ptr(bias), _ptr(residual), _ptr(y), act, _stream()), "p2i_conv_fwd")
    _prof_end(e0, "fwd", spec, d, True)
    return y


def conv_dgrad(spec: ConvSpec, dy, wp_d, in_shape, y_act=None, act=ACT_NONE, add=None, mask_y=None, mask_act=ACT_NONE):
    """dx for input of shape in_shape; if y_act is given dy is first multiplied by act'(y_act)."""
    lib = _hip.load()
    if len(in_shape) == 4:
        b, c, h, w = in_shape
        t = 1
    else:
        b, c, t, h, w = in_shape
    to, ho, wo = spec.out_dims(t, h, w)
    eshape = (b, spec.cout, ho, wo) if len(in_shape) == 4 else (b, spec.cout, to, ho, wo)
    if tuple(dy.shape) != eshape or c != spec.cin or tuple(wp_d.shape) != spec.wp_d_shape():
        raise RuntimeError(f"conv_dgrad: shape mismatch dy={tuple(dy.shape)} expected {eshape}")
    if y_act is not None and y_act.shape != dy.shape:
        raise RuntimeError("conv_dgrad: y_act shape mismatch")
    if add is not None and tuple(add.shape) != tuple(in_shape):
        raise RuntimeError("conv_dgrad: add shape mismatch")
    if mask_y is not None and tuple(mask_y.shape) != tuple(in_shape):
        raise RuntimeError("conv_dgrad: mask_y shape mismatch")
    dx = torch.empty(in_shape, device=dy.device, dtype=torch.float32)
    _chk(dy, wp_d, y_act, dx, add, mask_y)
    d = spec.desc(b, t, h, w)
    e0 = _prof_begin()
    pre = _x6s_of(wp_d, d, True) if y_act is None else None
    ws = _x6_scratch(3 * wp_d.numel(), dy.device) if (pre is None and y_act is None and spec.cout % 16 == 0 and spec.cin > 1) else None
    if pre is not None:
        _hip.check(lib.p2i_conv_dgrad_x6s(d, _ptr(dy), _ptr(wp_d), pre[0], pre[1], _ptr(add), _ptr(mask_y), mask_act, _ptr(dx), _stream()),
                   "p2i_conv_dgrad_x6s")
    elif ws is not None:
        _hip.check(lib.p2i_conv_dgrad_x6(d, _ptr(dy), _ptr(wp_d), _ptr(ws), _ptr(add), _ptr(mask_y), mask_act, _ptr(dx), _stream()),
                   "p2i_conv_dgrad_x6")
    else:
        _hip.check(lib.p2i_conv_dgrad(d, _ptr(dy), _ptr(y_act), act, _ptr(wp_d), _ptr(add), _ptr(mask_y), mask_act, _ptr(dx), _stream()),
                   "p2i_conv_dgrad")
    _prof_end(e0, "dgrad", spec, d, spec.stride == (1, 1, 1))
    return dx


class ZeroArena:
    """One zero-filled buffer carved into the (atomically accumulated) gradient outputs of a backward pass:
    a single memset instead of one fill kernel per weight tensor."""

    def __init__(self, numel: int, device, buf: Optional[torch.Tensor] = None):
        """buf: a caller-owned buffer of >= numel floats to reuse (zeroed here by p2i_zero, no ATen fill kernel)."""
        if buf is not None and buf.numel() >= numel:
            self.buf = buf
            zero_(buf[:numel])
        else:
            self.buf = torch.zeros(numel, device=device, dtype=torch.float32)
        self.off = 0

    def take(self, shape):
        n = 1
        for d in shape:
            n *= d
        n4 = (n + 3) // 4 * 4                       # keep every carve 16-B aligned
        if self.off + n4 > self.buf.numel():
            return zero_(torch.empty(shape, device=self.buf.device, dtype=torch.float32))
        t = self.buf[self.off:self.off + n].view(shape)
        self.off += n4
        return t


def conv_wgrad(spec: ConvSpec, x, dy, y_act=None, act=ACT_NONE, want_bias=False, arena: Optional["ZeroArena"] = None, db_out=None,
               dwp_out=None):
    """Packed weight gradient dwp_f [ntaps][cin][pad32(cout)] (+ bias gradient).  db_out: caller-owned (cout,) tensor the bias
    gradient is ADDED to (e.g. the parameter's view of the flat gradient buffer)."""
    lib = _hip.load()
    det_ready(x.device)
    b, c, t, h, w = _dims5(x)
    to, ho, wo = spec.out_dims(t, h, w)
    eshape = (b, spec.cout, ho, wo) if x.dim() == 4 else (b, spec.cout, to, ho, wo)
    if tuple(dy.shape) != eshape or c != spec.cin:
        raise RuntimeError(f"conv_wgrad: shape mismatch dy={tuple(dy.shape)} expected {eshape}")
    if y_act is not None and y_act.shape != dy.shape:
        raise RuntimeError("conv_wgrad: y_act shape mismatch")
    if db_out is not None and (tuple(db_out.shape) != (spec.cout,) or not want_bias):
        raise RuntimeError("conv_wgrad: db_out must be (cout,) and want_bias set")
    if dwp_out is not None:                      # caller-owned, zeroed packed-gradient target
        if tuple(dwp_out.shape) != spec.wp_f_shape() or (want_bias and db_out is None):
            raise RuntimeError("conv_wgrad: dwp_out shape mismatch (and a bias gradient needs db_out with it)")
        dwp, db = dwp_out, db_out
    elif arena is not None:
        dwp = arena.take(spec.wp_f_shape())
        db = db_out if db_out is not None else (arena.take((spec.cout,)) if want_bias else None)
    else:
        dwp = zero_(torch.empty(spec.wp_f_shape(), device=x.device, dtype=torch.float32))
        db = db_out if db_out is not None else (zero_(torch.empty(spec.cout, device=x.device, dtype=torch.float32)) if want_bias else None)
    _chk(x, dy, y_act, dwp, db)
    d = spec.desc(b, t, h, w)
    e0 = _prof_begin()
    ws = _wgrad_scratch(x.device)
    if ws is not None:
        _hip.check(lib.p2i_conv_wgrad_ws(d, _ptr(x), _ptr(dy), _ptr(y_act), act, _ptr(dwp), _ptr(db), _ptr(ws), ws.numel(), _stream()),
                   "p2i_conv_wgrad_ws")
    else:
        _hip.check(lib.p2i_conv_wgrad(d, _ptr(x), _ptr(dy), _ptr(y_act), act, _ptr(dwp), _ptr(db), _stream()), "p2i_conv_wgrad")
    _prof_end(e0, "wgrad", spec, d, True)
    return dwp, db


# --------------------------------------------------------------------------- weights
def doconv_fold(W, D, D_diag, out_ch, in_ch, groups, ksz, identity_rep=0, need_d=True):
    lib = _hip.load()
    nt = ksz * ksz
    exp_w = (out_ch, in_ch // groups, nt)
    if tuple(W.shape) != exp_w:
        raise RuntimeError(f"doconv_fold: W shape {tuple(W.shape)} != {exp_w}")
    if ksz == 3 and (tuple(D.shape) != (in_ch, 9, 9) or tuple(D_diag.shape) != (in_ch, 9, 9)):
        raise RuntimeError("doconv_fold: D/D_diag shape mismatch")
    wp_f = torch.empty((nt, in_ch, pad32(out_ch)), device=W.device, dtype=torch.float32)
    wp_d = torch.empty((nt, out_ch, pad32(in_ch)), device=W.device, dtype=torch.float32) if need_d else None
    _chk(W, D, D_diag)
    _hip.check(lib.p2i_doconv_fold_fwd(_ptr(W), _ptr(D), _ptr(D_diag), out_ch, in_ch, groups, ksz, identity_rep,
                                       _ptr(wp_f), _ptr(wp_d), _stream()), "p2i_doconv_fold_fwd")
    return wp_f, wp_d


def doconv_fold_bwd(dwp_f, W, D, D_diag, out_ch, in_ch, groups, ksz, out=None):
    """out = (dW target, dD target or None): written in place (caller-owned, e.g. views of the flat gradient buffer)."""
    lib = _hip.load()
    if tuple(dwp_f.shape) != (ksz * ksz, in_ch, pad32(out_ch)):
        raise RuntimeError("doconv_fold_bwd: dwp shape mismatch")
    if out is not None:
        dW, dD = out
        if dW.shape != W.shape or (ksz == 3 and (dD is None or dD.shape != D.shape)):
            raise RuntimeError("doconv_fold_bwd: output target shape mismatch")
        _chk(dW, dD)
    else:
        dW = torch.empty_like(W)
        dD = torch.empty_like(D) if ksz == 3 else None
    _chk(dwp_f, W, D, D_diag)
    _hip.check(lib.p2i_doconv_fold_bwd(_ptr(dwp_f), _ptr(W), _ptr(D), _ptr(D_diag), out_ch, in_ch, groups, ksz,
                                       _ptr(dW), _ptr(dD), _stream()), "p2i_doconv_fold_bwd")
    return dW, dD


def _ptr_array(ts):
    import ctypes
    return (ctypes.c_void_p * len(ts))(*[None if t is None else t.data_ptr() for t in ts])


def doconv_fold_batched(layers, out_ch, in_ch, need_d=True):
    """doconv_fold for a list of same-shape 3x3 layers [(W, D, D_diag), ...] (groups 1) in ONE launch.
    Returns [(wp_f, wp_d), ...] (views of two stacked buffers)."""
    lib = _hip.load()
    n = len(layers)
    for W, D, Dd in layers:
        if tuple(W.shape) != (out_ch, in_ch, 9) or tuple(D.shape) != (in_ch, 9, 9) or tuple(Dd.shape) != (in_ch, 9, 9):
            raise RuntimeError("doconv_fold_batched: shape mismatch")
        _chk(W, D, Dd)
    dev = layers[0][0].device
    wf = torch.empty((n, 9, in_ch, out_ch), device=dev, dtype=torch.float32)
    wd = torch.empty((n, 9, out_ch, in_ch), device=dev, dtype=torch.float32) if need_d else None
    _hip.check(lib.p2i_doconv_fold_fwd_batched(_ptr_array([l[0] for l in layers]), _ptr_array([l[1] for l in layers]),
                                               _ptr_array([l[2] for l in layers]), n, out_ch, in_ch, _ptr_array([wf[i] for i in range(n)]),
                                               _ptr_array([wd[i] for i in range(n)]) if need_d else None, _stream()),
               "p2i_doconv_fold_fwd_batched")
    wfs = [wf[i] for i in range(n)]
    wds = [wd[i] for i in range(n)] if need_d else [None] * n
    X6Stack.attach(wf, wfs)
    if need_d:
        X6Stack.attach(wd, wds)
    return list(zip(wfs, wds))


def doconv_fold_bwd_batched(dwps, layers, out_ch, in_ch, outs=None):
    """doconv_fold_bwd for same-shape layers in TWO launches.  Returns ([dW...], [dD...]).  outs = ([dW targets], [dD targets])
    writes the results straight into caller-owned tensors (the flat gradient buffer's views)."""
    lib = _hip.load()
    n = len(layers)
    if len(dwps) != n:
        raise RuntimeError("doconv_fold_bwd_batched: list length mismatch")
    for g in dwps:
        if tuple(g.shape) != (9, in_ch, out_ch):
            raise RuntimeError("doconv_fold_bwd_batched: dwp shape mismatch")
        _chk(g)
    dev = dwps[0].device
    if outs is not None:
        dWs, dDs = outs
        for t, shp in [(t, (out_ch, in_ch, 9)) for t in dWs] + [(t, (in_ch, 9, 9)) for t in dDs]:
            if tuple(t.shape) != shp:
                raise RuntimeError("doconv_fold_bwd_batched: output target shape mismatch")
            _chk(t)
    else:
        dW = torch.empty((n, out_ch, in_ch, 9), device=dev, dtype=torch.float32)
        dD = torch.empty((n, in_ch, 9, 9), device=dev, dtype=torch.float32)
        dWs, dDs = [dW[i] for i in range(n)], [dD[i] for i in range(n)]
    _hip.check(lib.p2i_doconv_fold_bwd_batched(_ptr_array(dwps), _ptr_array([l[0] for l in layers]), _ptr_array([l[1] for l in layers]),
                                               _ptr_array([l[2] for l in layers]), n, out_ch, in_ch, _ptr_array(dWs),
                                               _ptr_array(dDs), _stream()), "p2i_doconv_fold_bwd_batched")
    return dWs, dDs


def weight_pack(w, sigma=None, need_f=True, need_d=True):
    """w (O, I, *k) -> packed wp_f / wp_d, optionally divided by the device scalar sigma."""
    lib = _hip.load()
    O, I = w.shape[0], w.shape[1]
    nt = w[0, 0].numel()
    wp_f = torch.empty((nt, I, pad32(O)), device=w.device, dtype=torch.float32) if need_f else None
    wp_d = torch.empty((nt, O, pad32(I)), device=w.device, dtype=torch.float32) if need_d else None
    _chk(w, sigma)
    _hip.check(lib.p2i_weight_pack(_ptr(w), O, I, nt, _ptr(sigma), _ptr(wp_f), _ptr(wp_d), _stream()), "p2i_weight_pack")
    return wp_f, wp_d


def weight_pack_batched(ws, sigmas=None, need_d=True, buf=None):
    """weight_pack for a list of (O, I, ntaps) weights (different shapes) in ONE launch; one zero-filled buffer holds every
    packed output (the padded columns must read 0).  Returns [(wp_f, wp_d), ...].  buf: a caller-owned buffer that was
    zero-filled ONCE and is only ever used for this same list of shapes (the pack kernel rewrites exactly the non-padding
    entries, so the padding stays zero): saves the per-call fill."""
    import ctypes
    lib = _hip.load()
    n = len(ws)
    Os, Is, NTs = [w.shape[0] for w in ws], [w.shape[1] for w in ws], [w.shape[2] for w in ws]
    for w in ws:
        if w.dim() != 3:
            raise RuntimeError("weight_pack_batched: weights must be (O, I, ntaps)")
        _chk(w)
    sizes_f = [nt * i * pad32(o) for o, i, nt in zip(Os, Is, NTs)]
    sizes_d = [nt * o * pad32(i) for o, i, nt in zip(Os, Is, NTs)] if need_d else [0] * n
    need = sum(sizes_f) + sum(sizes_d)
    if buf is None:
        buf = zero_(torch.empty(need, device=ws[0].device, dtype=torch.float32))
    elif buf.numel() != need:
        raise RuntimeError("weight_pack_batched: reused buffer has the wrong size")
    outs, off = [], 0
    for k in range(n):
        f = buf[off:off + sizes_f[k]].view(NTs[k], Is[k], pad32(Os[k]))
        off += sizes_f[k]
        d = None
        if need_d:
            d = buf[off:off + sizes_d[k]].view(NTs[k], Os[k], pad32(Is[k]))
            off += sizes_d[k]
        outs.append((f, d))
    arr_i = ctypes.c_int * n
    _hip.check(lib.p2i_weight_pack_batched(_ptr_array(ws), arr_i(*Os), arr_i(*Is), arr_i(*NTs),
                                           _ptr_array(sigmas) if sigmas is not None else None, _ptr_array([o[0] for o in outs]),
                                           _ptr_array([o[1] for o in outs]) if need_d else None, n, _stream()), "p2i_weight_pack_batched")
    return outs


def weight_unpack_grad_batched(dwps, likes, w_origs=None, sigmas=None, us=None, vs=None, outs=None, accumulate=False):
    """weight_unpack_grad for a list of layers (different shapes) in two launches.  Returns [dw, ...] shaped like `likes`.
    outs: caller-owned targets (numel must match); accumulate: dw += instead of dw = (second backward of a D step)."""
    import ctypes
    lib = _hip.load()
    det_ready(dwps[0].device)
    n = len(dwps)
    Os, Is, NTs = [l.shape[0] for l in likes], [l.shape[1] for l in likes], [l.shape[2] for l in likes]
    for g, o, i, nt in zip(dwps, Os, Is, NTs):
        if tuple(g.shape) != (nt, i, pad32(o)):
            raise RuntimeError("weight_unpack_grad_batched: dwp shape mismatch")
        _chk(g)
    if outs is not None:
        if len(outs) != n or any(o.numel() != l.numel() for o, l in zip(outs, likes)):
            raise RuntimeError("weight_unpack_grad_batched: output targets do not match")
        _chk(*outs)
        dws = outs
    else:
        if accumulate:
            raise RuntimeError("weight_unpack_grad_batched: accumulate needs caller-owned targets")
        dws = [torch.empty_like(l) for l in likes]
    dots = torch.empty(n, device=dwps[0].device, dtype=torch.float32)
    arr_i = ctypes.c_int * n
    none = [None] * n
    _hip.check(lib.p2i_weight_unpack_grad_batched_acc(_ptr_array(dwps), arr_i(*Os), arr_i(*Is), arr_i(*NTs), _ptr_array(w_origs or none),
                                                      _ptr_array(sigmas or none), _ptr_array(us or none), _ptr_array(vs or none),
                                                      _ptr(dots), _ptr_array(dws), n, int(accumulate), _stream()),
               "p2i_weight_unpack_grad_batched_acc")
    return dws


def weight_unpack_grad(dwp_f, like, w_orig=None, sigma=None, u=None, v=None, out=None):
    lib = _hip.load()
    det_ready(dwp_f.device)
    O, I = like.shape[0], like.shape[1]
    nt = like[0, 0].numel()
    if tuple(dwp_f.shape) != (nt, I, pad32(O)):
        raise RuntimeError("weight_unpack_grad: dwp shape mismatch")
    if out is not None and out.numel() != like.numel():
        raise RuntimeError("weight_unpack_grad: output target size mismatch")
    dw = out if out is not None else torch.empty_like(like)
    _chk(dw)
    scratch = torch.empty(4, device=like.device, dtype=torch.float32) if sigma is not None else None
    _chk(dwp_f, w_orig, sigma, u, v)
    _hip.check(lib.p2i_weight_unpack_grad(_ptr(dwp_f), O, I, nt, _ptr(w_orig), _ptr(sigma), _ptr(u), _ptr(v),
                                          _ptr(scratch), _ptr(dw), _stream()), "p2i_weight_unpack_grad")
    return dw


def spectral_norm(w, u, v, training: bool):
    """In-place power iteration on u, v (training) and sigma (device scalar tensor of shape (1,))."""
    lib = _hip.load()
    O = w.shape[0]
    K = w.numel() // O
    if u.numel() != O or v.numel() != K:
        raise RuntimeError("spectral_norm: u/v size mismatch")
    sigma = torch.empty(1, device=w.device, dtype=torch.float32)
    scratch = torch.empty(O + K + 4, device=w.device, dtype=torch.float32)
    _chk(w, u, v)
    _hip.check(lib.p2i_spectral_norm(_ptr(w), O, K, _ptr(u), _ptr(v), int(training), _ptr(sigma), _ptr(scratch), _stream()),
               "p2i_spectral_norm")
    return sigma


def spectral_norm_batched(ws, us, vs, training: bool, snapshot: bool = False):
    """spectral_norm() for a list of layers in 4 launches (these kernels are launch-latency bound).  Returns the list of
    sigma tensors (views of one buffer); with snapshot=True also (u copies, v copies) of the updated vectors (the backward
    pass needs the u, v of THIS forward; the next forward overwrites them in place)."""
    import ctypes
    lib = _hip.load()
    n = len(ws)
    if not (1 <= n <= 16 and len(us) == n and len(vs) == n):
        raise RuntimeError("spectral_norm_batched: 1..16 layers")
    Os = [w.shape[0] for w in ws]
    Ks = [w.numel() // w.shape[0] for w in ws]
    for w, u, v, O, K in zip(ws, us, vs, Os, Ks):
        if u.numel() != O or v.numel() != K:
            raise RuntimeError("spectral_norm_batched: u/v size mismatch")
        _chk(w, u, v)
    dev = ws[0].device
    sig = torch.empty(n, device=dev, dtype=torch.float32)
    offs = [0]
    for O, K in zip(Os, Ks):
        offs.append(offs[-1] + (O + K + 4 + 3) // 4 * 4)
    scratch = torch.empty(offs[-1], device=dev, dtype=torch.float32)
    arr_p = ctypes.c_void_p * n
    arr_i = ctypes.c_int * n
    wp, up, vp = arr_p(*[w.data_ptr() for w in ws]), arr_p(*[u.data_ptr() for u in us]), arr_p(*[v.data_ptr() for v in vs])
    sp = arr_p(*[sig.data_ptr() + 4 * i for i in range(n)])
    cp = arr_p(*[scratch.data_ptr() + 4 * o for o in offs[:-1]])
    usn = vsn = None
    usp = vsp = None
    if snapshot and training:
        uo, vo = [0], [0]
        for O, K in zip(Os, Ks):
            uo.append(uo[-1] + O)
            vo.append(vo[-1] + K)
        ubuf = torch.empty(uo[-1], device=dev, dtype=torch.float32)
        vbuf = torch.empty(vo[-1], device=dev, dtype=torch.float32)
        usn = [ubuf[uo[i]:uo[i + 1]] for i in range(n)]
        vsn = [vbuf[vo[i]:vo[i + 1]] for i in range(n)]
        usp, vsp = arr_p(*[t.data_ptr() for t in usn]), arr_p(*[t.data_ptr() for t in vsn])
    _hip.check(lib.p2i_spectral_norm_batched(wp, arr_i(*Os), arr_i(*Ks), up, vp, int(training), sp, cp, usp, vsp, n, _stream()),
               "p2i_spectral_norm_batched")
    sigs = [sig[i:i + 1] for i in range(n)]
    if snapshot:
        if not training:
            usn, vsn = [u.clone() for u in us], [v.clone() for v in vs]
        return sigs, usn, vsn
    return sigs


# --------------------------------------------------------------------------- generator glue
def attn_fwd(x, w0, b0, w1, b1):
    lib = _hip.load()
    B, T, H, W = x.shape
    out = torch.empty_like(x)
    _chk(x, w0, b0, w1, b1)
    if w0.numel() != T * T or b0.numel() != T or w1.numel() != T * T or b1.numel() != T:
        raise RuntimeError("attn_fwd: parameter size mismatch")
    _hip.check(lib.p2i_attn_fwd(_ptr(x), _ptr(w0), _ptr(b0), _ptr(w1), _ptr(b1), _ptr(out), B, T, H * W, _stream()), "p2i_attn_fwd")
    return out


def attn_bwd(x, w0, b0, w1, b1, dout, out=None):
    """out: four caller-owned, ZEROED targets shaped like (w0, b0, w1, b1) (the kernel adds atomically)."""
    lib = _hip.load()
    det_ready(x.device)
    B, T, H, W = x.shape
    if dout.shape != x.shape:
        raise RuntimeError("attn_bwd: dout shape mismatch")
    if out is not None:
        if len(out) != 4 or any(o.numel() != p_.numel() for o, p_ in zip(out, (w0, b0, w1, b1))):
            raise RuntimeError("attn_bwd: output targets do not match")
        g = list(out)
        _chk(*g)
    else:
        g = [zero_(torch.empty_like(p_)) for p_ in (w0, b0, w1, b1)]
    _chk(x, w0, b0, w1, b1, dout)
    _hip.check(lib.p2i_attn_bwd(_ptr(x), _ptr(w0), _ptr(b0), _ptr(w1), _ptr(b1), _ptr(dout), _ptr(g[0]), _ptr(g[1]), _ptr(g[2]),
                                _ptr(g[3]), B, T, H * W, _stream()), "p2i_attn_bwd")
    return g


_LINSPACE = {}


def _grid_tables(T, H, W, device):
    """torch.linspace(0,1,n) tables of layer.py:246-256, computed on the host CPU exactly as the
    reference's CPU path does (CPU and GPU linspace kernels differ in the last bit)."""
    key = (T, H, W, str(device))
    if key not in _LINSPACE:
        _LINSPACE[key] = tuple(torch.linspace(0, 1, n).to(device) for n in (W, H, T))
    return _LINSPACE[key]


def _idw_strict(pt_count):
    """A sample with 1..3 mask points: the reference raises in torch.topk(k=4) (layer.py:282); the kernels write zeros for it
    (include/p2i_hip.h) because raising needs the point counts on the host -- a device sync in the middle of the step.
    P2I_IDW_STRICT=1 pays that sync and raises the reference's error."""
    if _os.environ.get("P2I_IDW_STRICT", "0") == "1":
        n = pt_count.cpu()
        if bool(((n > 0) & (n < 4)).any()):
            raise RuntimeError("selected index k out of range (idw_3d_knn needs at least 4 mask points per sample, got %s)" % n.tolist())


def idw_amb_counts(amb, B, Q):
    """(voxels per sample that pass 1 left to pass 2, voxels per sample that pass 2's fixed-bound phase left to the exact replay) of
    the work buffer ops.idw_fwd(..., _amb_out=[...]) hands out (tests, tools)."""
    per = Q + 1 + (Q + 255) // 256
    return amb[:B * per].view(B, per)[:, 0].cpu(), amb[B * per:B * per + B * (Q + 1)].view(B, Q + 1)[:, 0].cpu()


def idw_fwd(vals_src, mask, tau=0.05, save=True, _amb_out=None):
    """vals_src, mask: (B,T,H,W).  Returns out and the saved selection (pt_pos, sel_idx, sel_w)."""
    lib = _hip.load()
    B, T, H, W = vals_src.shape
    if mask.shape != vals_src.shape:
        raise RuntimeError("idw_fwd: mask shape mismatch")
    dev = vals_src.device
    Q = T * H * W
    gx, gy, gz = _grid_tables(T, H, W, dev)
    out = torch.empty_like(vals_src)
    pt_pos = torch.empty(B * Q, device=dev, dtype=torch.int32)
    pt_count = torch.empty(B, device=dev, dtype=torch.int32)
    frame_count = torch.empty(B * T, device=dev, dtype=torch.int32)
    row_start = torch.empty(B * T * (H + 1), device=dev, dtype=torch.int32)
    pt_xyzn = torch.empty(B * Q * 4, device=dev, dtype=torch.float32)
    sel_idx = torch.empty(B * Q * 4, device=dev, dtype=torch.int32) if save else None
    sel_w = torch.empty(B * Q * 4, device=dev, dtype=torch.float32) if save else None
    _chk(vals_src, mask)
    if _os.environ.get("P2I_IDW_FAST", "1") != "0":     # two-pass search (p2i_hip.h); "0": the reference's scan for every voxel (A/B, tests)
        amb = torch.empty(B * (2 * Q + 2 + (Q + 255) // 256), device=dev, dtype=torch.int32)
        _hip.check(lib.p2i_idw_fwd_ws(_ptr(vals_src), _ptr(mask), _ptr(gx), _ptr(gy), _ptr(gz), _ptr(out), _ptr(pt_pos), _ptr(pt_count),
                                      _ptr(frame_count), _ptr(row_start), _ptr(pt_xyzn), _ptr(sel_idx), _ptr(sel_w), _ptr(amb), B, T, H, W,
                                      float(tau), _stream()), "p2i_idw_fwd_ws")
        if _amb_out is not None:                       # tests: the per-sample count of voxels left to the replay pass
            _amb_out.append(amb)
        _idw_strict(pt_count)
        return out, (pt_pos, pt_count, sel_idx, sel_w)
    _hip.check(lib.p2i_idw_fwd(_ptr(vals_src), _ptr(mask), _ptr(gx), _ptr(gy), _ptr(gz), _ptr(out), _ptr(pt_pos), _ptr(pt_count),
                               _ptr(frame_count), _ptr(row_start), _ptr(pt_xyzn), _ptr(sel_idx), _ptr(sel_w), B, T, H, W, float(tau), _stream()),
               "p2i_idw_fwd")
    _idw_strict(pt_count)
    return out, (pt_pos, pt_count, sel_idx, sel_w)


def idw_bwd(dout, saved):
    lib = _hip.load()
    det_ready(dout.device)
    B, T, H, W = dout.shape
    pt_pos, pt_count, sel_idx, sel_w = saved
    dvals = torch.empty_like(dout)
    _chk(dout)
    _hip.check(lib.p2i_idw_bwd(_ptr(dout), _ptr(pt_pos), _ptr(pt_count), _ptr(sel_idx), _ptr(sel_w), _ptr(dvals), B, T, H, W, _stream()),
               "p2i_idw_bwd")
    return dvals


def pooldup_fwd(x):
    lib = _hip.load()
    B, Cc, H, W = x.shape
    y = torch.empty((B, 2 * Cc, H // 2, W // 2), device=x.device, dtype=torch.float32)
    _chk(x)
    _hip.check(lib.p2i_pooldup_fwd(_ptr(x), _ptr(y), B, Cc, H, W, _stream()), "p2i_pooldup_fwd")
    return y


def pooldup_bwd(x, dy):
    lib = _hip.load()
    B, Cc, H, W = x.shape
    if tuple(dy.shape) != (B, 2 * Cc, H // 2, W // 2):
        raise RuntimeError("pooldup_bwd: dy shape mismatch")
    dx = torch.empty_like(x)
    _chk(x, dy)
    _hip.check(lib.p2i_pooldup_bwd(_ptr(x), _ptr(dy), _ptr(dx), B, Cc, H, W, _stream()), "p2i_pooldup_bwd")
    return dx


def upmod_fwd(x, pos, bias=None, act=ACT_NONE):
    """bilinear x2 (align_corners) * 2 sigmoid(pos); with `bias`: act(that + bias[c]) -- UPPos's tail when the 1x1 projection has
    already been applied at the low resolution (p2i_upmod_fwd_ba in p2i_hip.h has the algebra)."""
    lib = _hip.load()
    B, Cc, Sh, Sw = x.shape
    if pos.numel() != 4 * Sh * Sw:
        raise RuntimeError(f"upmod_fwd: pos has {pos.numel()} elements, expected {4 * Sh * Sw}")
    u = torch.empty((B, Cc, 2 * Sh, 2 * Sw), device=x.device, dtype=torch.float32)
    _chk(x, pos, bias)
    if bias is not None:
        if bias.numel() != Cc:
            raise RuntimeError("upmod_fwd: bias size mismatch")
        _hip.check(lib.p2i_upmod_fwd_ba(_ptr(x), _ptr(pos), _ptr(bias), act, _ptr(u), B, Cc, Sh, Sw, _stream()), "p2i_upmod_fwd_ba")
    else:
        _hip.check(lib.p2i_upmod_fwd(_ptr(x), _ptr(pos), _ptr(u), B, Cc, Sh, Sw, _stream()), "p2i_upmod_fwd")
    return u


def upmod_bwd(x, pos, du, need_dx=True, dpos_out=None):
    """dpos_out: caller-owned ZEROED target shaped like pos (the kernel adds atomically)."""
    lib = _hip.load()
    det_ready(x.device)
    B, Cc, Sh, Sw = x.shape
    if tuple(du.shape) != (B, Cc, 2 * Sh, 2 * Sw):
        raise RuntimeError("upmod_bwd: du shape mismatch")
    dx = torch.empty_like(x) if need_dx else None
    if dpos_out is not None and dpos_out.numel() != pos.numel():
        raise RuntimeError("upmod_bwd: dpos target size mismatch")
    dpos = dpos_out if dpos_out is not None else zero_(torch.empty_like(pos))
    _chk(dpos)
    _chk(x, pos, du)
    _hip.check(lib.p2i_upmod_bwd(_ptr(x), _ptr(pos), _ptr(du), _ptr(dx), _ptr(dpos), B, Cc, Sh, Sw, _stream()), "p2i_upmod_bwd")
    return dx, dpos


# --------------------------------------------------------------------------- discriminator tail
def dtail_fwd(out2d, out3d, alpha2d):
    lib = _hip.load()
    B, c2, H2, W2 = out2d.shape
    B3, c3, T3, H3, W3 = out3d.shape
    if c2 != 1 or c3 != 1 or B3 != B:
        raise RuntimeError("dtail_fwd: expects single-channel branch outputs")
    fused = torch.empty((B, H2 * W2), device=out2d.device, dtype=torch.float32)
    _chk(out2d, out3d, alpha2d)
    _hip.check(lib.p2i_dtail_fwd(_ptr(out2d), _ptr(out3d), _ptr(alpha2d), _ptr(fused), B, H2, W2, T3, H3, W3, _stream()), "p2i_dtail_fwd")
    return fused


def dtail_bwd(out2d, out3d_shape, alpha2d, dfused, need_alpha=True, da_out=None):
    """da_out: caller-owned (1,) target the alpha2d gradient is ADDED to."""
    lib = _hip.load()
    det_ready(out2d.device)
    B, _, H2, W2 = out2d.shape
    _, _, T3, H3, W3 = out3d_shape
    d2 = torch.empty_like(out2d)
    d3 = torch.empty(out3d_shape, device=out2d.device, dtype=torch.float32)
    da = (da_out if da_out is not None else zero_(torch.empty_like(alpha2d))) if need_alpha else None
    _chk(da)
    _chk(out2d, alpha2d, dfused)
    _hip.check(lib.p2i_dtail_bwd(_ptr(out2d), _ptr(alpha2d), _ptr(dfused), _ptr(d2), _ptr(d3), _ptr(da), B, H2, W2, T3, H3, W3,
                                 _stream()), "p2i_dtail_bwd")
    return d2, d3, da


# --------------------------------------------------------------------------- losses / optimiser
def recloss(pred, target, k1_alpha):
    """Returns (out3 device tensor [pool, reg, pool+k1*reg], dpred)."""
    lib = _hip.load()
    B, T = pred.shape[0], pred.shape[1]
    HW = pred[0, 0].numel()
    if pred.shape != target.shape:
        raise RuntimeError("recloss: shape mismatch")
    out3 = torch.empty(3, device=pred.device, dtype=torch.float32)
    dpred = torch.empty_like(pred)
    scratch = torch.empty(B * (T - 1) * HW + 4096, device=pred.device, dtype=torch.float32)
    _chk(pred, target)
    _hip.check(lib.p2i_recloss(_ptr(pred), _ptr(target), float(k1_alpha), _ptr(out3), _ptr(dpred), _ptr(scratch), B, T, HW, _stream()),
               "p2i_recloss")
    return out3, dpred


_LOSS_TYPES = {"hinge": 0, "lsgan": 1, "nsgan": 2}


def _check_bce(loss, loss_type):
    """nn.BCELoss (the reference's 'nsgan', losses.py:201-202) raises for inputs outside [0, 1]; the kernel flags that with a NaN
    loss.  Costs a host sync, on the nsgan path only (the named configs use hinge)."""
    if loss_type == "nsgan" and bool(torch.isnan(loss).any()):
        raise RuntimeError("all elements of input should be between 0 and 1")


def gan_loss_d(logits_real, logits_fake, loss_type="hinge", real_label=1.0, fake_label=0.0):
    lib = _hip.load()
    if loss_type not in _LOSS_TYPES:
        raise ValueError(f"Unsupported GAN loss type: {loss_type}")
    n = logits_real.numel()
    loss = torch.empty(1, device=logits_real.device, dtype=torch.float32)
    da, db = torch.empty_like(logits_real), torch.empty_like(logits_fake)
    _chk(logits_real, logits_fake)
    _hip.check(lib.p2i_gan_loss(_ptr(logits_real), _ptr(logits_fake), n, _LOSS_TYPES[loss_type], 0, 1.0, real_label, fake_label,
                                _ptr(loss), _ptr(da), _ptr(db), _stream()), "p2i_gan_loss")
    _check_bce(loss, loss_type)
    return loss, da, db


def gan_loss_g(logits, weight, loss_type="hinge", real_label=1.0):
    lib = _hip.load()
    if loss_type not in _LOSS_TYPES:
        raise ValueError(f"Unsupported GAN loss type: {loss_type}")
    loss = torch.empty(1, device=logits.device, dtype=torch.float32)
    da = torch.empty_like(logits)
    _chk(logits)
    _hip.check(lib.p2i_gan_loss(_ptr(logits), None, logits.numel(), _LOSS_TYPES[loss_type], 1, float(weight), real_label, 0.0,
                                _ptr(loss), _ptr(da), None, _stream()), "p2i_gan_loss")
    _check_bce(loss, loss_type)
    return loss, da


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step):
    lib = _hip.load()
    n = p.numel()
    if not (g.numel() == n and m.numel() == n and v.numel() == n):
        raise RuntimeError("adam_step: buffer size mismatch")
    _chk(p, g, m, v)
    _hip.check(lib.p2i_adam(_ptr(p), _ptr(g), _ptr(m), _ptr(v), n, lr, beta1, beta2, eps, step, _stream()), "p2i_adam")


def adam_step_dev(p, g, m, v, lr, beta1, beta2, eps, step_dev, coef):
    """Adam with the step counter on the device (int32 tensor, incremented by the call): capturable in a hipGraph."""
    lib = _hip.load()
    n = p.numel()
    if not (g.numel() == n and m.numel() == n and v.numel() == n) or step_dev.dtype != torch.int32 or coef.numel() < 2:
        raise RuntimeError("adam_step_dev: bad buffers")
    _chk(p, g, m, v, step_dev, coef)
    _hip.check(lib.p2i_adam_dev(_ptr(p), _ptr(g), _ptr(m), _ptr(v), n, lr, beta1, beta2, eps, _ptr(step_dev), _ptr(coef), _stream()),
               "p2i_adam_dev")


def axpy_(y, x, a=1.0):
    lib = _hip.load()
    if y.numel() != x.numel():
        raise RuntimeError("axpy_: size mismatch")
    _chk(y, x)
    _hip.check(lib.p2i_axpy(_ptr(y), _ptr(x), float(a), y.numel(), _stream()), "p2i_axpy")
    return y


def zero_(t):
    """t[...] = 0 (hipMemsetAsync on torch's current stream)."""
    lib = _hip.load()
    _chk(t)
    _hip.check(lib.p2i_zero(_ptr(t), t.numel(), _stream()), "p2i_zero")
    return t


def add2(a, b):
    """a + b (new tensor)."""
    lib = _hip.load()
    if a.shape != b.shape:
        raise RuntimeError("add2: shape mismatch")
    out = torch.empty_like(a)
    _chk(a, b)
    _hip.check(lib.p2i_add2(_ptr(out), _ptr(a), _ptr(b), a.numel(), _stream()), "p2i_add2")
    return out


def act_bwd(dy, y, act):
    """dy * act'(y) for the saved post-activation tensor y."""
    lib = _hip.load()
    if dy.shape != y.shape:
        raise RuntimeError("act_bwd: shape mismatch")
    out = torch.empty_like(dy)
    _chk(dy, y)
    _hip.check(lib.p2i_act_bwd(_ptr(dy), _ptr(y), act, _ptr(out), dy.numel(), _stream()), "p2i_act_bwd")
    return out


def act_bwd_bias(dy, y, act, db_out=None):
    """(dy * act'(y), its per-channel sum) in one pass; db_out: caller-owned ZEROED (C,) target the kernel adds into."""
    lib = _hip.load()
    det_ready(dy.device)
    B, Cc = dy.shape[0], dy.shape[1]
    inner = dy[0, 0].numel()
    if inner % 4:
        dz = act_bwd(dy, y, act)
        return dz, bias_grad(dz, out=db_out)
    out = torch.empty_like(dy)
    db = db_out if db_out is not None else zero_(torch.empty(Cc, device=dy.device, dtype=torch.float32))
    _chk(dy, y, db)
    _hip.check(lib.p2i_act_bwd_bias(_ptr(dy), _ptr(y), act, _ptr(out), _ptr(db), B, Cc, inner, _stream()), "p2i_act_bwd_bias")
    return out, db


def bias_grad(dy, y_act=None, act=ACT_NONE, out=None):
    """out: caller-owned ZEROED (Cc,) target the kernel adds into."""
    lib = _hip.load()
    det_ready(dy.device)
    B, Cc = dy.shape[0], dy.shape[1]
    inner = dy[0, 0].numel()
    if out is not None and out.numel() != Cc:
        raise RuntimeError("bias_grad: target size mismatch")
    db = out if out is not None else zero_(torch.empty(Cc, device=dy.device, dtype=torch.float32))
    _chk(dy, y_act)
    _hip.check(lib.p2i_bias_grad(_ptr(dy), _ptr(y_act), act, _ptr(db), B, Cc, inner, _stream()), "p2i_bias_grad")
    return db


# --------------------------------------------------------------------------- sliding-window inference
def window_gather(a, b, L, w0, nw, win, step):
    """Windows w0 .. w0+nw-1 (win frames every `step`, the last frame repeated past the end) of the (L, ...) tensors a and b
    (b may be None) as (nw, win, ...) tensors."""
    lib = _hip.load()
    hw = a[0].numel()
    wa = torch.empty((nw, win) + tuple(a.shape[1:]), device=a.device, dtype=torch.float32)
    wb = torch.empty_like(wa) if b is not None else None
    _chk(a, b)
    _hip.check(lib.p2i_window_gather(_ptr(a), _ptr(b), _ptr(wa), _ptr(wb), L, hw, w0, nw, win, step, _stream()), "p2i_window_gather")
    return wa, wb


def window_mean(pred_windows, L, win, step, scale):
    """(nwin, win, ...) predictions of all windows of an event -> (L, ...): mean over the windows that hold each frame, * scale, >= 0."""
    lib = _hip.load()
    nwin = pred_windows.shape[0]
    hw = pred_windows[0, 0].numel()
    out = torch.empty((L,) + tuple(pred_windows.shape[2:]), device=pred_windows.device, dtype=torch.float32)
    _chk(pred_windows)
    _hip.check(lib.p2i_window_mean(_ptr(pred_windows), _ptr(out), L, hw, nwin, win, step, float(scale), _stream()), "p2i_window_mean")
    return out


# --------------------------------------------------------------------------- batch assembly
def assemble_batch(frames_u8, mask_u8):
    """uint8 frames (B,T,H,W) + uint8 mask (H,W) | (T,H,W) | (B,T,H,W) -> (frames, masked, masks), each (B,T,1,H,W) fp32:
    the loader's /255, video*mask and channel permute (sti_dataset.py:209,223-224; train.py:468-473) in one pass."""
    lib = _hip.load()
    if frames_u8.dtype != torch.uint8 or mask_u8.dtype != torch.uint8 or frames_u8.dim() != 4:
        raise RuntimeError("assemble_batch: expects uint8 frames (B,T,H,W) and a uint8 mask")
    B, T, H, W = frames_u8.shape
    if tuple(mask_u8.shape) not in ((H, W), (T, H, W), (B, T, H, W)):
        raise RuntimeError(f"assemble_batch: mask shape {tuple(mask_u8.shape)} does not match frames {tuple(frames_u8.shape)}")
    _chk(frames_u8, mask_u8)
    out = [torch.empty((B, T, 1, H, W), device=frames_u8.device, dtype=torch.float32) for _ in range(3)]
    _hip.check(lib.p2i_assemble_batch(_ptr(frames_u8), _ptr(mask_u8), mask_u8.numel(), _ptr(out[0]), _ptr(out[1]), _ptr(out[2]),
                                      B, T, H, W, _stream()), "p2i_assemble_batch")
    return out[0], out[1], out[2]
