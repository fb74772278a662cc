"""P2I-GAN generator / discriminator on the HIP path.

Drop-in for ``p2igan_bench.models.p2igan`` of the reference (p2igan.py:23-173): same constructor
arguments, same ``forward`` signatures, same ``state_dict`` keys/shapes (so reference checkpoints
load), same RNG consumption order at construction (so the same seed gives the same weights).
The module tree below only HOLDS parameters; all arithmetic runs in libp2i_hip.so through one
``torch.autograd.Function`` per network that sequences the kernels of ``p2igan_bench.ops`` and
keeps the activations it needs for the hand-written backward.
"""
from __future__ import annotations

import math
import os
from typing import List

import torch
import torch.nn as nn
from torch.nn import init

from .. import ops
from ..ops import ACT_LEAKY, ACT_NONE, ACT_RELU, ACT_TANH, ConvSpec

BASE_CH = 64                      # p2igan.py:46 (T = 16); the build's generalisation is base = 4 * T, see P2IGenerator
SUPPORTED_T = (8, 16, 32)         # AttentionBlock kernel instantiations (csrc/glue.hip)


# ----------------------------------------------------------------------------- parameter holders
class _DOConvParams(nn.Module):
    """Parameters of DOConv2d (deconv_pytorch.py:52-70): W (O, I/g, k*k), D (I, k*k, k*k), frozen D_diag.
    folded=True is DOConv2d_eval (deconv_pytorch.py:133-209): the composite kernel W (O, I/g, k, k) is the only
    parameter; it runs through the same fold kernel with D + D_diag = identity (non-persistent buffers)."""

    def __init__(self, in_ch: int, out_ch: int, ksz: int, groups: int = 1, folded: bool = False):
        super().__init__()
        self.in_ch, self.out_ch, self.ksz, self.groups, self.folded = in_ch, out_ch, ksz, groups, folded
        mn = ksz * ksz
        if folded:
            self.W = nn.Parameter(torch.empty(out_ch, in_ch // groups, ksz, ksz))
            init.kaiming_uniform_(self.W, a=math.sqrt(5))
            if mn > 1:
                self.register_buffer("_D0", torch.zeros(in_ch, mn, mn), persistent=False)
                self.register_buffer("_Dd", torch.eye(mn).reshape(1, mn, mn).repeat(in_ch, 1, 1), persistent=False)
            return
        self.W = nn.Parameter(torch.empty(out_ch, in_ch // groups, mn))
        init.kaiming_uniform_(self.W, a=math.sqrt(5))
        if mn > 1:
            self.D = nn.Parameter(torch.zeros(in_ch, mn, mn))
            self.D_diag = nn.Parameter(torch.eye(mn).reshape(1, mn, mn).repeat(in_ch, 1, 1), requires_grad=False)

    def tensors(self):
        if self.folded:
            w = self.W.view(self.out_ch, self.in_ch // self.groups, self.ksz * self.ksz)
            return (w, self._D0, self._Dd) if self.ksz > 1 else (w, None, None)
        if self.ksz > 1:
            return self.W, self.D, self.D_diag
        return self.W, None, None


class _Holder(nn.Module):
    """`main` = ModuleList so that keys read ....main.<idx>.… like nn.Sequential in the reference."""

    def __init__(self, mods: List[nn.Module]):
        super().__init__()
        self.main = nn.ModuleList(mods)


def _basic_conv(in_ch, out_ch, ksz, groups=1, folded=False):   # BasicConv_do / BasicConv_do_eval, layer.py:68-94
    return _Holder([_DOConvParams(in_ch, out_ch, ksz, groups, folded)])


class _EBlockParams(nn.Module):                          # EBlock / ResBlock_do, p2igan.py:176-183, layer.py:126-135
    def __init__(self, ch: int, num_res: int, folded: bool = False):
        super().__init__()
        self.layers = nn.ModuleList([_Holder([_basic_conv(ch, ch, 3, folded=folded), _basic_conv(ch, ch, 3, folded=folded)])
                                     for _ in range(num_res)])


class _AttnParams(nn.Module):                            # AttentionBlock, layer.py:296-299
    def __init__(self, c: int):
        super().__init__()
        self.conv = nn.Conv1d(c, c, kernel_size=1)       # parameter container only


class _InputParams(nn.Module):                           # InputBlock, layer.py:307-314
    def __init__(self, depth: int, t: int):
        super().__init__()
        self.layers = nn.ModuleList([_AttnParams(t) for _ in range(depth)])


class _UPPosParams(nn.Module):                           # UPPos, layer.py:384-390
    def __init__(self, in_ch, out_ch, H, W):
        super().__init__()
        self.pos = nn.Parameter(torch.zeros(1, 1, H, W))
        self.proj = nn.Conv2d(in_ch, out_ch, kernel_size=1, bias=True)   # parameter container only


def _spec2d(cin, cout, k, stride=1):
    return ConvSpec(cin, cout, (1, k, k), (1, stride, stride), (0, k // 2, k // 2))


# ----------------------------------------------------------------------------- generator
class P2IGenerator(nn.Module):
    def __init__(self, config, length: int = 16, num_res: int = 4, inference: bool = False, init_weights: bool = True):
        super().__init__()
        data_cfg = config.get("data_loader") or config["data"]["train"]
        self.keep = data_cfg.get("mask", {}).get("keep", 0)
        self.H = data_cfg["h"]
        self.W = data_cfg["w"]
        length = data_cfg.get("sample_length", length)
        # The reference is hard-wired to T = 16 (AttentionBlock(16) layer.py:310, base_channel 64 = 4*16 with
        # repeat_interleave(4) p2igan.py:46,66,79) and raises for anything else.  BASELINE configs[4] asks for T = 32, so
        # this build generalises the way SURVEY.md H5 prescribes: AttentionBlock(T), Convsin T -> 4T (groups 4),
        # base_channel = 4T, ConvsOut 4T -> T, discriminator in_channels = T.  T = 16 is unchanged (same keys, shapes and
        # arithmetic); T != 16 has NO reference behaviour: "parity unpinned -- self-consistency with the oracle only".
        if length not in SUPPORTED_T:
            raise RuntimeError(f"P2IGenerator: sample_length {length} not in {SUPPORTED_T} (the reference itself only runs T=16, layer.py:310)")
        self.length = length
        self.base = base = 4 * length
        self.num_res = num_res
        self.inference = inference
        # construction order == reference (p2igan.py:44-67) so that torch's RNG is consumed identically
        self.input = _InputParams(depth=2, t=length)
        self.Decoder = nn.ModuleList([_EBlockParams(base << l, num_res, folded=inference) for l in range(4)])
        self.ConvsOut = nn.ModuleList([_basic_conv(base, length, 1, groups=4, folded=inference)])
        self.UP = nn.ModuleList([
            _UPPosParams(base * 2, base, self.H, self.W),
            _UPPosParams(base * 4, base * 2, self.H // 2, self.W // 2),
            _UPPosParams(base * 8, base * 4, self.H // 4, self.W // 4),
        ])
        self.Convsin = nn.ModuleList([_basic_conv(length, base, 3, groups=4, folded=inference)])
        if init_weights:
            self.init_weights()
        self._pnames = [n for n, _ in self.named_parameters()]
        # packed/folded weights of the last no-grad forward, reused while the parameters are unchanged
        # (sliding-window inference calls the generator once per window batch): see _cached()
        self._wp_cache = {}
        self._weights_epoch = 0
        self.debug_taps = None      # set to a dict to receive intermediate tensors of the next forward (tests localise regressions)

    def invalidate_weight_cache(self):
        """Must be called by anything that rewrites parameters behind autograd's back (FusedAdam's raw-pointer
        update does); torch-level writes (load_state_dict, copy_) are seen through the version counters."""
        self._weights_epoch += 1

    def _cached(self, key, tensors, make):
        """make() -> packed weights, memoised on (epoch, version counters, storage) of `tensors`."""
        ver = (self._weights_epoch,) + tuple((t._version, t.data_ptr()) for t in tensors if t is not None)
        hit = self._wp_cache.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
        val = make()
        self._wp_cache[key] = (ver, val)
        return val

    def init_weights(self, init_type: str = "kaiming", gain: float = 0.02):
        """BaseNetwork.init_weights (layer.py:20-40): only modules exposing `.weight` are touched, i.e. the
        AttentionBlock Conv1d and UPPos proj Conv2d; DO-Conv W keeps kaiming_uniform(a=sqrt(5))."""
        if init_type != "kaiming":
            raise NotImplementedError(init_type)
        for m in [l.conv for l in self.input.layers] + [u.proj for u in self.UP]:
            init.kaiming_normal_(m.weight.data, a=0, mode="fan_in")
            init.constant_(m.bias.data, 0.0)

    def _arena_numel(self) -> int:
        """Floats of all packed weight / bias gradients one backward pass accumulates into."""
        if not hasattr(self, "_arena_n"):
            n = 0
            for lvl in range(4):
                ch = self.base << lvl
                n += 2 * self.num_res * 9 * ch * ops.pad32(ch)
            for up in self.UP:
                co, ci = up.proj.weight.shape[0], up.proj.weight.shape[1]
                n += ci * ops.pad32(co) + co + 8
            n += 9 * self.length * ops.pad32(self.base) + self.base * ops.pad32(self.length) + 64
            self._arena_n = n
        return self._arena_n

    def forward(self, masked_frames, masks):
        params = [p for _, p in self.named_parameters()]
        self._grad_on = torch.is_grad_enabled()      # Function.forward always runs with grad mode off: capture it here
        return _GeneratorFn.apply(self, masked_frames, masks, *params)


def _doconv_of(holder):      # BasicConv holder -> DOConv params
    return holder.main[0]


import os as _os

SIDE_WGRAD = _os.environ.get("P2I_SIDE_WGRAD", "1") != "0"
LATE_JOIN = _os.environ.get("P2I_LATE_JOIN", "1") != "0"       # generator_backward: weight-side kernels on the side stream, one join


def _side_of(net, device):
    """The network's side stream (ops.SideStream) for weight-gradient kernels, or None when disabled -- or when the caller already
    runs on a side stream (TrainEngine runs D's real half on the generator's): streams are forked from the step's origin stream only
    (a fork of a fork crashes hipGraph capture on ROCm 7.2, see ops._SIDE_DEPTH; P2I_NESTED_SIDE=1 allows it for eager A/B runs)."""
    if not SIDE_WGRAD:
        return None
    if ops.side_depth() > 0 and _os.environ.get("P2I_NESTED_SIDE", "0") != "1":
        return None
    sd = getattr(net, "_side", None)
    if sd is None or sd.stream.device != device:
        sd = net._side = ops.SideStream(device)
    return sd


def _grad_target(prm, inplace):
    """The parameter's view of the flat gradient buffer when gradients are written in place (TrainEngine), else None."""
    return prm.grad if inplace else None


def generator_prepare(net: "P2IGenerator", b: int, h: int, w: int):
    """Everything of a training forward that depends on the PARAMETERS only: the DO-Conv folds of all levels, the 1x1 packs and the
    bf16 split images the conv calls will ask for.  TrainEngine runs this on a side stream while the main stream computes the
    attention block and the IDW (latency-bound kernels that leave most of the chip idle); generator_forward(prep=...) then finds
    every weight ready.  Returns the dict generator_forward consumes."""
    t, BASE_CH = net.length, net.base
    prep = {"folded": {}, "up": {}}
    cin = _doconv_of(net.Convsin[0])
    prep["wp_in"] = ops.doconv_fold(*cin.tensors(), BASE_CH, t, 4, 3, identity_rep=4, need_d=True)
    for lvl in (3, 2, 1, 0):
        ch = BASE_CH << lvl
        convs = [_doconv_of(m) for rb in net.Decoder[lvl].layers for m in (rb.main[0], rb.main[1])]
        wps = ops.doconv_fold_batched([cv.tensors() for cv in convs], ch, ch, need_d=True)
        for cv, wp in zip(convs, wps):
            prep["folded"][id(cv)] = wp
        d = _spec2d(ch, ch, 3).desc(b, 1, h >> lvl, w >> lvl)
        ops._x6s_of(wps[0][0], d, False, ACT_RELU)       # splits the level's whole forward / data-gradient stacks now
        ops._x6s_of(wps[0][1], d, True)                   # (no-ops where the layer will not run on the bf16-split kernels)
    for i in (2, 1, 0):
        up = net.UP[i]
        cin_, cout_ = up.proj.weight.shape[1], up.proj.weight.shape[0]
        prep["up"][i] = ops.weight_pack(up.proj.weight.reshape(cout_, cin_, 1), need_d=True)
    cout = _doconv_of(net.ConvsOut[0])
    prep["wp_out"] = ops.doconv_fold(*cout.tensors(), t, BASE_CH, 4, 1, need_d=True)
    return prep


def generator_forward(net: "P2IGenerator", masked_frames, masks, need_grad: bool, prep=None, weights_ready=None, after_head=None):
    """P2IGenerator.forward (p2igan.py:72-112) as a plain function: returns (frames (B,T,1,H,W), saved state or None).
    Sequences the kernels of p2igan_bench.ops; no autograd involved (TrainEngine calls this directly, _GeneratorFn wraps it).
    prep: generator_prepare's result (training only); weights_ready(): called once, right before the first kernel that reads a
    prepared weight (TrainEngine: joins the side stream the preparation runs on).  after_head(): called right after the attention
    block and the IDW are enqueued -- latency-bound kernels that need no weights -- and may RETURN the prep dict: TrainEngine enqueues
    its side-stream work there, so that the GPU starts on the step's head before the host has issued the ~100 side-stream launches."""
    b, t, c, h, w = masked_frames.shape
    BASE_CH = net.base
    if c != 1 or t != net.length:
        raise RuntimeError(f"generator expects (B,{net.length},1,H,W), got {tuple(masked_frames.shape)}")
    if h % 8 or w % 8:
        raise RuntimeError("H and W must be multiples of 8")
    if need_grad and net.inference:
        raise RuntimeError("P2IGenerator(inference=True) holds folded DO-Conv kernels and is forward-only")

    def fold(conv, out_ch, in_ch, groups, ksz, identity_rep=0):
        make = lambda: ops.doconv_fold(*conv.tensors(), out_ch, in_ch, groups, ksz, identity_rep=identity_rep, need_d=need_grad)
        return make() if need_grad else net._cached(id(conv), conv.tensors(), make)

    x0 = masked_frames.reshape(b, t, h, w).contiguous().float()
    mk = masks.reshape(b, t, h, w).contiguous().float()
    att = [l.conv for l in net.input.layers]
    a = ops.attn_fwd(x0, att[0].weight, att[0].bias, att[1].weight, att[1].bias)
    idw, sel = ops.idw_fwd(a, mk, tau=0.05, save=need_grad)
    del a
    if after_head is not None:
        late = after_head()
        if late is not None:
            prep, weights_ready = late
    if weights_ready is not None:
        weights_ready()
    cin = _doconv_of(net.Convsin[0])
    wp_in = prep["wp_in"] if prep is not None else fold(cin, BASE_CH, t, 4, 3, identity_rep=4)
    spec_in = _spec2d(t, BASE_CH, 3)
    x_ = ops.conv_fwd(spec_in, idw, wp_in[0])
    x_2 = ops.pooldup_fwd(x_)
    x_4 = ops.pooldup_fwd(x_2)
    x_8 = ops.pooldup_fwd(x_4)

    def eblock(lvl, hcur):
        ch = BASE_CH << lvl
        spec = _spec2d(ch, ch, 3)
        rec = []
        folded = prep["folded"] if prep is not None else {}
        if need_grad and prep is None:               # the level's 2*num_res same-shape folds in ONE launch
            convs = [_doconv_of(m) for rb in net.Decoder[lvl].layers for m in (rb.main[0], rb.main[1])]
            for cv, wp in zip(convs, ops.doconv_fold_batched([cv.tensors() for cv in convs], ch, ch, need_d=True)):
                folded[id(cv)] = wp
        for rb in net.Decoder[lvl].layers:
            c1, c2 = _doconv_of(rb.main[0]), _doconv_of(rb.main[1])
            w1 = folded[id(c1)] if need_grad else fold(c1, ch, ch, 1, 3)
            w2 = folded[id(c2)] if need_grad else fold(c2, ch, ch, 1, 3)
            y1 = ops.conv_fwd(spec, hcur, w1[0], act=ACT_RELU)
            hn = ops.conv_fwd(spec, y1, w2[0], residual=hcur)
            rec.append((hcur, y1, w1[1], w2[1]))
            hcur = hn
        return hcur, rec

    def uppos(i, hcur):
        # UPPos (layer.py:384-399): relu(W (up(h) * s) + b) with s = 2 sigmoid(pos) ONE factor per pixel.  The projection acts on
        # channels, the upsampling and s on positions, alike for every channel: they commute, relu(up(W h) * s + b) is the same
        # function -- projected at the LOW resolution (a quarter of the GEMM's positions, and the upsampling runs on C/2 channels),
        # finished by one bandwidth kernel.  Equal to the reference's order to fp32 rounding of the reordered sums.
        up = net.UP[i]
        cin_, cout_ = up.proj.weight.shape[1], up.proj.weight.shape[0]
        pack = lambda: ops.weight_pack(up.proj.weight.reshape(cout_, cin_, 1), need_d=need_grad)
        wp = prep["up"][i] if prep is not None else (pack() if need_grad else net._cached(id(up), (up.proj.weight,), pack))
        v = ops.conv_fwd(_spec2d(cin_, cout_, 1), hcur, wp[0])
        r = ops.upmod_fwd(v, up.pos, bias=up.proj.bias, act=ACT_RELU)
        return r, (hcur, v, r, wp[1])

    h3, rec3 = eblock(3, x_8)
    res1, up2 = uppos(2, h3)
    x4s = ops.add2(x_4, res1)                                   # x_4 + res1, p2igan.py:95 (the only skip)
    h2, rec2 = eblock(2, x4s)
    res2, up1 = uppos(1, h2)
    h1, rec1 = eblock(1, res2)
    res3, up0 = uppos(0, h1)
    h0, rec0 = eblock(0, res3)
    cout = _doconv_of(net.ConvsOut[0])
    wp_out = prep["wp_out"] if prep is not None else fold(cout, t, BASE_CH, 4, 1)
    spec_out = _spec2d(BASE_CH, t, 1)
    z = ops.conv_fwd(spec_out, h0, wp_out[0], act=ACT_TANH)
    if net.debug_taps is not None:
        net.debug_taps.update(idw=idw, x_=x_, x_8=x_8, dec3=h3, res1=res1, res3=res3)
    S = None
    if need_grad:
        S = dict(x0=x0, idw=idw, sel=sel, wp_in_d=wp_in[1], x_=x_, x_2=x_2, x_4=x_4,
                 rec=[rec0, rec1, rec2, rec3], up=[up0, up1, up2], h0=h0, z=z, wp_out_d=wp_out[1], shape=(b, t, h, w))
    return z.view(b, t, c, h, w), S


def generator_backward(net: "P2IGenerator", S, dout, inplace: bool = False, on_level_done=None):
    """Hand-sequenced backward of generator_forward.  inplace=False: returns {id(parameter): gradient}.  inplace=True (TrainEngine:
    every .grad is a zeroed view of the flat gradient buffer and every parameter is used once): gradients are written straight
    into the .grad views -- no AccumulateGrad add, no copy -- and an empty dict is returned.  on_level_done(lvl) (inplace only) is
    called on the launch stream right after the gradients of Decoder[lvl] are complete (data-parallel runs start that level's
    all-reduce there, beside the rest of the backward)."""
    b, t, h, w = S["shape"]
    BASE_CH = net.base
    grads = {}
    if inplace:
        plist = getattr(net, "_plist", None)          # (Module.parameters() walks the tree: cached; parameters are never replaced)
        if plist is None:
            plist = net._plist = [p_ for p_ in net.parameters() if p_.requires_grad]
        if any(p_.grad is None or not p_.grad.is_contiguous() for p_ in plist):
            raise RuntimeError("generator_backward(inplace=True) needs contiguous .grad views on every trainable parameter")
    # in-place mode owns a persistent arena (nothing of it outlives this call); the autograd path hands arena views to
    # AccumulateGrad, which may keep them, so it gets a fresh one
    if inplace and (getattr(net, "_arena_buf", None) is None or net._arena_buf.numel() < net._arena_numel() or net._arena_buf.device != dout.device):
        net._arena_buf = torch.empty(net._arena_numel(), device=dout.device, dtype=torch.float32)
    arena = ops.ZeroArena(net._arena_numel(), dout.device, buf=net._arena_buf if inplace else None)
    side = _side_of(net, dout.device) if inplace else None       # weight-gradient kernels beside the data-gradient chain

    def wgrad(spec, x, dy, **kw):
        if side is None:
            return ops.conv_wgrad(spec, x, dy, arena=arena, **kw)
        tgt = arena.take(spec.wp_f_shape())                        # carve on the main thread of control: the arena is not stream-safe
        return side.run(lambda: ops.conv_wgrad(spec, x, dy, dwp_out=tgt, **kw), x, dy)

    # The kernels that CONSUME packed weight gradients (DO-Conv fold backward, 1x1 unpack) read weights and wgrad results only:
    # they follow the wgrads on the side stream, so the data-gradient chain never waits for a weight gradient (round 2 joined
    # the streams at every level, 1x1 projection and at both ends: nine stalls of one wgrad each); one join at the very end.
    late = side is not None and LATE_JOIN

    def weight_side(fn):
        if late:
            return side.run(fn)
        if side is not None:
            side.join()
        return fn()

    dz = dout.reshape(b, t, h, w).contiguous().float()
    # ---- ConvsOut (grouped 1x1, dense-lowered) + tanh
    spec_out = _spec2d(BASE_CH, t, 1)
    cout = _doconv_of(net.ConvsOut[0])
    dz = ops.act_bwd(dz, S["z"], ACT_TANH)                      # * (1 - z^2): prologue-free kernels below
    dwp, _ = wgrad(spec_out, S["h0"], dz)
    gw, _ = weight_side(lambda: ops.doconv_fold_bwd(dwp, *cout.tensors(), t, BASE_CH, 4, 1, out=(cout.W.grad, None) if inplace else None))
    if not inplace:
        grads[id(cout.W)] = gw
    dh = ops.conv_dgrad(spec_out, dz, S["wp_out_d"], tuple(S["h0"].shape))

    def eblock_bwd(lvl, dh):
        ch = BASE_CH << lvl
        spec = _spec2d(ch, ch, 3)
        blocks = net.Decoder[lvl].layers
        pend = []                                    # (packed weight gradient, layer): folded back in two launches per level
        for rb, (hin, y1, w1d, w2d) in zip(reversed(list(blocks)), reversed(S["rec"][lvl])):
            c1, c2 = _doconv_of(rb.main[0]), _doconv_of(rb.main[1])
            dwp2, _ = wgrad(spec, y1, dh)
            pend.append((dwp2, c2))
            dy1 = ops.conv_dgrad(spec, dh, w2d, tuple(y1.shape), mask_y=y1, mask_act=ACT_RELU)   # * relu'(y1) fused
            dwp1, _ = wgrad(spec, hin, dy1)
            pend.append((dwp1, c1))
            dh = ops.conv_dgrad(spec, dy1, w1d, tuple(hin.shape), add=dh)                     # + skip path
        outs = ([cv.W.grad for _, cv in pend], [cv.D.grad for _, cv in pend]) if inplace else None

        def fold_level():                             # behind the level's weight gradients
            r_ = ops.doconv_fold_bwd_batched([g_ for g_, _ in pend], [cv.tensors() for _, cv in pend], ch, ch, outs=outs)
            if inplace and on_level_done is not None:
                on_level_done(lvl)                    # (on the stream that completed the level's gradients)
            return r_

        dWs, dDs = weight_side(fold_level)
        if not inplace:
            for (_, cv), dW_, dD_ in zip(pend, dWs, dDs):
                grads[id(cv.W)], grads[id(cv.D)] = dW_, dD_
        return dh

    def uppos_bwd(i, dr):
        up = net.UP[i]
        hin, v, r, wpd = S["up"][i]
        cin_, cout_ = up.proj.weight.shape[1], up.proj.weight.shape[0]
        spec = _spec2d(cin_, cout_, 1)
        dz, db = ops.act_bwd_bias(dr, r, ACT_RELU, db_out=_grad_target(up.proj.bias, inplace))      # * relu'(r), and its channel sums
        # through the modulation and the upsampling, down to the low resolution: d v, and d pos = sum_c dz * up(v) * s'
        dv, dpos = ops.upmod_bwd(v, up.pos, dz, dpos_out=_grad_target(up.pos, inplace))
        dwp, _ = wgrad(spec, hin, dv)
        gw_ = weight_side(lambda: ops.weight_unpack_grad(dwp, up.proj.weight.reshape(cout_, cin_, 1), out=_grad_target(up.proj.weight, inplace)))
        dx = ops.conv_dgrad(spec, dv, wpd, tuple(hin.shape))
        if not inplace:
            grads[id(up.proj.weight)] = gw_.reshape(up.proj.weight.shape)
            grads[id(up.proj.bias)] = db
            grads[id(up.pos)] = dpos
        return dx

    dh = eblock_bwd(0, dh)
    dh = uppos_bwd(0, dh)
    dh = eblock_bwd(1, dh)
    dh = uppos_bwd(1, dh)
    dx4s = eblock_bwd(2, dh)              # grad of x_4 + res1: flows to both terms
    dh = uppos_bwd(2, dx4s)
    dx8 = eblock_bwd(3, dh)
    dx4 = ops.axpy_(ops.pooldup_bwd(S["x_4"], dx8), dx4s)
    dx2 = ops.pooldup_bwd(S["x_2"], dx4)
    dx_ = ops.pooldup_bwd(S["x_"], dx2)
    # ---- Convsin (grouped 3x3 + repeat_interleave skip, dense-lowered with centre identity)
    spec_in = _spec2d(t, BASE_CH, 3)
    cin = _doconv_of(net.Convsin[0])
    dwp, _ = wgrad(spec_in, S["idw"], dx_)
    gW, gD = weight_side(lambda: ops.doconv_fold_bwd(dwp, *cin.tensors(), BASE_CH, t, 4, 3, out=(cin.W.grad, cin.D.grad) if inplace else None))
    if not inplace:
        grads[id(cin.W)], grads[id(cin.D)] = gW, gD
    didw = ops.conv_dgrad(spec_in, dx_, S["wp_in_d"], tuple(S["idw"].shape))
    da = ops.idw_bwd(didw, S["sel"])
    att = [l.conv for l in net.input.layers]
    prm4 = (att[0].weight, att[0].bias, att[1].weight, att[1].bias)
    g = ops.attn_bwd(S["x0"], *prm4, da, out=[p_.grad for p_ in prm4] if inplace else None)
    if not inplace:
        for prm, gr in zip(prm4, g):
            grads[id(prm)] = gr.reshape(prm.shape)
    if side is not None:
        side.join()
    return grads


class _GeneratorFn(torch.autograd.Function):
    """Autograd face of generator_forward / generator_backward (drop-in users: loss.backward() works as with the reference)."""

    @staticmethod
    def forward(ctx, net: P2IGenerator, masked_frames, masks, *params):
        need_grad = net._grad_on and any(ctx.needs_input_grad[3:])
        z, S = generator_forward(net, masked_frames, masks, need_grad)
        if need_grad:
            ctx.net, ctx.S = net, S
        return z

    @staticmethod
    def backward(ctx, dout):
        net, S = ctx.net, ctx.S
        grads = generator_backward(net, S, dout, inplace=False)
        out = []
        for (name, prm), need in zip(net.named_parameters(), ctx.needs_input_grad[3:]):
            out.append(grads.get(id(prm)) if need else None)
        ctx.S = None
        return (None, None, None, *out)


def fold_generator_state_dict(sd):
    """Training checkpoint (W (O,I/g,k*k), D, D_diag per DO-Conv) -> state_dict of P2IGenerator(inference=True)
    (W (O,I/g,k,k) only): DoW = einsum('ims,ois->oim', D + D_diag, W.reshape(O/g, I, k*k)) reinterpreted as
    (O, I/g, k, k), exactly DOConv2d.forward (deconv_pytorch.py:111-127).  Checkpoint conversion, not the hot path."""
    out = {}
    for k, v in sd.items():
        if k.endswith(".D") or k.endswith(".D_diag"):
            continue
        if k.endswith(".W") and v.dim() == 3:
            O, Ig, mn = v.shape
            ksz = int(round(mn ** 0.5))
            dk = k[:-2] + ".D"
            if dk in sd:
                D = sd[dk] + sd[k[:-2] + ".D_diag"]
                I = D.shape[0]
                g = I // Ig
                dow = torch.einsum("ims,ois->oim", D, v.reshape(O // g, I, mn))
                out[k] = dow.reshape(O, Ig, ksz, ksz).contiguous()
            else:
                out[k] = v.reshape(O, Ig, ksz, ksz).contiguous()
        else:
            out[k] = v
    return out


# ----------------------------------------------------------------------------- discriminator
D2D_LAYERS = [(0, 64, 1), (2, 128, 2), (4, 256, 2), (6, 256, 1), (8, 1, 1)]           # p2igan.py:120-130
D3D_LAYERS = [(0, 32, 3, (1, 2, 2), 1), (2, 64, 3, (1, 2, 2), 1), (4, 128, 3, (1, 2, 2), 1),
              (6, 128, 3, (2, 1, 1), 1), (8, 1, 1, (1, 1, 1), 0)]                      # p2igan.py:132-142


class P2IDiscriminator(nn.Module):
    def __init__(self, in_channels: int = 16, init_weights: bool = True):
        super().__init__()
        self.in_channels = in_channels
        sn = nn.utils.spectral_norm            # parameter containers (weight_orig / weight_u / weight_v / bias)
        mods, cin = [], in_channels
        for n, (_, cout, s) in enumerate(D2D_LAYERS):
            mods.append(sn(nn.Conv2d(cin, cout, kernel_size=3, stride=s, padding=1)))
            if n < 4:
                mods.append(nn.LeakyReLU(0.2, True))
            cin = cout
        self.d2d = nn.Sequential(*mods)
        mods, cin = [], 1
        for n, (_, cout, k, st, p) in enumerate(D3D_LAYERS):
            mods.append(sn(nn.Conv3d(cin, cout, kernel_size=k, stride=st, padding=p)))
            if n < 4:
                mods.append(nn.LeakyReLU(0.2, True))
            cin = cout
        self.d3d = nn.Sequential(*mods)
        self.alpha2d = nn.Parameter(torch.tensor(0.0))
        self.alpha3d = nn.Parameter(torch.tensor(0.0))      # declared but unused by forward (p2igan.py:145,170)
        self.debug_taps = None
        self._pack_pool, self._pack_turn = {}, 0            # reusable packed-weight buffers (discriminator_forward)
        self._x6_wants = {}                                 # (input dims, ...) -> which packed tensors the split-pipe kernels will take
        if init_weights:
            self.init_weights()
        self.specs2d, cin = [], in_channels
        for _, cout, s in D2D_LAYERS:
            self.specs2d.append(_spec2d(cin, cout, 3, s))
            cin = cout
        self.specs3d, cin = [], 1
        for _, cout, k, st, p in D3D_LAYERS:
            self.specs3d.append(ConvSpec(cin, cout, (k, k, k), st, (p, p, p)))
            cin = cout

    def init_weights(self):
        """p2igan.py:150-155 (writes through the spectral-norm `weight` alias into weight_orig)."""
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Conv3d)):
                init.kaiming_normal_(m.weight_orig.data, a=0.2, nonlinearity="leaky_relu")
                if m.bias is not None:
                    init.zeros_(m.bias)

    def _layer_in_dims(self, b, cin0, t, h, w, spec):
        """(T, H, W) of the INPUT of layer `spec` for a (b, t, 1, h, w) clip (2-D branch: frames are channels, T = 1)."""
        if spec in self.specs2d:
            dims = (1, h, w)
            for sp in self.specs2d:
                if sp is spec:
                    return dims
                dims = sp.out_dims(*dims)
        dims = (t, h, w)
        for sp in self.specs3d:
            if sp is spec:
                return dims
            dims = sp.out_dims(*dims)
        raise RuntimeError("layer not found")

    def layers(self):
        l2 = [self.d2d[i] for i, _, _ in D2D_LAYERS]
        l3 = [self.d3d[i] for i, *_ in D3D_LAYERS]
        return l2, l3

    def forward(self, x):
        l2, l3 = self.layers()
        params = []
        for m in l2 + l3:
            params += [m.weight_orig, m.bias]
        params.append(self.alpha2d)
        self._grad_on = torch.is_grad_enabled()
        return _DiscriminatorFn.apply(self, x, *params)


def discriminator_prepare(net: "P2IDiscriminator", xshape, device, need_x: bool, need_p: bool, pool: bool = False):
    """The parameter-only part of one discriminator forward: ONE spectral-norm power iteration of all ten layers (u, v updated in
    place, as torch does in every training forward), weight / sigma packed for the conv engine, and the bf16 split images.  Calls
    must come in the order of the forwards they serve (each advances u, v).  TrainEngine prepares the fake and the real pass of
    the D step on a side stream while the generator's forward runs; discriminator_forward(prep=...) consumes the result."""
    b, t, c, h, w = xshape
    l2, l3 = net.layers()
    training = net.training

    # power iteration of all ten layers in 4 launches (parameters only; u, v are updated in place as torch does)
    sn = ops.spectral_norm_batched([m.weight_orig for m in l2 + l3], [m.weight_u for m in l2 + l3],
                                   [m.weight_v for m in l2 + l3], training, snapshot=need_p)
    sig_all, usn, vsn = sn if need_p else (sn, None, None)
    sig_of = {id(m): sg for m, sg in zip(l2 + l3, sig_all)}
    uv_of = {id(m): (usn[i], vsn[i]) for i, m in enumerate(l2 + l3)} if need_p else {}
    # weight / sigma, packed for the conv engine: all ten layers in ONE launch.  The packed buffers come from a small pool
    # of zero-initialised buffers (the pack kernel rewrites exactly the non-padding entries); three forwards can be pending
    # in one train step (fake, real, fake-for-G), hence the round-robin of four.
    wflats = [m.weight_orig.reshape(m.weight_orig.shape[0], m.weight_orig.shape[1], -1) for m in l2 + l3]
    need_d = need_x or need_p
    pbuf = None
    if pool:
        bufs = net._pack_pool.setdefault((need_d, str(device)), [None] * 4)
        net._pack_turn = (net._pack_turn + 1) % 4
        if bufs[net._pack_turn] is None:
            tot = sum(wf.shape[2] * wf.shape[1] * ops.pad32(wf.shape[0]) + (wf.shape[2] * wf.shape[0] * ops.pad32(wf.shape[1]) if need_d else 0)
                      for wf in wflats)
            bufs[net._pack_turn] = ops.zero_(torch.empty(tot, device=device, dtype=torch.float32))
        pbuf = bufs[net._pack_turn]
    packed = ops.weight_pack_batched(wflats, sig_all, need_d=need_d, buf=pbuf)
    pack_of = {id(m): pk for m, pk in zip(l2 + l3, packed)}
    # bf16 split of every packed tensor a conv call of this pass (forward now, data gradient later) will take on the split pipe:
    # one launch for all of them instead of one per call
    wkey = (b, t, h, w, need_d, ops.CONV_ENGINE, os.environ.get("P2I_X6C_MIN_WG"))
    wants = net._x6_wants.get(wkey)
    if wants is None:
        wants = []
        for spec in net.specs2d + net.specs3d:
            dsc = spec.desc(b, *net._layer_in_dims(b, None, t, h, w, spec))
            wants += [spec.cin % 16 == 0 and bool(ops._hip.load().p2i_x6c_would_take(dsc, 0, ACT_LEAKY)),
                      need_d and spec.cout % 16 == 0 and spec.cin > 1 and bool(ops._hip.load().p2i_x6c_would_take(dsc, 1, ACT_NONE))]
        net._x6_wants[wkey] = wants
    tens = [t_ for pk in packed for t_ in pk]
    if pool:
        sb = net._pack_pool.setdefault(("x6", need_d, str(device)), [None] * 4)
        sb[net._pack_turn] = ops.x6_presplit(tens, wants, buf=sb[net._pack_turn])
    else:
        ops.x6_presplit(tens, wants)
    return dict(sig_of=sig_of, uv_of=uv_of, pack_of=pack_of, key=(tuple(xshape), need_x, need_p))


def discriminator_forward(net: "P2IDiscriminator", x, need_x: bool, need_p: bool, pool: bool = False, prep=None):
    """P2IDiscriminator.forward (p2igan.py:157-173) incl. the spectral-norm power iteration, as a plain function: returns
    (logits (B, H/4*W/4), saved context or None).  pool=True (TrainEngine only: at most three forwards pending): the packed
    weights live in a round-robin of four reusable buffers instead of a fresh zero-filled one per call.  prep: this forward's
    discriminator_prepare result when the caller has run it ahead of time."""
    b, t, c, h, w = x.shape
    if c * t != net.in_channels:
        raise RuntimeError(f"discriminator expects {net.in_channels} frames, got {t}x{c}")
    xin = x.contiguous().float()
    l2, l3 = net.layers()
    if prep is None:
        prep = discriminator_prepare(net, (b, t, c, h, w), xin.device, need_x, need_p, pool)
    elif prep["key"] != ((b, t, c, h, w), need_x, need_p):
        raise RuntimeError("discriminator_forward: prep was made for another call")
    sig_of, uv_of, pack_of = prep["sig_of"], prep["uv_of"], prep["pack_of"]

    def branch(layers, specs, inp):
        recs, cur = [], inp
        for n, (m, spec) in enumerate(zip(layers, specs)):
            sigma = sig_of[id(m)]
            wp_f, wp_d = pack_of[id(m)]
            act = ACT_LEAKY if n < 4 else ACT_NONE
            y = ops.conv_fwd(spec, cur, wp_f, bias=m.bias, act=act)
            recs.append(dict(x=cur, y=y, wp_d=wp_d, sigma=sigma, act=act,
                             u=uv_of[id(m)][0] if need_p else None, v=uv_of[id(m)][1] if need_p else None))
            cur = y
        return cur, recs

    o2, r2 = branch(l2, net.specs2d, xin.view(b, t * c, h, w))
    o3, r3 = branch(l3, net.specs3d, xin.view(b, c, t, h, w))      # permute(0,2,1,3,4) with c == 1 is a view
    fused = ops.dtail_fwd(o2, o3, net.alpha2d.reshape(1))
    if net.debug_taps is not None:
        net.debug_taps.update(out2d=o2, out3d=o3)
    ctx = dict(r2=r2, r3=r3, xshape=(b, t, c, h, w)) if (need_x or need_p) else None
    return fused, ctx


def discriminator_backward(net: "P2IDiscriminator", ctx, dfused, need_x: bool, needs=None, inplace: bool = False,
                           accumulate: bool = False, dx_add=None):
    """Backward of discriminator_forward.  needs: per-parameter flags in the order [w0, b0, w1, b1, ..., alpha2d] (None: all /
    none according to ctx).  Returns (dx or None, gw {layer: dW}, gb {layer: db}, dalpha).  inplace=True (TrainEngine): the
    gradients are ADDED into the parameters' .grad views (zeroed by the caller before the first backward of a D step;
    accumulate=True on the second one, whose weight gradients pass through a different sigma / u / v).  dx_add: a tensor shaped
    like the input that is added to dx (the reconstruction loss's gradient w.r.t. the frames: fused into the first layer's dgrad)."""
    r2, r3 = ctx["r2"], ctx["r3"]
    b, t, c, h, w = ctx["xshape"]
    l2, l3 = net.layers()
    nl = len(l2) + len(l3)
    if needs is None:
        needs = [r2[0]["u"] is not None] * (2 * nl + 1)
    need_alpha = needs[2 * nl]
    o2, o3 = r2[-1]["y"], r3[-1]["y"]
    da_t = net.alpha2d.grad.reshape(1) if (inplace and need_alpha) else None
    d2, d3, da = ops.dtail_bwd(o2, tuple(o3.shape), net.alpha2d.reshape(1), dfused.contiguous().float(), need_alpha=need_alpha, da_out=da_t)
    gw, gb = {}, {}
    pend = []                 # (layer index, packed weight gradient, flat weight, record, shape, module): unpacked together at the end
    arena = None
    if any(needs[:2 * nl]):
        tot = sum(sp.ntaps * sp.cin * ops.pad32(sp.cout) + sp.cout + 8 for sp in net.specs2d + net.specs3d)
        if inplace and (getattr(net, "_arena_buf", None) is None or net._arena_buf.numel() < tot or net._arena_buf.device != dfused.device):
            net._arena_buf = torch.empty(tot, device=dfused.device, dtype=torch.float32)
        arena = ops.ZeroArena(tot, dfused.device, buf=net._arena_buf if inplace else None)
    side = _side_of(net, dfused.device) if (inplace and arena is not None) else None

    def branch_bwd(layers, specs, recs, dy, base, first_add=None):
        # dy arrives already multiplied by act'(y_n): the dgrad of layer n+1 applies it in its epilogue
        # (mask_y = that layer's input = y_n), so no kernel here needs an activation prologue
        for n in reversed(range(len(layers))):
            m, spec, rc = layers[n], specs[n], recs[n]
            if needs[2 * (base + n)] or needs[2 * (base + n) + 1]:
                if side is None:
                    dwp, db = ops.conv_wgrad(spec, rc["x"], dy, want_bias=True, arena=arena, db_out=_grad_target(m.bias, inplace))
                else:
                    tgt = arena.take(spec.wp_f_shape())
                    dwp, db = side.run(lambda sp=spec, xx=rc["x"], gg=dy, tt=tgt, bb=m.bias.grad: ops.conv_wgrad(
                        sp, xx, gg, want_bias=True, dwp_out=tt, db_out=bb), rc["x"], dy)
                wo = m.weight_orig
                pend.append((base + n, dwp, wo.reshape(wo.shape[0], wo.shape[1], -1), rc, wo.shape, m))
                gb[base + n] = db
            if n > 0 or need_x:
                dy = ops.conv_dgrad(spec, dy, rc["wp_d"], tuple(rc["x"].shape), add=first_add if n == 0 else None,
                                    mask_y=rc["x"] if n > 0 else None, mask_act=ACT_LEAKY)
            else:
                dy = None
        return dy

    add3 = dx_add.reshape(b, c, t, h, w).contiguous() if (dx_add is not None and need_x) else None
    dx3 = branch_bwd(l3, net.specs3d, r3, d3, len(l2), first_add=add3)
    dx = branch_bwd(l2, net.specs2d, r2, d2, 0, first_add=dx3.view(b, t * c, h, w) if dx3 is not None else None)
    if side is not None:
        side.join()
    if pend:                  # d(weight_orig) through weight / sigma for every layer, in two launches
        outs = [p_[5].weight_orig.grad for p_ in pend] if inplace else None
        dws = ops.weight_unpack_grad_batched([p_[1] for p_ in pend], [p_[2] for p_ in pend], [p_[2] for p_ in pend],
                                             [p_[3]["sigma"] for p_ in pend], [p_[3]["u"] for p_ in pend], [p_[3]["v"] for p_ in pend],
                                             outs=outs, accumulate=inplace and accumulate)
        for p_, dw_ in zip(pend, dws):
            gw[p_[0]] = dw_.reshape(p_[4])
    return (dx.view(b, t, c, h, w) if need_x else None), gw, gb, da


class _DiscriminatorFn(torch.autograd.Function):
    """Autograd face of discriminator_forward / discriminator_backward."""

    @staticmethod
    def forward(ctx, net: P2IDiscriminator, x, *params):
        need_x = net._grad_on and ctx.needs_input_grad[1]
        need_p = net._grad_on and any(ctx.needs_input_grad[2:])
        fused, saved = discriminator_forward(net, x, need_x, need_p)
        if need_x or need_p:
            ctx.net, ctx.saved = net, saved
        return fused

    @staticmethod
    def backward(ctx, dfused):
        net = ctx.net
        need_x = ctx.needs_input_grad[1]
        needs = ctx.needs_input_grad[2:]
        nl = (len(needs) - 1) // 2
        dx, gw, gb, da = discriminator_backward(net, ctx.saved, dfused, need_x, needs=list(needs))
        out = []
        for i in range(nl):
            out.append(gw.get(i) if needs[2 * i] else None)
            out.append(gb.get(i) if needs[2 * i + 1] else None)
        out.append(da.reshape(net.alpha2d.shape) if needs[2 * nl] else None)
        ctx.saved = None
        return (None, dx, *out)
