"""CPU restatement (torch-CPU fp32, functional form) of the P2I-GAN hot path.

TEST INFRASTRUCTURE — see ``oracle/__init__.py``.  Every function cites the file:line of
``NTU-CompHydroMet-Lab/P2I-GAN-benchmark`` (paths relative to the reference root) whose
arithmetic it restates.  Parameters are plain dicts ``{state_dict key: tensor}`` using the
reference's own key names, so a reference checkpoint can be fed in directly.

The backward pass of the oracle is torch autograd applied to this restated forward.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]

BASE_CH = 64          # p2igan.py:46 (T = 16).  T != 16 has no reference behaviour (layer.py:310 raises); the build's
                      # generalisation base = 4*T is restated below so that the HIP path has a checker:
                      # "parity unpinned -- self-consistency only" for T != 16.
NUM_RES = 4           # p2igan.py:24
IDW_K, IDW_RHO, IDW_TAU, IDW_CHUNK = 4, 2.0, 0.05, 16384   # p2igan.py:44
# "c": pinned host-independent selection (parity checks); "torch": literal cdist/topk as the reference
# executes it (used when TIMING the CPU baseline, bench.py)
IDW_IMPL = "c"


# --------------------------------------------------------------------------- DO-Conv
def doconv_fold(W: torch.Tensor, D: Optional[torch.Tensor], D_diag: Optional[torch.Tensor],
                out_ch: int, in_ch: int, groups: int, ksz: int) -> torch.Tensor:
    """DoW of deconv_pytorch.py:111-127: einsum('ims,ois->oim', D+D_diag, W.reshape(O/g, I, s))
    reshaped (memory reinterpretation) to (O, I/g, k, k); the 1x1 case is W.reshape."""
    shape = (out_ch, in_ch // groups, ksz, ksz)
    if ksz * ksz > 1:
        Dm = D + D_diag
        Wr = W.reshape(out_ch // groups, in_ch, W.shape[-1])
        return torch.einsum("ims,ois->oim", Dm, Wr).reshape(shape)
    return W.reshape(shape)


def doconv(p: Params, prefix: str, x: torch.Tensor, out_ch: int, in_ch: int, groups: int, ksz: int):
    """DOConv2d.forward, deconv_pytorch.py:111-132 (stride 1, pad k//2, no bias: layer.py:78)."""
    W = p[prefix + ".W"]
    D = p.get(prefix + ".D")
    Dd = p.get(prefix + ".D_diag")
    w = doconv_fold(W, D, Dd, out_ch, in_ch, groups, ksz)
    return F.conv2d(x, w, None, 1, ksz // 2, 1, groups)


# --------------------------------------------------------------------------- input block
def attention_block(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """AttentionBlock.forward, layer.py:301-304: x is (P, 16, 1); relu(x + x*conv1d(x))."""
    gate = F.conv1d(x, weight, bias)
    return F.relu(x + x * gate)


_GRID: Dict[Tuple[int, int, int], torch.Tensor] = {}


def grid_points(D: int, H: int, W: int) -> torch.Tensor:
    """_get_grid_points, layer.py:246-256: (Q,3) of (x,y,z) in [0,1], z slowest."""
    key = (D, H, W)
    if key not in _GRID:
        z = torch.linspace(0, 1, D)
        y = torch.linspace(0, 1, H)
        x = torch.linspace(0, 1, W)
        gz, gy, gx = torch.meshgrid(z, y, x, indexing="ij")
        _GRID[key] = torch.stack([gx, gy, gz], dim=-1).reshape(-1, 3).contiguous()
    return _GRID[key]


def idw_3d_knn_torch(points: torch.Tensor, values: torch.Tensor, shape: Tuple[int, int, int],
                     k: int = IDW_K, tau: float = IDW_TAU, chunk: int = IDW_CHUNK, return_sel: bool = False):
    """idw_3d_knn, layer.py:259-293 on the CPU fp32 path (rho == 2 branch), literally through
    torch.cdist / torch.topk.  NOTE: cdist's sgemm rounding (hence which of two mathematically
    equidistant gauges (t-1 / t+1) wins rank 4) depends on the MKL code path of the HOST CPU:
    Intel AVX-512 hosts accumulate a k-ordered fmaf chain, the MI355X box's EPYC host does not.
    Used for documentation and for timing the CPU baseline; parity uses ``idw_3d_knn`` below."""
    D, H, W = shape
    gp_all = grid_points(D, H, W)
    Q = gp_all.shape[0]
    out = torch.empty(Q, dtype=torch.float32)
    idx_all = torch.empty(Q, k, dtype=torch.int64) if return_sel else None
    for s in range(0, Q, chunk):
        e = min(s + chunk, Q)
        d = torch.cdist(gp_all[s:e], points)
        d_k, i_k = torch.topk(d, k, dim=1, largest=False)
        v_k = values[i_k]
        inv = 1.0 / (d_k + tau)
        w = inv * inv
        w = w / (w.sum(dim=1, keepdim=True) + 1e-12)
        out[s:e] = (v_k * w).sum(dim=1)
        if return_sel:
            idx_all[s:e] = i_k
    out = out.reshape(D, H, W)
    return (out, idx_all) if return_sel else out


_CLIB = None


def _c_lib():
    """oracle/_build/libp2i_oracle.so (oracle/idw_knn.c), built on demand with gcc."""
    global _CLIB
    if _CLIB is None:
        import ctypes
        import os
        import subprocess
        here = os.path.dirname(os.path.abspath(__file__))
        so = os.path.join(here, "_build", "libp2i_oracle.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", here])
        _CLIB = ctypes.CDLL(so)
    return _CLIB


def idw_select_c(points: torch.Tensor, shape: Tuple[int, int, int], tau: float = IDW_TAU):
    """4-NN selection by oracle/idw_knn.c: the host-independent restatement of cdist's fmaf chain +
    topk's partial_sort heap as captured from the reference on an Intel AVX-512 host (goldens).
    Returns (sel int64 (Q,4), d float32 (Q,4)) sorted by ascending distance."""
    import ctypes
    D, H, W = shape
    Q = D * H * W
    gx, gy, gz = (torch.linspace(0, 1, n) for n in (W, H, D))
    pts = points.detach().float().contiguous()
    N = pts.shape[0]
    if N < 4:
        raise RuntimeError("selected index k out of range")      # torch.topk(k=4) on fewer than 4 points
    out = torch.empty(Q, dtype=torch.float32)
    sel = torch.empty(Q, 4, dtype=torch.int32)
    seld = torch.empty(Q, 4, dtype=torch.float32)
    dummy = torch.zeros(N, dtype=torch.float32)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    _c_lib().idw_knn4(P(gx), P(gy), P(gz), D, H, W, P(pts), P(dummy), N, ctypes.c_float(tau), P(out), P(sel), P(seld))
    return sel.long(), seld


def idw_3d_knn(points: torch.Tensor, values: torch.Tensor, shape: Tuple[int, int, int],
               k: int = IDW_K, tau: float = IDW_TAU, chunk: int = IDW_CHUNK, return_sel: bool = False):
    """idw_3d_knn, layer.py:259-293: selection by the pinned C restatement, weights/values in torch
    (differentiable w.r.t. ``values`` exactly like the reference: weights are data)."""
    assert k == 4
    D, H, W = shape
    sel, d_k = idw_select_c(points, shape, tau)
    v_k = values[sel]
    inv = 1.0 / (d_k + tau)
    w = inv * inv
    w = w / (w.sum(dim=1, keepdim=True) + 1e-12)
    out = (v_k * w).sum(dim=1).reshape(D, H, W)
    return (out, sel) if return_sel else out


def mask_points(mask_b: torch.Tensor):
    """layer.py:329-342: nonzero(mask>0) in (t,y,x) row-major order -> normalised (x,y,z)."""
    D, H, W = mask_b.shape
    tz, ty, tx = torch.nonzero(mask_b > 0, as_tuple=True)
    pts = torch.stack([tx.float() / max(W - 1, 1), ty.float() / max(H - 1, 1),
                       tz.float() / max(D - 1, 1)], dim=-1)
    return tz, ty, tx, pts


def input_block(p: Params, frames: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """InputBlock.forward, layer.py:316-361 (depth=2)."""
    B, D, H, W = frames.shape
    x = frames.permute(0, 2, 3, 1).contiguous().view(B * H * W, D, 1)
    for i in range(2):
        x = attention_block(x, p[f"input.layers.{i}.conv.weight"], p[f"input.layers.{i}.conv.bias"])
    x = x.view(B, H, W, D).permute(0, 3, 1, 2).contiguous()
    outs = []
    for b in range(B):
        tz, ty, tx, pts = mask_points(mask[b])
        if tz.numel() == 0:                                  # layer.py:330-332
            outs.append(torch.zeros(1, D, H, W))
            continue
        vals = x[b][tz, ty, tx]
        fn = idw_3d_knn if IDW_IMPL == "c" else idw_3d_knn_torch
        outs.append(fn(pts, vals, (D, H, W)).unsqueeze(0))
    return torch.cat(outs, dim=0)


# --------------------------------------------------------------------------- generator glue
def pool_dup(x: torch.Tensor, t: int = 16) -> torch.Tensor:
    """DownsampleDuplicateChannels.forward, layer.py:205-214."""
    b, c, h, w = x.shape
    x = F.max_pool2d(x, 2, 2)
    x = x.view(b * t, c // t, h // 2, w // 2).repeat_interleave(2, dim=1)
    return x.view(b, 2 * c, h // 2, w // 2)


def uppos(p: Params, i: int, x: torch.Tensor) -> torch.Tensor:
    """UPPos.forward, layer.py:392-399."""
    x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    pos = 2 * torch.sigmoid(p[f"UP.{i}.pos"]) - 1
    x = x + x * pos
    x = F.conv2d(x, p[f"UP.{i}.proj.weight"], p[f"UP.{i}.proj.bias"])
    return F.relu(x)


def eblock(p: Params, lvl: int, x: torch.Tensor) -> torch.Tensor:
    """EBlock (p2igan.py:176-183) of 4 ResBlock_do (layer.py:126-135)."""
    c = x.shape[1]
    for r in range(NUM_RES):
        pre = f"Decoder.{lvl}.layers.{r}.main"
        y = F.relu(doconv(p, pre + ".0.main.0", x, c, c, 1, 3))
        y = doconv(p, pre + ".1.main.0", y, c, c, 1, 3)
        x = y + x
    return x


def generator_forward(p: Params, masked_frames: torch.Tensor, masks: torch.Tensor,
                      taps: Optional[dict] = None) -> torch.Tensor:
    """P2IGenerator.forward, p2igan.py:72-112.  (B,T,1,H,W) x2 -> (B,T,1,H,W)."""
    b, t, c, h, w = masked_frames.shape
    mf = masked_frames.reshape(b, c * t, h, w)
    mk = masks.reshape(b, c * t, h, w)
    x = input_block(p, mf, mk).float()
    base = 4 * t                                              # 64 at the reference's T = 16
    x_ = doconv(p, "Convsin.0.main.0", x, base, t, 4, 3) + x.repeat_interleave(4, dim=1)
    x_2 = pool_dup(x_, t)
    x_4 = pool_dup(x_2, t)
    x_8 = pool_dup(x_4, t)
    dec3 = eblock(p, 3, x_8)
    res1 = uppos(p, 2, dec3)
    x_4 = x_4 + res1
    res2 = uppos(p, 1, eblock(p, 2, x_4))
    res3 = uppos(p, 0, eblock(p, 1, res2))
    z = eblock(p, 0, res3)
    z = doconv(p, "ConvsOut.0.main.0", z, t, base, 4, 1)
    if taps is not None:
        taps.update(idw=x, x_=x_, x_8=x_8, dec3=dec3, res1=res1, res3=res3)
    return torch.tanh(z).view(b, t, c, h, w)


# --------------------------------------------------------------------------- discriminator
def spectral_norm_weight(p: Params, prefix: str, training: bool, eps: float = 1e-12):
    """torch.nn.utils.spectral_norm (third-party; call sites layer.py:402-407, p2igan.py:141):
    one power iteration per training-mode forward (u, v updated in place under no_grad),
    sigma = u^T W v, weight = weight_orig / sigma with u, v constants for autograd."""
    w = p[prefix + ".weight_orig"]
    u = p[prefix + ".weight_u"]
    v = p[prefix + ".weight_v"]
    wm = w.reshape(w.shape[0], -1)
    if training:
        with torch.no_grad():
            v_new = F.normalize(torch.mv(wm.t(), u), dim=0, eps=eps)
            u_new = F.normalize(torch.mv(wm, v_new), dim=0, eps=eps)
            u.copy_(u_new)
            v.copy_(v_new)
    uu, vv = u.clone(), v.clone()
    sigma = torch.dot(uu, torch.mv(wm, vv))
    return w / sigma


D2D = [(0, 1), (2, 2), (4, 2), (6, 1), (8, 1)]                      # p2igan.py:120-130
D3D = [(0, (1, 2, 2), 1), (2, (1, 2, 2), 1), (4, (1, 2, 2), 1), (6, (2, 1, 1), 1), (8, (1, 1, 1), 0)]  # :132-142


def discriminator_forward(p: Params, x: torch.Tensor, training: bool = True,
                          taps: Optional[dict] = None) -> torch.Tensor:
    """P2IDiscriminator.forward, p2igan.py:157-173.  (B,T,1,H,W) -> (B, H/4*W/4)."""
    b, t, c, h, w = x.shape
    y = x.reshape(b, t * c, h, w)
    for n, (i, s) in enumerate(D2D):
        wgt = spectral_norm_weight(p, f"d2d.{i}", training)
        y = F.conv2d(y, wgt, p[f"d2d.{i}.bias"], s, 1)
        if n < 4:
            y = F.leaky_relu(y, 0.2)
    z = x.permute(0, 2, 1, 3, 4)
    for n, (i, s, pad) in enumerate(D3D):
        wgt = spectral_norm_weight(p, f"d3d.{i}", training)
        z = F.conv3d(z, wgt, p[f"d3d.{i}.bias"], s, pad)
        if n < 4:
            z = F.leaky_relu(z, 0.2)
    z2 = z.mean(dim=2)
    if z2.shape[-2:] != y.shape[-2:]:
        z2 = F.interpolate(z2, size=y.shape[-2:], mode="bilinear", align_corners=False)
    fused = torch.sigmoid(p["alpha2d"]) * y + z2
    if taps is not None:
        taps.update(out2d=y, out3d=z)
    return fused.view(b, -1)


# --------------------------------------------------------------------------- losses
def weighted_l1(pred: torch.Tensor, true: torch.Tensor) -> torch.Tensor:
    """weighted_l1_distance, losses.py:56-65."""
    a, bb, c, xmax = 0.50, 5.14, 0.12, 0.70
    xm = torch.tensor(xmax, dtype=true.dtype)
    wmax = a * torch.exp(bb * xm) + c
    w = a * torch.exp(bb * true) + c
    w = torch.where(true > xm, wmax, w)
    return torch.mean(w * torch.abs(pred - true))


def temporal_kl(pred: torch.Tensor, true: torch.Tensor, temperature: float = 0.1) -> torch.Tensor:
    """losses.py:41-45 via :68-85: KL(softmax(dT true/0.1) || softmax(dT pred/0.1)), batchmean."""
    pd = pred[:, 1:] - pred[:, :-1]
    td = true[:, 1:] - true[:, :-1]
    sz = pd.shape
    pp = F.softmax(pd.reshape(sz[0], sz[1], -1) / temperature, dim=-1)
    tp = F.softmax(td.reshape(sz[0], sz[1], -1) / temperature, dim=-1)
    return F.kl_div(pp.log(), tp, reduction="batchmean")


def reconstruction_loss(pred, true, k1_alpha: float):
    """ReconstructionLoss.__call__, losses.py:38-48."""
    pool = weighted_l1(pred, true)
    reg = temporal_kl(pred, true)
    return pool + k1_alpha * reg, pool, reg


def gan_loss(logits: torch.Tensor, is_real: bool, loss_type: str = "hinge", is_disc: bool = False,
             real_label: float = 1.0, fake_label: float = 0.0) -> torch.Tensor:
    """AdversarialLoss.forward, losses.py:210-226 (hinge / lsgan; nsgan = BCELoss on raw logits)."""
    if loss_type == "hinge":
        if is_disc:
            return F.relu(1 - logits).mean() if is_real else F.relu(1 + logits).mean()
        return (-logits).mean()
    label = torch.full_like(logits, real_label if is_real else fake_label)
    if loss_type == "lsgan":
        return F.mse_loss(logits, label)
    if loss_type == "nsgan":
        return F.binary_cross_entropy(logits, label)
    raise ValueError(loss_type)


# --------------------------------------------------------------------------- optimiser / step
def adam_step(params: List[torch.Tensor], grads: List[Optional[torch.Tensor]], state: List[dict],
              lr: float, beta1: float, beta2: float, eps: float = 1e-8):
    """torch.optim.Adam (third-party; call site train.py:125-136): no weight decay, no amsgrad."""
    with torch.no_grad():
        for prm, g, st in zip(params, grads, state):
            if g is None:
                continue
            if not st:
                st["step"] = 0
                st["m"] = torch.zeros_like(prm)
                st["v"] = torch.zeros_like(prm)
            st["step"] += 1
            st["m"].mul_(beta1).add_(g, alpha=1 - beta1)
            st["v"].mul_(beta2).addcmul_(g, g, value=1 - beta2)
            bc1 = 1 - beta1 ** st["step"]
            bc2 = 1 - beta2 ** st["step"]
            denom = (st["v"].sqrt() / math.sqrt(bc2)).add_(eps)
            prm.addcdiv_(st["m"], denom, value=-lr / bc1)


def trainable_keys(p: Params) -> List[str]:
    return [k for k in p if not (k.endswith("D_diag") or k.endswith("weight_u") or k.endswith("weight_v"))]


class TrainState:
    """Holds G/D params + Adam state; ``step`` restates Trainer._train_one_epoch's body, train.py:240-326."""

    def __init__(self, gp: Params, dp: Optional[Params], cfg_loss: dict, cfg_opt: dict):
        self.gp = {k: v.clone() for k, v in gp.items()}
        self.dp = {k: v.clone() for k, v in dp.items()} if dp is not None else None
        self.gkeys = trainable_keys(self.gp)
        self.dkeys = trainable_keys(self.dp) if dp is not None else []
        self.gstate = [dict() for _ in self.gkeys]
        self.dstate = [dict() for _ in self.dkeys]
        self.k1 = cfg_loss.get("k1_weight", 0.0)
        self.adv_w = cfg_loss.get("adversarial_weight", 0.01)
        self.gan_type = cfg_loss.get("gan_loss", "hinge")
        self.lr = cfg_opt["lr"]
        self.b1 = cfg_opt.get("beta1", 0.0)
        self.b2 = cfg_opt.get("beta2", 0.99)

    def step(self, frames, masked, masks, keep_grads: bool = False, taps: Optional[dict] = None):
        for k in self.gkeys:
            self.gp[k].requires_grad_(True)
        preds = generator_forward(self.gp, masked, masks, taps)
        loss_g, pool, reg = reconstruction_loss(preds, frames, self.k1)
        out = {"rec": float(loss_g.detach()), "pool": float(pool.detach()), "reg": float(reg.detach())}
        dgrads = None
        if self.dp is not None:
            for k in self.dkeys:
                self.dp[k].requires_grad_(True)
            lf = discriminator_forward(self.dp, preds.detach(), True)
            lr_ = discriminator_forward(self.dp, frames, True)
            loss_d = (gan_loss(lr_, True, self.gan_type, True) + gan_loss(lf, False, self.gan_type, True)) * 0.5
            dgrads = torch.autograd.grad(loss_d, [self.dp[k] for k in self.dkeys], allow_unused=True)
            adam_step([self.dp[k] for k in self.dkeys], list(dgrads), self.dstate, self.lr, self.b1, self.b2)
            for k in self.dkeys:
                self.dp[k].requires_grad_(False)
            lg = discriminator_forward(self.dp, preds, True)
            adv = gan_loss(lg, True, self.gan_type, False) * self.adv_w
            loss_g = loss_g + adv
            out.update(loss_d=float(loss_d.detach()), adv=float(adv.detach()),
                       logits_real=lr_.detach(), logits_fake=lf.detach())
        ggrads = torch.autograd.grad(loss_g, [self.gp[k] for k in self.gkeys], allow_unused=True)
        for k in self.gkeys:
            self.gp[k].requires_grad_(False)
        adam_step([self.gp[k] for k in self.gkeys], list(ggrads), self.gstate, self.lr, self.b1, self.b2)
        out.update(loss_g=float(loss_g.detach()), preds=preds.detach())
        if keep_grads:
            out["ggrads"] = dict(zip(self.gkeys, ggrads))
            out["dgrads"] = dict(zip(self.dkeys, dgrads)) if dgrads is not None else None
        return out


# --------------------------------------------------------------------------- inference
def infer_event(gp: Params, masked: torch.Tensor, masks: torch.Tensor, stride: int = 16,
                overlap: int = 12, output_scale: float = 255.0) -> torch.Tensor:
    """Sliding-window loop of infer.py:188-245 for one event: (1,L,1,H,W) -> (L,1,H,W)."""
    L = masked.shape[1]
    step = max(1, stride - overlap)
    acc = torch.zeros(L, *masked.shape[2:])
    cnt = torch.zeros(L, 1, 1, 1)
    with torch.no_grad():
        for s in range(0, L, step):
            e = s + stride
            if e > L:
                pad = e - L
                fpad = lambda x: torch.cat([x, x[:, -1:].repeat(1, pad, 1, 1, 1)], dim=1)
                cf, cm, valid = fpad(masked[:, s:e]), fpad(masks[:, s:e]), L - s
            else:
                cf, cm, valid = masked[:, s:e], masks[:, s:e], stride
            o = generator_forward(gp, cf, cm)
            acc[s:s + valid] += o[0, :valid]
            cnt[s:s + valid] += 1.0
    return torch.clamp(acc / torch.clamp(cnt, min=1e-5) * output_scale, min=0.0)
