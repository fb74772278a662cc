"""ReconstructionLoss / gan_loss with the reference's call signatures (losses.py:32-48, 232-253),
computed by fused HIP loss+gradient kernels (p2i_recloss, p2i_gan_loss)."""
from __future__ import annotations

import torch

from .. import ops


class DeviceScalar:
    """A loss component that stays on the device until someone asks for a float (the reference
    returns python floats here, which costs a host sync per step: train.py:245)."""

    def __init__(self, t: torch.Tensor):
        self._t = t

    def __float__(self):
        return float(self._t)

    def tensor(self):
        return self._t

    def __repr__(self):
        return f"DeviceScalar({float(self._t):.6g})"

    def __format__(self, spec):
        return format(float(self._t), spec)


class _RecLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, k1_alpha):
        out3, dpred = ops.recloss(pred.contiguous().float(), target.contiguous().float(), k1_alpha)
        ctx.save_for_backward(dpred)
        ctx.mark_non_differentiable(out3)
        return out3[2].clone(), out3

    @staticmethod
    def backward(ctx, gloss, _g3):
        (dpred,) = ctx.saved_tensors
        return dpred * gloss, None, None


class ReconstructionLoss:
    """Weighted L1 + k1_alpha * KL of temporal-difference softmaxes (losses.py:32-48)."""

    def __init__(self, k1_alpha: float = 0.0):
        self.k1_alpha = k1_alpha

    def __call__(self, prediction: torch.Tensor, target: torch.Tensor, mask: torch.Tensor | None = None):
        loss, out3 = _RecLossFn.apply(prediction, target, float(self.k1_alpha))
        return loss, {"pool": DeviceScalar(out3[0]), "reg": DeviceScalar(out3[1])}


class _GanDFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits_real, logits_fake, loss_type, real_label, fake_label):
        loss, da, db = ops.gan_loss_d(logits_real.contiguous(), logits_fake.contiguous(), loss_type, real_label, fake_label)
        ctx.save_for_backward(da, db)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        da, db = ctx.saved_tensors
        return da * g, db * g, None, None, None


class _GanGFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, weight, loss_type, real_label):
        loss, da = ops.gan_loss_g(logits.contiguous(), weight, loss_type, real_label)
        ctx.save_for_backward(da)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (da,) = ctx.saved_tensors
        return da * g, None, None, None


def discriminator_loss(logits_real, logits_fake, loss_type="hinge", target_real_label=1.0, target_fake_label=0.0):
    """0.5 * (L(real) + L(fake)) of train.py:266-283 in one fused kernel."""
    return _GanDFn.apply(logits_real, logits_fake, loss_type, target_real_label, target_fake_label)


def generator_adv_loss(logits, weight, loss_type="hinge", target_real_label=1.0):
    """gan_loss(logits, True, is_disc=False) * adversarial_weight of train.py:301-308."""
    return _GanGFn.apply(logits, float(weight), loss_type, target_real_label)


def gan_loss(logits: torch.Tensor, target_is_real: bool, *, loss_type: str = "nsgan", is_disc: bool = False,
             target_real_label: float = 1.0, target_fake_label: float = 0.0) -> torch.Tensor:
    """Reference-compatible single-term helper (losses.py:232-253): hinge / lsgan / nsgan.  'nsgan' is nn.BCELoss on the RAW
    logits (losses.py:201-202) and, like torch, raises for any logit outside [0, 1]."""
    if loss_type not in ("hinge", "lsgan", "nsgan"):
        raise ValueError(f"Unsupported GAN loss type: {loss_type}")
    if loss_type == "hinge" and is_disc is None:
        raise ValueError("`is_disc` must be set when using hinge loss.")
    if loss_type == "hinge" and not is_disc:
        return generator_adv_loss(logits, 1.0, "hinge", target_real_label)
    if loss_type in ("lsgan", "nsgan"):
        label = target_real_label if target_is_real else target_fake_label
        return generator_adv_loss(logits, 1.0, loss_type, label)       # mean((x-label)^2) / mean(BCE(x, label))
    # hinge discriminator single term: relu(1 -/+ x).mean() == 2 * D-loss with the other side saturated
    big = torch.full_like(logits, 1e30)
    if target_is_real:
        return 2.0 * discriminator_loss(logits, -big, "hinge")
    return 2.0 * discriminator_loss(big, logits, "hinge")
