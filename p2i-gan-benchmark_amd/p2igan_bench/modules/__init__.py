"""Loss entry points of the training path.

``ReconstructionLoss`` (pooled weighted L1 + temporal KL) and ``gan_loss`` (hinge / lsgan) are the two names the training
script imports from this package; both run through the fused HIP kernels of ``csrc/loss.hip``.
"""
from . import losses as _losses

ReconstructionLoss = _losses.ReconstructionLoss
gan_loss = _losses.gan_loss

__all__ = ["ReconstructionLoss", "gan_loss"]
