"""Training utilities (losses) — same import surface as the reference's modules/__init__.py:3."""
from .losses import ReconstructionLoss, gan_loss

__all__ = ["ReconstructionLoss", "gan_loss"]
