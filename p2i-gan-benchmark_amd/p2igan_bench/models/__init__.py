"""Model registry — the construction API of the reference (models/__init__.py:13-46).

Only the P2I-GAN family is built on the HIP path; the reference's DeepKriging baselines and the
`simple` fallback are out of this build's scope and raise a clear error."""
from typing import Any, Dict

import torch.nn as nn

from .p2igan import P2IDiscriminator, P2IGenerator


def _model_name(cfg: Dict[str, Any]) -> str:
    return cfg.get("model", {}).get("name", "simple").lower()


def build_generator(cfg: Dict[str, Any]) -> nn.Module:
    name = _model_name(cfg)
    if name == "p2igan":
        return P2IGenerator(cfg)
    raise NotImplementedError(f"model '{name}': only 'p2igan' is implemented on the MI355X path")


def build_discriminator(cfg: Dict[str, Any]) -> nn.Module:
    name = _model_name(cfg)
    if name == "p2igan":
        in_channels = cfg.get("model", {}).get("in_channels", 1)
        data_cfg = cfg.get("data_loader") or cfg.get("data", {}).get("train", {})
        sample_length = data_cfg.get("sample_length", 16)
        return P2IDiscriminator(in_channels=in_channels * sample_length)
    raise NotImplementedError(f"model '{name}': only 'p2igan' is implemented on the MI355X path")


__all__ = ["build_generator", "build_discriminator", "P2IGenerator", "P2IDiscriminator"]
