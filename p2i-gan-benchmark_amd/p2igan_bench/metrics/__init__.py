"""Evaluation metrics (metrics/__init__.py:3-5 of the reference)."""
from .metric import MetricConfig, RainfallMetricSuite

__all__ = ["MetricConfig", "RainfallMetricSuite"]
