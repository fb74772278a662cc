"""Rainfall evaluation metrics on the HIP path.

Public names follow the reference package (``RainfallMetricSuite``, ``MetricConfig``); the individual metric groups and
the rain-rate ``transform`` are exported too, since this build's tests address them directly.
"""
from . import metric as _metric

RainfallMetricSuite = _metric.RainfallMetricSuite
MetricConfig = _metric.MetricConfig
RegressionMetrics = _metric.RegressionMetrics
CategoricalMetrics = _metric.CategoricalMetrics
FractionalSkillScoreMetric = _metric.FractionalSkillScoreMetric
transform = _metric.transform

__all__ = [
    "RainfallMetricSuite",
    "MetricConfig",
    "RegressionMetrics",
    "CategoricalMetrics",
    "FractionalSkillScoreMetric",
    "transform",
]
