from .base import LossAggregator  # noqa: F401
from .grad_weight import NTK, GradNorm  # noqa: F401
from .sum import Sum  # noqa: F401

__all__ = ["LossAggregator", "Sum", "GradNorm", "NTK"]
