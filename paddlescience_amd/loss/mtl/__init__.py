from .base import LossAggregator  # noqa: F401
from .sum import Sum  # noqa: F401

__all__ = ["LossAggregator", "Sum"]
