"""ppsci.loss.base.Loss (/root/reference/ppsci/loss/base.py)."""
from typing import Dict, Optional, Union


class Loss:
    def __init__(self, reduction: str, weight: Optional[Union[float, Dict[str, float]]] = None):
        self.reduction = reduction
        self.weight = weight

    def __str__(self):
        return f"{self.__class__.__name__}(reduction={self.reduction}, weight={self.weight})"

    def __call__(self, *args, **kwargs):
        return self.forward(*args, **kwargs)
