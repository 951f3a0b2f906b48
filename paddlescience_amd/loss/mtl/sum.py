"""ppsci.loss.mtl.Sum (/root/reference/ppsci/loss/mtl/sum.py:27-60): left fold `+=` over the loss dict in
insertion order.  (GradNorm / NTK: grad_weight.py; AGDA / PCGrad / Relobralo are not implemented.)"""
from .base import LossAggregator


class Sum(LossAggregator):
    should_persist = False

    def __init__(self) -> None:
        self.step = 0

    def __call__(self, losses, step: int = 0):
        assert len(losses) > 0, f"Number of given losses({len(losses)}) can not be empty."
        self.step = step
        total = 0.0
        for i, key in enumerate(losses):
            total = losses[key] if i == 0 else total + losses[key]
        return total
