"""ppsci.loss.mtl.LossAggregator (/root/reference/ppsci/loss/mtl/base.py:28-68)."""


class LossAggregator:
    should_persist = False

    def __init__(self, model=None) -> None:
        self.model = model
        self.step = 0

    def __call__(self, losses, step: int = 0):
        raise NotImplementedError
