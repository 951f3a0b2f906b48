"""ppsci.loss.{PeriodicMSELoss, PeriodicL1Loss, PeriodicL2Loss} (/root/reference/ppsci/loss/mse.py:269-355,
l1.py:123-218, l2.py:118-207): the batch is [points on one side ; their periodic images] (PeriodicConstraint) and the
loss compares the first half of every output with its second half, d_i = out_i - out_{i+n}.

Fused path: the pair term f(|d_i|) is symmetric, so with the *partner's detached value* written into each point's
label slot, the ordinary per-point epilogue loss over all 2n points has exactly the pair loss's gradient
(out_i gets f'(d_i), out_{i+n} gets -f'(d_i)) and twice its value.  The engine therefore runs one value-only epilogue
pass, swaps the halves of the values into the label rows, runs the ordinary pass, and halves the reported term
(engine.FusedConstraint.set_periodic).  `term_scale` is taken over the n pairs, not the 2n points."""
from typing import Dict

import torch

from .l1l2 import L1Loss, L2Loss
from .mse import MSELoss


def _halves(x):
    n = len(x)
    if n % 2 > 0:
        raise ValueError(f"Length of output({n}) should be even.")
    return x[:n // 2], x[n // 2:]


class _Periodic:
    periodic = True

    def term_scale(self, key: str, n_global: int) -> float:
        return self.key_weight(key) / (n_global // 2 if self.reduction == "mean" else 1.0)

    def forward(self, output_dict, label_dict, weight_dict=None) -> Dict[str, torch.Tensor]:
        losses = {}
        for key in label_dict:
            a, b = _halves(output_dict[key])
            loss = self._pair(a, b)
            if weight_dict and key in weight_dict:
                loss = loss * weight_dict[key]  # (the reference multiplies the [n] terms by the [2n] weights: only None works)
            if "area" in output_dict:
                loss = loss * output_dict["area"]
            loss = self._over_features(loss)
            loss = loss.sum() if self.reduction == "sum" else loss.mean()
            losses[key] = loss * self.key_weight(key)
        return losses

    def _over_features(self, loss):
        return loss


class PeriodicMSELoss(_Periodic, MSELoss):
    def _pair(self, a, b):
        return (a - b) ** 2


class PeriodicL1Loss(_Periodic, L1Loss):
    def _pair(self, a, b):
        return (a - b).abs()

    def _over_features(self, loss):
        return loss.sum(dim=1)


class PeriodicL2Loss(_Periodic, L2Loss):
    def _pair(self, a, b):
        return (a - b) ** 2

    def _over_features(self, loss):
        return loss.sum(dim=1).sqrt()
