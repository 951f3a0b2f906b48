"""ppsci.loss.MSELoss (/root/reference/ppsci/loss/mse.py:27-105).

On the fused path the loss is part of the epilogue kernel; `term_scale` tells the lowering which
scalar multiplies sum_p w_p * area_p * (out_p - label_p)^2 for a key.  `forward` on tensors is the
same arithmetic in torch for host-side use (validators, tests, known answers mse.py:46-68)."""
from typing import Dict, Optional, Union

import torch

from .base import Loss


class MSELoss(Loss):
    def __init__(self, reduction: str = "mean", weight: Optional[Union[float, Dict[str, float]]] = None):
        if reduction not in ["mean", "sum"]:
            raise ValueError(f"reduction should be 'mean' or 'sum', but got {reduction}")
        super().__init__(reduction, weight)

    def key_weight(self, key: str) -> float:
        if isinstance(self.weight, (float, int)):
            return float(self.weight)
        if isinstance(self.weight, dict) and key in self.weight:
            return float(self.weight[key])
        return 1.0

    def term_scale(self, key: str, n_global: int) -> float:
        return self.key_weight(key) / (n_global if self.reduction == "mean" else 1.0)

    def forward(self, output_dict, label_dict, weight_dict=None) -> Dict[str, torch.Tensor]:
        losses = {}
        for key in label_dict:
            loss = (output_dict[key] - label_dict[key]) ** 2
            if weight_dict and key in weight_dict:
                loss = loss * weight_dict[key]
            if "area" in output_dict:
                loss = loss * output_dict["area"]
            loss = loss.sum() if self.reduction == "sum" else loss.mean()
            if isinstance(self.weight, (float, int)):
                loss = loss * self.weight
            elif isinstance(self.weight, dict) and key in self.weight:
                loss = loss * self.weight[key]
            losses[key] = loss
        return losses


def _mse_value_and_grad(self, y_net, label, key):
    """Operator-learning path (operator_engine.OperatorConstraint): the loss on a [B, C, H, W] network output and its adjoint
    from the field-loss kernels (loss/field.py, mode "sq"): weight * sum or mean of (y - label)^2."""
    from . import field

    plan = self.__dict__.setdefault("_field_plan", field.FieldLossPlan(0, field.SQ))
    coef = self.key_weight(key) / (y_net.numel() if self.reduction == "mean" else 1.0)
    loss, g = plan.value_and_grad(y_net, label, coef)
    return {key: loss.reshape(())}, g


MSELoss.value_and_grad = _mse_value_and_grad


class CausalMSELoss(MSELoss):
    """mse.py:109-189: the batch is `n_chunks` consecutive time windows; window i is weighted with
    exp(-tol * sum of the mean losses of the windows before it) (a constant w.r.t. the parameters).
    On the fused path: one extra value-only epilogue pass + ppsci_causal_weights per key (engine.py)."""

    def __init__(self, n_chunks: int, reduction: str = "mean", weight: Optional[Union[float, Dict[str, float]]] = None,
                 tol: float = 1.0):
        if n_chunks <= 0:
            raise ValueError(f"n_chunks should be positive, but got {n_chunks}")
        super().__init__(reduction, weight)
        self.n_chunks, self.tol = n_chunks, tol
        self.causal = {"n_chunks": n_chunks, "tol": tol}
        self.acc_mat = torch.tril(torch.ones(n_chunks, n_chunks), -1)

    value_and_grad = None  # (fields are not time-windowed batches)

    def forward(self, output_dict, label_dict, weight_dict=None) -> Dict[str, torch.Tensor]:
        losses = {}
        for key in label_dict:
            loss = (output_dict[key] - label_dict[key]) ** 2
            if weight_dict and key in weight_dict:
                loss = loss * weight_dict[key]
            if "area" in output_dict:
                loss = loss * output_dict["area"]
            loss_t = loss.reshape(self.n_chunks, -1)
            weight_t = torch.exp(-self.tol * (self.acc_mat.to(loss_t) @ loss_t.mean(-1, keepdim=True)))
            loss = loss_t * weight_t.detach()
            loss = loss.sum() if self.reduction == "sum" else loss.mean()
            if isinstance(self.weight, (float, int)):
                loss = loss * self.weight
            elif isinstance(self.weight, dict) and key in self.weight:
                loss = loss * self.weight[key]
            losses[key] = loss
        return losses
