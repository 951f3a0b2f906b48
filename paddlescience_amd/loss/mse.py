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
