"""ppsci.loss.FunctionalLoss (/root/reference/ppsci/loss/func.py:27-92): the loss is a user function of
(output_dict, label_dict, weight_dict) returning a dict of scalar tensors.  Usable on the operator-learning
path: a loss object with kernels of its own hands over value and adjoint (`value_and_grad`), any other function of the
network OUTPUT is differentiated by torch w.r.t. that one tensor; the fused PINN kernels need a closed-form loss (MSELoss)."""
from typing import Callable, Dict, Optional, Union

import torch

from .base import Loss


class FunctionalLoss(Loss):
    def __init__(self, loss_expr: Callable[..., Dict[str, torch.Tensor]],
                 weight: Optional[Union[float, Dict[str, float]]] = None):
        super().__init__(None, weight)
        self.loss_expr = loss_expr
        if hasattr(loss_expr, "value_and_grad"):
            # a loss object with its own kernels for value AND adjoint (LpLoss_train / H1Loss_train, loss/lp_h1.py): the
            # operator engine takes dL/d(network output) from it instead of from an autograd tape
            self.value_and_grad = self._value_and_grad

    def _value_and_grad(self, y_net, label, key):
        losses, g = self.loss_expr.value_and_grad(y_net, label, key)
        w = self.weight if isinstance(self.weight, (float, int)) else (self.weight or {}).get(key) if isinstance(self.weight, dict) else None
        if w is not None:
            losses, g = {k: v * w for k, v in losses.items()}, g * w
        return losses, g

    def forward(self, output_dict, label_dict=None, weight_dict=None) -> Dict[str, torch.Tensor]:
        losses = self.loss_expr(output_dict, label_dict, weight_dict)
        assert isinstance(losses, dict), ("Loss computed by custom function should be type of 'dict', "
                                          f"but got {type(losses)}. Please check the return type of custom loss function.")
        for key in losses:
            if isinstance(self.weight, (float, int)):
                losses[key] = losses[key] * self.weight
            elif isinstance(self.weight, dict) and key in self.weight:
                losses[key] = losses[key] * self.weight[key]
        return losses
