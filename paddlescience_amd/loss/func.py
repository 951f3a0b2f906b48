"""ppsci.loss.FunctionalLoss (/root/reference/ppsci/loss/func.py:27-92): the loss is a user function of
(output_dict, label_dict, weight_dict) returning a dict of scalar tensors.  Usable on the operator-learning
path (torch autograd); the fused PINN kernels need a closed-form loss (MSELoss)."""
from typing import Callable, Dict, Optional, Union

import torch

from .base import Loss


class FunctionalLoss(Loss):
    def __init__(self, loss_expr: Callable[..., Dict[str, torch.Tensor]],
                 weight: Optional[Union[float, Dict[str, float]]] = None):
        super().__init__(None, weight)
        self.loss_expr = loss_expr

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
