"""Training engine of the operator-learning path (FNO): forward + loss + backward through torch autograd
around the HIP spectral-convolution kernel, gradients accumulated into the model's flat buffer, one SUM
all-reduce of that buffer per step and the fused Adam kernel on it -- the same contract
(`forward_backward`, `allreduce`, `grad`, `dp_reduce`) as `engine.Engine` for the PINN path.

Mirrors ExpressionSolver.train_forward + train_epoch_func for a supervised constraint
(/root/reference/ppsci/utils/expression.py:60-131, ppsci/solver/train.py:58-213)."""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import numpy as np
import torch
import torch.distributed as dist


def _to_dev(d: Optional[dict], device) -> Dict[str, torch.Tensor]:
    out = {}
    for k, v in (d or {}).items():
        if isinstance(v, torch.Tensor):
            out[k] = v.to(device=device, dtype=torch.float32)
        elif isinstance(v, (int, float)):
            out[k] = v
        else:
            out[k] = torch.as_tensor(np.asarray(v, dtype=np.float32)).to(device)
    return out


class OperatorConstraint:
    """One supervised constraint bound to device tensors."""

    def __init__(self, name: str, model, output_expr: Dict[str, Callable], loss, device, label_keys: List[str],
                 batch_size: int):
        self.name, self.model, self.output_expr, self.loss_fn, self.device = name, model, output_expr, loss, device
        self.label_keys = list(label_keys)
        self.batch_size = batch_size
        self.inp = self.lab = self.w = None
        self._last: Dict[str, torch.Tensor] = {}

    def bind(self, inp, lab, w=None):
        self.inp, self.lab, self.w = _to_dev(inp, self.device), _to_dev(lab, self.device), _to_dev(w, self.device)

    def outputs(self) -> Dict[str, torch.Tensor]:
        out = self.model(self.inp)
        data = {**self.inp, **out}
        return {k: f(data) for k, f in self.output_expr.items()}

    def forward_loss(self) -> torch.Tensor:
        vals = self.outputs()
        losses = self.loss_fn(vals, self.lab, self.w)
        self._last = {k: v.detach() for k, v in losses.items()}
        total = None
        for v in losses.values():  # mtl.Sum: left fold in insertion order
            total = v if total is None else total + v
        return total

    def losses(self) -> Dict[str, float]:
        return {k: float(v) for k, v in self._last.items()}


class OperatorEngine:
    dp_reduce = "mean"  # DataParallel semantics of the reference: gradients are averaged over ranks

    def __init__(self, model):
        self.model = model
        self.grad = model.flat_grad
        self.world = dist.get_world_size() if dist.is_initialized() else 1

    def forward_backward(self, constraints: List[OperatorConstraint]):
        self.grad.zero_()
        for c in constraints:
            c.forward_loss().backward()

    def allreduce(self):
        if self.world > 1:
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM)
