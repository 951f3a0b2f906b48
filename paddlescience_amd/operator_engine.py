"""Training engine of the operator-learning path (FNO): forward + loss + backward through torch autograd
around the HIP spectral-convolution kernel, gradients accumulated into the model's flat buffer, one SUM
all-reduce of that buffer per step and the fused Adam kernel on it -- the same contract
(`forward_backward`, `allreduce`, `grad`, `dp_reduce`) as `engine.Engine` for the PINN path.

Mirrors ExpressionSolver.train_forward + train_epoch_func for a supervised constraint
(/root/reference/ppsci/utils/expression.py:60-131, ppsci/solver/train.py:58-213)."""
from __future__ import annotations

import os
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
            # a Python scalar becomes a 0-d device tensor: bind() can then copy it in place like every other entry, and
            # the captured step keeps one graph instead of re-capturing (and leaking one) per iteration
            out[k] = torch.tensor(float(v), dtype=torch.float32, device=device)
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
        self.version = 0  # bumped when the device tensors are re-allocated
        self._last: Dict[str, torch.Tensor] = {}

    def bind(self, inp, lab, w=None):
        """Upload a batch.  Same-shaped batches are copied into the SAME device tensors, so that a captured step
        (OperatorEngine: HIP graph) keeps reading valid addresses."""
        new = (_to_dev(inp, self.device), _to_dev(lab, self.device), _to_dev(w, self.device))
        old = (self.inp, self.lab, self.w)
        same = all(o is not None and o.keys() == n.keys() and all(
            isinstance(n[k], torch.Tensor) == isinstance(o[k], torch.Tensor)
            and (not isinstance(n[k], torch.Tensor) or n[k].shape == o[k].shape) for k in n) for o, n in zip(old, new))
        if same and all(isinstance(v, torch.Tensor) for d in new for v in d.values()):
            for o, n in zip(old, new):
                for k in n:
                    o[k].copy_(n[k])
        else:
            self.inp, self.lab, self.w = new
            self.version += 1

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

    def forward_backward_native(self, native) -> None:
        """Network forward + backward on the hand-written kernels (fno_engine.FnoNative); only the user's loss
        expression -- arbitrary Python on the network OUTPUT -- is differentiated by torch, which yields dL/dy."""
        m = self.model
        xs = [self.inp[k] for k in m.input_keys]
        x = xs[0] if len(xs) == 1 else torch.cat(xs, dim=1)
        y = native.forward(x).detach().requires_grad_(True)
        data = {**self.inp, m.output_keys[0]: y}
        vals = {k: f(data) for k, f in self.output_expr.items()}
        losses = self.loss_fn(vals, self.lab, self.w)
        self._last = {k: v.detach() for k, v in losses.items()}
        total = None
        for v in losses.values():
            total = v if total is None else total + v
        (gy,) = torch.autograd.grad(total, y)
        native.backward(gy)

    def losses(self) -> Dict[str, float]:
        return {k: float(v) for k, v in self._last.items()}


class OperatorEngine:
    dp_reduce = "mean"  # DataParallel semantics of the reference: gradients are averaged over ranks

    def __init__(self, model):
        self.model = model
        self.grad = model.flat_grad
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        from .engine import StepGraph

        self._step_graph = StepGraph(self.grad.is_cuda and os.environ.get("PPSCI_HIP_GRAPH", "1") != "0")
        # FNO / TFNO in the supported configuration: forward and backward on this framework's own kernels, no autograd
        # graph (fno_engine.py); PPSCI_FNO_NATIVE=0 keeps the torch-autograd path around the spectral kernel
        self.native = None
        if os.environ.get("PPSCI_FNO_NATIVE", "1") != "0":
            from . import fno_engine

            why = fno_engine.supports(model)
            if why is None:
                self.native = fno_engine.FnoNative(model)
            else:
                from .utils import logger

                logger.message(f"FNO: native forward/backward not used ({why}); training through torch autograd")

    def _forward_backward_eager(self, constraints: List[OperatorConstraint]):
        if self.native is not None and len(constraints) == 1:
            return constraints[0].forward_backward_native(self.native)
        self.grad.zero_()
        for c in constraints:
            c.forward_loss().backward()

    def forward_backward(self, constraints: List[OperatorConstraint]):
        # An FNO step is ~200 kernels of ~10 us (FFTs, the spectral contraction, 1x1 convolutions, norms, their
        # backward): launch-bound from Python/autograd, so forward + loss + backward is captured once per batch shape
        # into a HIP graph and replayed (engine.StepGraph; PPSCI_HIP_GRAPH=0 or a failed capture -> eager).
        key = tuple((id(c), c.version) for c in constraints)
        self._step_graph.run(key, lambda: self._forward_backward_eager(constraints))

    def allreduce(self):
        if self.world > 1:
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM)
