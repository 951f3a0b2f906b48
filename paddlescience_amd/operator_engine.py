"""Training engine of the operator-learning path (FNO): network forward + hand-written backward on this framework's
kernels (fno_engine.FnoNative), the loss and its adjoint w.r.t. the network output from the loss object's own kernels
(loss.field: LpLoss / H1Loss / MSELoss on fields) -- gradients land in the model's flat buffer, one SUM all-reduce of that
buffer per step and the fused Adam kernel on it: the same contract (`forward_backward`, `allreduce`, `grad`,
`dp_reduce`) as `engine.Engine` for the PINN path.

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

    def forward_backward_native(self, native) -> None:
        """Network forward + backward on the hand-written kernels (fno_engine.FnoNative).  dL/dy comes from the loss
        object itself when it has kernels for fields (`value_and_grad`: LpLoss / H1Loss / MSELoss on the raw network
        output); any other loss / output expression -- arbitrary Python on the network OUTPUT -- is differentiated by
        torch w.r.t. that one tensor (nothing of the network is on an autograd tape)."""
        m = self.model
        if m._input_transform is not None or m._output_transform is not None:
            raise NotImplementedError("training an FNONet with registered input / output transforms")
        xs = [self.inp[k] for k in m.input_keys]
        x = xs[0] if len(xs) == 1 else torch.cat(xs, dim=1)
        key = m.output_keys[0]
        y = native.forward(x)
        raw = (not self.output_expr or (list(self.output_expr) == [key] and getattr(self.output_expr[key], "is_identity", False))
               or self._expr_is_identity(key, y))
        if raw and hasattr(self.loss_fn, "value_and_grad") and list(self.lab) == [key] and not self.w:
            losses, gy = self.loss_fn.value_and_grad(y, self.lab[key], key)
            self._last = {k: v.detach() for k, v in losses.items()}
            native.backward(gy)
            return
        y = y.detach().requires_grad_(True)
        data = {**self.inp, key: y}
        vals = {k: f(data) for k, f in self.output_expr.items()}
        losses = self.loss_fn(vals, self.lab, self.w)
        self._last = {k: v.detach() for k, v in losses.items()}
        total = None
        for v in losses.values():  # mtl.Sum: left fold in insertion order
            total = v if total is None else total + v
        (gy,) = torch.autograd.grad(total, y)
        native.backward(gy)

    def _expr_is_identity(self, key: str, y: torch.Tensor) -> bool:
        """Is output_expr == {key: lambda out: out[key]} (the reference examples' form)?  Probed once with a marker."""
        hit = getattr(self, "_identity", None)
        if hit is None:
            hit = False
            if list(self.output_expr) == [key]:
                try:
                    hit = self.output_expr[key]({**self.inp, key: y}) is y
                except Exception:  # noqa: BLE001 -- an expression that needs more than the output is not the identity
                    hit = False
            self._identity = hit
        return hit

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
        # forward and backward on this framework's own kernels, no autograd graph of the network (fno_engine.py)
        from . import fno_engine, uno_engine

        why = fno_engine.supports(model)
        if why is not None and uno_engine.supports(model) is not None:
            raise NotImplementedError(f"operator engine: {why}")
        self.native = model.native()  # fno_engine.FnoNative / uno_engine.UnoNative

    def _forward_backward_eager(self, constraints: List[OperatorConstraint]):
        if len(constraints) != 1:
            # (the hand-written backward WRITES the parameter gradients; several constraints on one operator model would
            # need an accumulating variant -- the reference's FNO examples train one supervised constraint)
            raise NotImplementedError("the operator engine trains one constraint per model")
        constraints[0].forward_backward_native(self.native)

    def forward_backward(self, constraints: List[OperatorConstraint]):
        # An FNO step is ~200 kernels of ~10 us (FFTs, the spectral contraction, 1x1 convolutions, norms, their
        # backward): launch-bound from Python/autograd, so forward + loss + backward is captured once per batch shape
        # into a HIP graph and replayed (engine.StepGraph; PPSCI_HIP_GRAPH=0 or a failed capture -> eager).
        # (the executor's generation: a captured step must not outlive the activation buffers it was captured on)
        key = tuple((id(c), c.version) for c in constraints) + (self.native.generation,)
        self._step_graph.run(key, lambda: self._forward_backward_eager(constraints))

    def forward_backward_deferred(self, constraints: List[OperatorConstraint]):
        """forward_backward with the sums over the weight-gradient partials left to the caller (one launch together with its
        Adam update, hp.reduce_rows_multi_adam): returns them as (source, destination, rows, cols) pointers."""
        self.native.defer_wgrad_sums = True
        try:
            key = tuple((id(c), c.version) for c in constraints) + (self.native.generation, "deferred")
            self._step_graph.run(key, lambda: self._forward_backward_eager(constraints))
        finally:
            self.native.defer_wgrad_sums = False
        # (a replayed graph does not run Python: the segment list of the capturing / eager pass stays valid -- same buffers)
        segs = list(self.native._wsegs) if self.native._wsegs else self._deferred_segs
        self._deferred_segs = segs
        return segs

    _deferred_segs = None

    def flush_deferred(self, segs) -> None:
        self.native._wsegs = list(segs)
        self.native._flush_wgrads()

    def allreduce(self):
        if self.world > 1:
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM)
