"""Eager fallback: the reference's own execution model -- model forward, `jacobian` / `hessian` and the operator tree run op
by op on real device tensors with a dynamic autograd graph (/root/reference/ppsci/utils/expression.py:89-126,
ppsci/autodiff/ad.py:56-77, ppsci/solver/train.py:158) -- for constraints whose expressions the tracer cannot lower to the
fused kernels: data-dependent Python control flow, row windows wider than one row (one-row slices such as `d["u"][0:1]` of
examples/euler_beam/euler_beam.py:49-54 ARE lowered, compile.py), derivative sets beyond the instantiated stream sets,
expressions that call tensor methods the proxy does not have.

This path is torch library kernels + torch.autograd on the GPU -- correct, general and slow; the Solver takes it per
constraint, only after the fused lowering of that constraint raised, and says so in the log.  The other constraints of the
same run stay on the fused kernels; the gradients meet in the flat gradient buffer before the all-reduce.

Not supported here (raise): factored / tied layers (weight_norm, random_weight, fourier -- their gradient lives in the
kernel layout on the fused path), learnable activations and learnable equation parameters, SPINN / FNO models."""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import sympy as sp
import torch

from . import _lib as L
from . import autodiff


def _act(name: str, y: torch.Tensor) -> torch.Tensor:
    F = torch.nn.functional
    if name == "tanh":
        return torch.tanh(y)
    if name == "silu":
        return y * torch.sigmoid(y)
    if name == "sin":
        return torch.sin(y)
    if name == "siren":
        return torch.sin(L.SIREN_W0 * y)
    if name == "cos":
        return torch.cos(y)
    if name == "sigmoid":
        return torch.sigmoid(y)
    if name == "gelu":
        return F.gelu(y)
    if name == "relu":
        return F.relu(y)
    if name == "leaky_relu":
        return F.leaky_relu(y, 0.01)
    if name == "elu":
        return F.elu(y, 1.0)
    if name == "selu":
        return F.selu(y)
    if name == "identity":
        return y
    raise NotImplementedError(f"activation {name!r} on the eager fallback path")


def supports(model) -> Optional[str]:
    from .arch.mlp import MLP
    from .arch.model_list import ModelList

    members = model.model_list if isinstance(model, ModelList) else [model]
    for m in members:
        if not isinstance(m, MLP) or type(m).__name__ == "LayerwiseMLP":  # (the layer-by-layer MLP is an MLP to isinstance)
            return f"{type(m).__name__} models"
        if getattr(m, "reparam", False):
            return "factored / tied layers or learnable activations"
    return None


def mlp_forward(model, x: Dict[str, torch.Tensor], flat: torch.Tensor) -> Dict[str, torch.Tensor]:
    """ppsci.arch.MLP.forward (mlp.py:281-315) in torch ops; the parameters are slices of `flat` (an autograd leaf that
    shares storage with model.flat_params), so `flat.grad` is the gradient in the trainable layout."""
    xin = model._input_transform(dict(x)) if model._input_transform is not None else x
    cols = []
    for k in model.input_keys:
        v = xin[k]
        if model.periods and k in model.period_emb.freqs_dict:  # PeriodEmbedding (mlp.py:108-114)
            w = model.period_emb.freqs_dict[k]
            cols += [torch.cos(w * v), torch.sin(w * v)]
        else:
            cols.append(v)
    y = torch.cat(cols, dim=-1)
    off = 0
    views = {}
    for name, shp in model._shapes:
        n = int(np.prod(shp))
        views[name] = flat[off:off + n].view(tuple(shp))
        off += n
    nl = model._n_hidden_linears
    for i in range(nl):
        y = y @ views[f"linears.{i}.weight"] + views[f"linears.{i}.bias"]
        if model.skip_connection and i % 2 == 0:  # mlp.py:286-291
            if i >= 2:
                y = y + y  # `skip = y; y = y + skip` -- the reference adds the tensor to itself
        y = _act(model.activation, y)
    y = y @ views["last_fc.weight"] + views["last_fc.bias"]
    out = {k: y[:, i:i + 1] for i, k in enumerate(model.output_keys)}
    if model._output_transform is not None:
        out = model._output_transform(x, out)
    return out


def model_forward(model, x: Dict[str, torch.Tensor], flat: torch.Tensor) -> Dict[str, torch.Tensor]:
    members = getattr(model, "model_list", None)
    if members is None:
        return mlp_forward(model, x, flat)
    out: Dict[str, torch.Tensor] = {}
    for m in members:
        n = m.flat_params.numel()
        out.update(mlp_forward(m, x, flat[m._train_offset:m._train_offset + n]))
    if model._output_transform is not None:
        out = model._output_transform(x, out)
    return out


class _EagerModel:
    """What the expressions see as `model` while the eager constraint runs: calling it evaluates the eager forward."""

    def __init__(self, model, flat):
        self._m, self._flat = model, flat
        self.input_keys, self.output_keys = model.input_keys, model.output_keys

    def __call__(self, x):
        return model_forward(self._m, x, self._flat)


class EagerConstraint:
    """One constraint on the eager path; same surface as compile.CompiledConstraint where the Solver touches it."""

    is_eager = True

    def __init__(self, name: str, model, exprs: Dict[str, Callable], input_keys: Sequence[str], label_keys: Sequence[str],
                 weight_keys: Sequence[str], loss, batch_size: int, n_global: int, device, reason: str = ""):
        why = supports(model)
        if why is not None:
            raise NotImplementedError(f"constraint {name}: not lowerable to the fused kernels ({reason}) and the eager "
                                      f"fallback does not cover {why}")
        self.name, self.model, self.loss, self.device = name, model, loss, device
        self.exprs = dict(exprs)
        self.input_keys, self.label_keys, self.weight_keys = list(input_keys), list(label_keys), list(weight_keys)
        self.batch_size, self.n_global = batch_size, n_global
        self.inp = self.lab = self.w = None
        self._last: Dict[str, float] = {}
        self._graph = None  # None: not captured yet; False: capture failed; (key, CUDAGraph)
        self._fns: Dict[str, Callable] = {}

    def _t(self, a):
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=torch.float32)
        if isinstance(a, (int, float)):
            return torch.tensor(float(a), dtype=torch.float32, device=self.device)
        return torch.as_tensor(np.asarray(a, dtype=np.float32)).to(self.device)

    def bind(self, input, label, weight=None):
        """New batch: written IN PLACE into the tensors of the previous one when names and shapes agree, so that a captured
        graph of the step (below) keeps reading the right memory."""
        def put(old, new):
            new = {k: self._t(v) for k, v in (new or {}).items()}
            if old is not None and old.keys() == new.keys() and all(old[k].shape == new[k].shape for k in new):
                for k, v in new.items():
                    old[k].copy_(v)
                return old
            self._graph = None  # shapes changed: the captured step is stale
            return new

        self.inp, self.lab, self.w = put(self.inp, input), put(self.lab, label), put(self.w, weight)

    def _values(self, flat: torch.Tensor, need_grad: bool):
        from .utils.symbolic import lambdify

        data = {k: (v.detach().requires_grad_(True) if need_grad or True else v) for k, v in self.inp.items()}
        view = _EagerModel(self.model, flat)
        out = view({k: data[k] for k in data})
        data.update(out)
        vals = {}
        for k, ex in self.exprs.items():
            if isinstance(ex, sp.Basic):
                if k not in self._fns:
                    self._fns[k] = lambdify(ex, self.model)
                ex = self._fns[k]
                # LayerNode evaluation inside lambdify calls the model: it must be the eager one
                ex.models = (view,)
            vals[k] = ex(data)
        autodiff.clear()
        for k in self.label_keys:  # a label on a raw network output
            if k not in vals and k in out:
                vals[k] = out[k]
        if "area" in self.inp:
            vals["area"] = self.inp["area"]
        return vals

    def _step(self, grad: torch.Tensor, dp_scale: float) -> None:
        flat = self.model.flat_params.detach().requires_grad_(True)
        vals = self._values(flat, True)
        losses = self.loss(vals, self.lab, self.w if self.w else None)
        total = None
        for v in losses.values():  # mtl.Sum: left fold in insertion order
            total = v if total is None else total + v
        if dp_scale != 1.0:
            total = total * dp_scale
        (g,) = torch.autograd.grad(total, flat, allow_unused=True)
        if g is not None:
            grad.add_(g)
        self._loss_t = {k: v.detach() for k, v in losses.items()}  # fetched by losses(), i.e. only when they are logged

    def forward_backward(self, grad: torch.Tensor, dp_scale: float = 1.0) -> None:
        """loss terms -> `losses()`; d(sum of terms)/d(trainable parameters) is ADDED into `grad` (flat, trainable layout).
        `dp_scale`: batch_size / n_global for "mean" losses so that the SUM all-reduce over ranks yields the global mean.

        The op-by-op path is a few hundred tiny torch kernels (4 boundary points with fourth derivatives: 4 ms of launches); with
        static shapes the whole sequence -- forward, higher-order autograd, the gradient add -- is captured once into a HIP graph
        and replayed when PPSCI_EAGER_GRAPH=1 asks for it (default, or a failed capture: launched op by op)."""
        import os

        # A replayed graph freezes whatever the user's expressions did on the HOST at capture time (Python branches,
        # counters, closures over mutable state) and the device pointers of every tensor it touched: the capture is
        # OPT-IN (PPSCI_EAGER_GRAPH=1; the reference re-runs the Python every step) and keyed by those pointers.
        key = (grad.data_ptr(), float(dp_scale), self.model.flat_params.data_ptr(),
               tuple(int(t.data_ptr()) for d in (self.inp, self.lab, self.w) if isinstance(d, dict)
                     for t in d.values() if isinstance(t, torch.Tensor)))
        if not grad.is_cuda or os.environ.get("PPSCI_EAGER_GRAPH", "0") != "1" or self._graph is False:
            return self._step(grad, dp_scale)
        if self._graph is not None and self._graph[0] == key:
            self._graph[1].replay()
            return
        self._calls = getattr(self, "_calls", 0) + 1
        self._step(grad, dp_scale)  # the first calls run op by op (allocator warm-up, lazy initialisation inside torch)
        if self._calls < 3:
            return
        try:
            torch.cuda.synchronize()
            saved, eager_losses = grad.clone(), self._loss_t
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="relaxed"):
                self._step(grad, dp_scale)
            grad.copy_(saved)  # (a capture does not execute; this only guards against a partial one)
            for k, v in self._loss_t.items():  # the graph's loss tensors: this iteration's values come from the eager run above
                v.copy_(eager_losses[k])
            self._graph = (key, g)
        except Exception as e:  # noqa: BLE001 -- capture is an optimisation, never a requirement
            from .utils import logger

            logger.warning(f"constraint {self.name}: HIP-graph capture of the eager step failed ({type(e).__name__}: {e}); it "
                           "stays op by op")
            self._graph = False
            torch.cuda.synchronize()

    def values(self) -> Dict[str, torch.Tensor]:
        with torch.enable_grad():
            vals = self._values(self.model.flat_params.detach(), False)
        return {k: v.detach() for k, v in vals.items() if isinstance(v, torch.Tensor)}

    def losses(self) -> Dict[str, float]:
        self._last = {k: float(v) for k, v in getattr(self, "_loss_t", {}).items()}
        return dict(self._last)
