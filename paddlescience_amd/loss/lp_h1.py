"""LpLoss / H1Loss of the neural-operator examples (/root/reference/examples/neuraloperator/metric.py:69-412) as
configurations of the field-loss kernels (loss/field.py, csrc/field_loss.hip).

Per sample and channel (a "row"), over the last d = 2 dimensions:
    LpLoss (p = 1, 2)  rel: |x - y|_p / |y|_p        abs: (h_x h_y)^(1/p) |x - y|_p
    H1Loss           the same with the squared norms of the central differences added (value + gradient: an H^1 norm);
                     differences wrap around periodically, or are one-sided on the first / last row / column under
                     fix_x_bnd / fix_y_bnd (metric.py:36-55)
the rows are then reduced over `reduce_dims` with sum / mean.  `LpLoss` / `H1Loss` are the validation metrics (rel error
divided by the batch size, keys "l2" / "h1"); `LpLoss_train` / `H1Loss_train` the training losses (key "y").  The
training variants also hand the operator engine the adjoint w.r.t. the network output (`value_and_grad`), so that no
autograd tape is involved in an FNO step."""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Union

import torch

from . import field


class _RowLoss:
    order = 0

    def __init__(self, d, L, reduce_dims, reductions, fix=(False, False)):
        if d != 2:  # (the reference's constructor default is d = 1, metric.py:88: its examples pass d = 2)
            raise NotImplementedError(f"{type(self).__name__}(d={d}): the field-loss kernels cover 2-D fields (the FNO path is 2-D); "
                                      "pass d=2")
        self.d = d
        self.L = [float(L)] * d if isinstance(L, (int, float)) else [float(v) for v in L]
        self.reduce_dims = [reduce_dims] if isinstance(reduce_dims, int) else (None if reduce_dims is None else list(reduce_dims))
        if self.reduce_dims is not None:
            rs = [reductions] * len(self.reduce_dims) if isinstance(reductions, str) else list(reductions)
            if any(r not in ("sum", "mean") for r in rs) or len(rs) != len(self.reduce_dims):
                raise AssertionError("reductions must be 'sum' or 'mean', one per reduced dimension")
            self.reductions = rs
        else:
            self.reductions = []
        self.fix = fix
        self._plans: Dict[tuple, field.FieldLossPlan] = {}

    def _spacing(self, x, h):
        if h is None:
            return tuple(self.L[-j] / x.shape[-j] for j in range(self.d, 0, -1))  # uniform grid on a domain of length L
        return tuple([float(h)] * self.d if isinstance(h, (int, float)) else [float(v) for v in h])

    def _plan(self, mode: int, spacing, abs_const: float = 1.0) -> field.FieldLossPlan:
        key = (mode, spacing, abs_const)
        if key not in self._plans:
            self._plans[key] = field.FieldLossPlan(self.order, mode, spacing, self.fix, abs_const)
        return self._plans[key]

    def _coef(self, x) -> float:
        c = field.reduce_coef(x.shape, self.reduce_dims, self.reductions)
        if c is None:
            raise NotImplementedError(f"{type(self).__name__}: reduce_dims={self.reduce_dims} leaves more than one number for "
                                      f"fields of shape {tuple(x.shape)}; the kernels produce the scalar loss")
        return c

    def _abs_const(self, spacing) -> float:
        raise NotImplementedError

    rel_mode, abs_mode = field.REL, field.ABS

    def rel(self, x, y, h=None):
        return field.scalar_loss(self._plan(self.rel_mode, self._spacing(x, h)), x, y, self._coef(x))

    def abs(self, x, y, h=None):
        sp = self._spacing(x, h)
        return field.scalar_loss(self._plan(self.abs_mode, sp, self._abs_const(sp)), x, y, self._coef(x))

    def rel_and_grad(self, x, y, h=None, scale: float = 1.0):
        loss, gx = self._plan(self.rel_mode, self._spacing(x, h)).value_and_grad(x, y, self._coef(x) * scale)
        return loss.reshape(()), gx


class LpLoss(_RowLoss):
    """metric.py:69-176 (p = 1 or 2)."""

    order = 0

    def __init__(self, d: int = 1, p: int = 2, L: Union[float, Sequence[float]] = 2 * math.pi, reduce_dims=0, reductions="sum"):
        if p not in (1, 2):
            raise NotImplementedError(f"LpLoss(p={p}): the kernels sum magnitudes (p = 1) or squares (p = 2)")
        self.p = p
        super().__init__(d, L, reduce_dims, reductions)
        if p == 1:
            self.order, self.rel_mode, self.abs_mode = field.VALUES_P1, field.REL1, field.ABS1

    def _abs_const(self, spacing) -> float:
        return math.prod(spacing)  # (prod h)^(1/p) |e|_p: sqrt(prod h * S_diff) for p = 2, prod h * S_diff for p = 1

    def __call__(self, output_dict: Dict[str, torch.Tensor], label_dict: Dict[str, torch.Tensor], weight_dict=None):
        x, y = output_dict["y"], label_dict["y"]
        return {"l2": self.rel(x, y) / x.shape[0]}


class LpLoss_train(LpLoss):  # noqa: N801 -- the reference's name (metric.py:179-193)
    def __call__(self, output_dict, label_dict, weight_dict=None):
        return {"y": self.rel(output_dict["y"], label_dict["y"])}

    def value_and_grad(self, y_net, label, key):
        loss, g = self.rel_and_grad(y_net, label)
        return {key: loss}, g


class H1Loss(_RowLoss):
    """metric.py:196-383 (d = 2)."""

    order = 1

    def __init__(self, d: int = 1, L: Union[float, Sequence[float]] = 2 * math.pi, reduce_dims=0, reductions="sum",
                 fix_x_bnd: bool = False, fix_y_bnd: bool = False, fix_z_bnd: bool = False):
        assert d > 0 and d < 4, "Currently only implemented for 1, 2, and 3-D."
        super().__init__(d, L, reduce_dims, reductions, (bool(fix_x_bnd), bool(fix_y_bnd)))

    def _abs_const(self, spacing) -> float:
        return math.prod(spacing)  # sqrt(prod h * (|e|^2 + |D e|^2))

    def __call__(self, output_dict, label_dict, weight_dict: Optional[dict] = None, h=None):
        x, y = output_dict["y"], label_dict["y"]
        return {"h1": self.rel(x, y, h=h) / x.shape[0]}


class H1Loss_train(H1Loss):  # noqa: N801 -- the reference's name (metric.py:386-412)
    def __call__(self, output_dict, label_dict, weight_dict: Optional[dict] = None, h=None):
        return {"y": self.rel(output_dict["y"], label_dict["y"], h=h)}

    def value_and_grad(self, y_net, label, key):
        loss, g = self.rel_and_grad(y_net, label)
        return {key: loss}, g
