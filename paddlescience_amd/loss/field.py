"""Losses on [B, C, H, W] fields (the operator-learning path) on this framework's kernels: value AND the adjoint w.r.t. the
network output, csrc/field_loss.hip -- per-row sums S_diff / S_y (values, optionally + first differences), the rows'
terms and their total, and the adjoint field, three small launches.  LpLoss / H1Loss (loss/lp_h1.py) and MSELoss are
configurations of it."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import torch

from .. import _lib as L
from ..hotpath import _p, _require_device, _stream_ptr

REL, ABS, SQ, REL1, ABS1 = 0, 1, 2, 3, 4
VALUES, H1, VALUES_P1 = 0, 1, 2  # `order`: what a row sums


class FieldLossPlan:
    """term(row) over `rows` = B x C fields of H x W samples; loss = coef * sum_rows term(row)."""

    def __init__(self, order: int, mode: int, spacing: Tuple[float, float] = (1.0, 1.0), fix: Tuple[bool, bool] = (False, False),
                 abs_const: float = 1.0):
        self.order, self.mode, self.abs_const = int(order), int(mode), float(abs_const)
        self.ih = (1.0 / spacing[0], 1.0 / spacing[1])
        self.fix = (1 if fix[0] else 0, 1 if fix[1] else 0)
        self._buf = {}

    def _buffers(self, x: torch.Tensor):
        rows = x.shape[0] * x.shape[1]
        key = (rows, x.device)
        if key not in self._buf:
            f = dict(dtype=torch.float32, device=x.device)
            self._buf[key] = (torch.empty(2 * rows, **f), torch.empty(1, **f), torch.empty(rows, **f))
        return self._buf[key]

    def value(self, x: torch.Tensor, y: torch.Tensor, coef: float, want_adjoint: bool = False):
        """(loss [1] tensor, rowcoef or None).  x, y: contiguous fp32 [B, C, H, W] on the device."""
        if x.dim() != 4 or x.shape != y.shape:
            raise ValueError(f"field losses take two [B, C, H, W] tensors, got {tuple(x.shape)} and {tuple(y.shape)}")
        _require_device(x)
        x, y = x.contiguous(), y.to(dtype=torch.float32).contiguous()
        B, Cc, H, W = x.shape
        sums, loss, rowcoef = self._buffers(x)
        st = _stream_ptr(x)
        lib = L.lib()
        L.check(lib.ppsci_field_loss_sums(B * Cc, H, W, self.order, self.ih[0], self.ih[1], self.fix[0], self.fix[1], _p(x), _p(y),
                                          _p(sums), st))
        L.check(lib.ppsci_field_loss_finish(B * Cc, self.mode, self.abs_const, float(coef), _p(sums), _p(loss),
                                            _p(rowcoef) if want_adjoint else None, st))
        return loss, (rowcoef if want_adjoint else None)

    def value_and_grad(self, x: torch.Tensor, y: torch.Tensor, coef: float):
        x, y = x.contiguous(), y.to(dtype=torch.float32).contiguous()
        loss, rowcoef = self.value(x, y, coef, want_adjoint=True)
        B, Cc, H, W = x.shape
        gx = torch.empty_like(x)
        L.check(L.lib().ppsci_field_loss_adjoint(B * Cc, H, W, self.order, self.ih[0], self.ih[1], self.fix[0], self.fix[1], _p(x),
                                                 _p(y), _p(rowcoef), _p(gx), _stream_ptr(x)))
        return loss, gx


class _FieldLossFn(torch.autograd.Function):
    """The scalar loss as a differentiable function of the field `x` (value and adjoint by the kernels): what lets a loss
    object be used inside arbitrary Python on the network output -- weighted sums of several losses, a non-identity output
    expression, a FunctionalLoss -- when the operator engine differentiates that Python with torch (operator_engine.py)."""

    @staticmethod
    def forward(ctx, plan, x, y, coef):
        loss, gx = plan.value_and_grad(x.detach(), y.detach(), coef)
        ctx.save_for_backward(gx)
        return loss.reshape(()).clone()

    @staticmethod
    def backward(ctx, g):
        (gx,) = ctx.saved_tensors
        return None, g * gx, None, None


def scalar_loss(plan: FieldLossPlan, x: torch.Tensor, y: torch.Tensor, coef: float) -> torch.Tensor:
    """plan's loss of (x, y) as a fresh 0-d tensor (never a view of the plan's cached buffer: a caller may keep it across
    batches); differentiable w.r.t. x when x requires a gradient."""
    if x.requires_grad:
        return _FieldLossFn.apply(plan, x, y, float(coef))
    return plan.value(x, y, coef)[0].reshape(()).clone()


def reduce_coef(shape: Sequence[int], reduce_dims: Optional[Sequence[int]], reductions: Sequence[str]) -> Optional[float]:
    """The reference reduces the [B, C] matrix of row terms over `reduce_dims` with sum / mean and squeezes the result;
    when that leaves ONE number, it is coef * (sum of all row terms): the coefficient, else None (not a scalar loss)."""
    B, Cc = int(shape[0]), int(shape[1])
    if reduce_dims is None:
        return 1.0 if B * Cc == 1 else None
    coef, left = 1.0, [B, Cc]
    for d, how in zip(reduce_dims, reductions):
        d = d % 2
        if how == "mean":
            coef /= left[d]
        left[d] = 1
    return coef if left[0] * left[1] == 1 else None
