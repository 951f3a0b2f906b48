"""Elementwise math for user closures (output / input transforms, `output_expr` lambdas).  The reference's scripts
call `paddle.sin(x)` etc. on tensors; here the same closure is traced once on proxy values (graph.Sym) and also run
on torch tensors (eager `model(dict)`), so these functions dispatch on the argument: a traced value becomes an op of
the epilogue program, a tensor / array / number is computed directly.

    import ppsci.functional as F
    model.register_input_transform(lambda d: {"sin(x)": F.sin(b * d["x"] + c), "y": d["y"]})
"""
from __future__ import annotations

import numpy as np
import torch

from .graph import Sym, _lift, apply

_TORCH = {"sin": torch.sin, "cos": torch.cos, "tanh": torch.tanh, "exp": torch.exp, "log": torch.log,
          "sqrt": torch.sqrt, "abs": torch.abs, "sinh": torch.sinh, "cosh": torch.cosh, "tan": torch.tan,
          "asin": torch.asin, "acos": torch.acos, "atan": torch.atan, "asinh": torch.asinh, "acosh": torch.acosh,
          "atanh": torch.atanh, "erf": torch.erf, "sign": torch.sign, "floor": torch.floor, "ceil": torch.ceil}


_NUMPY = {"abs": "abs", "asin": "arcsin", "acos": "arccos", "atan": "arctan", "asinh": "arcsinh", "acosh": "arccosh",
          "atanh": "arctanh"}


def _unary(name):
    def f(x):
        if isinstance(x, Sym):
            return apply(name, x)
        if isinstance(x, torch.Tensor):
            return _TORCH[name](x)
        if name == "erf":
            return torch.erf(torch.as_tensor(x)).numpy()
        return getattr(np, _NUMPY.get(name, name))(x)

    f.__name__ = name
    return f


for _n in _TORCH:
    globals()[_n] = _unary(_n)


def _binary(name, tfn, nfn):
    def f(a, b):
        if isinstance(a, Sym) or isinstance(b, Sym):
            return apply(name, _lift(a), _lift(b))
        if isinstance(a, torch.Tensor) or isinstance(b, torch.Tensor):
            return tfn(torch.as_tensor(a), torch.as_tensor(b))
        return nfn(a, b)

    f.__name__ = name
    return f


maximum = _binary("max", torch.maximum, np.maximum)
minimum = _binary("min", torch.minimum, np.minimum)
pow = _binary("pow", torch.pow, np.power)  # noqa: A001
atan2 = _binary("atan2", torch.atan2, np.arctan2)


def where(cond, x, y):
    """paddle.where(cond, x, y) per point.  Traced: `cond` is the 0 / 1 indicator a comparison of traced values yields
    (graph.Sym._compare; with or without a fixed batch behind the trace, and for `symA == symB`), and the result is  y + cond (x - y)  in the residual program -- both branches are evaluated at every
    point (a branch that is not finite where it is NOT selected would poison the result: the reference's where does not have
    that restriction), the adjoint reaches x with weight cond and y with 1 - cond, the condition itself has derivative zero."""
    ind = getattr(cond, "indicator", None)
    if callable(ind):  # a comparison made during a trace (of a fixed batch's values, or of two traced values): its traced form
        cond = ind()
    if isinstance(cond, Sym) or isinstance(x, Sym) or isinstance(y, Sym):
        if isinstance(cond, (bool, np.bool_)):
            return _lift(x) if cond else _lift(y)
        if not isinstance(cond, Sym):
            raise TypeError("where(): the condition is an array of values of ONE batch; a traced expression needs a traced condition "
                            "(compare traced values: d['bc'] == 1)")
        x, y = _lift(x), _lift(y)
        return y + cond * (x - y)
    if any(isinstance(v, torch.Tensor) for v in (cond, x, y)):
        return torch.where(torch.as_tensor(cond, dtype=torch.bool), torch.as_tensor(x), torch.as_tensor(y))
    return np.where(cond, x, y)


def mean(x):
    """paddle.mean(x) over the whole batch: one scalar, broadcast where it is used (traced: graph.Sym.mean)."""
    return x.mean() if isinstance(x, (Sym, torch.Tensor)) else np.mean(x)


def sum(x):  # noqa: A001
    return x.sum() if isinstance(x, (Sym, torch.Tensor)) else np.sum(x)


__all__ = sorted(list(_TORCH) + ["maximum", "minimum", "pow", "atan2", "where", "mean", "sum"])
