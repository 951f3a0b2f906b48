"""ppsci.autodiff surface (/root/reference/ppsci/autodiff/ad.py) on traced expressions.

In the reference `jacobian(ys, xs)` is one `paddle.grad(ys, xs, create_graph=True)` reverse sweep and
`hessian` two of them (ad.py:56-77, 181-236), cached per (ys, xs) object pair until `clear()`
(ad.py:326-341).  Here ys / xs are `graph.Sym` proxies: the derivative is taken symbolically down to
network-output leaves, which the Taylor-mode HIP kernel then evaluates in its forward pass.  The
call signatures, the cache-by-identity behaviour and the error conditions follow the reference."""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple, Union

import torch

from ..graph import Sym, diff


def _tensor_grad(ys: torch.Tensor, xs: torch.Tensor, i: int, j: Optional[int], create_graph: bool,
                 retain_graph: Optional[bool]) -> torch.Tensor:
    """The reference's eager semantics on real tensors (ad.py:56-77): d ys[:, i] / d xs with the implicit all-ones
    cotangent -- the per-point derivative, because points are independent -- optionally column j of it.  Used by the
    eager fallback (paddlescience_amd/eager.py) for expressions the tracer cannot lower."""
    if not (0 <= i < ys.shape[-1]):
        raise ValueError(f"i({i}) should in range [0, {ys.shape[-1]}).")
    if j is not None and not (0 <= j < xs.shape[-1]):
        raise ValueError(f"j({j}) should in range [0, {xs.shape[-1]}).")
    y = ys[:, i:i + 1] if ys.shape[-1] > 1 else ys
    (g,) = torch.autograd.grad(y, xs, grad_outputs=torch.ones_like(y), create_graph=create_graph,
                               retain_graph=create_graph if retain_graph is None else retain_graph, allow_unused=True)
    if g is None:
        g = torch.zeros_like(xs)
    return g if j is None or xs.shape[-1] == 1 else g[:, j:j + 1]


def _check_x(x):
    if not isinstance(x, Sym) or x.kind not in ("in", "aux"):
        raise TypeError(
            "jacobian/hessian: `xs` must be input variables of the data dict (e.g. out['x']); got "
            f"{x!r}.  Numeric tensors carry no autograd graph on the fused HIP path.")


class Jacobians:
    def __init__(self):
        self.Js: Dict[Tuple[int, int], Sym] = {}

    def _one(self, ys: Sym, x: Sym, i: int, j: Optional[int]) -> Sym:
        _check_x(x)
        if not 0 <= i < 1:
            raise ValueError(f"i({i}) should in range [0, 1).")
        if j is not None and not 0 <= j < 1:
            raise ValueError(f"j({j}) should in range [0, 1).")
        key = (id(ys), id(x))
        if key not in self.Js:
            self.Js[key] = diff(ys, x.name)
        return self.Js[key]

    def __call__(self, ys: Sym, xs: Union[Sym, List[Sym]], i: int = 0, j: Optional[int] = None,
                 retain_graph: Optional[bool] = None, create_graph: bool = True):
        if isinstance(ys, torch.Tensor):  # eager fallback: autograd on real tensors
            one = lambda x: self._tensor_one(ys, x, i, j, create_graph, retain_graph)  # noqa: E731
            return one(xs) if not isinstance(xs, (list, tuple)) else [one(x) for x in xs]
        if not isinstance(ys, Sym):
            raise TypeError("jacobian: `ys` must be a traced expression (output of model(...) or an expression of it)")
        if not isinstance(xs, (list, tuple)):
            return self._one(ys, xs, i, j)
        return [self._one(ys, x, i, j) for x in xs]

    def _tensor_one(self, ys, x, i, j, create_graph, retain_graph):
        if not isinstance(x, torch.Tensor):
            raise TypeError("jacobian: `xs` must be tensors when `ys` is a tensor")
        key = (id(ys), id(x), i, j)  # cached per object pair like the reference (ad.py:100-137)
        if key not in self.Js:
            self.Js[key] = (_tensor_grad(ys, x, i, j, create_graph, retain_graph), ys, x)  # keep ys / x alive: ids stay unique
        return self.Js[key][0]

    def _clear(self):
        self.Js = {}


class Hessians:
    def __init__(self, jac: Jacobians):
        self.Hs: Dict[Tuple[int, int, Optional[int]], Sym] = {}
        self._jac = jac

    def __call__(self, ys: Sym, xs: Sym, component: Optional[int] = None, i: int = 0, j: int = 0,
                 grad_y: Optional[Sym] = None, retain_graph: Optional[bool] = None, create_graph: bool = True) -> Sym:
        if isinstance(ys, torch.Tensor):  # eager fallback (ad.py:181-236)
            if ys.shape[-1] == 1 and component is not None:
                raise ValueError(f"component{component} should be set to None when dim_y(1)=1.")
            if ys.shape[-1] > 1 and component is None:
                raise ValueError("component should not be None when dim_y > 1.")
            if grad_y is None:
                grad_y = self._jac(ys, xs, i=component or 0, j=None)
            return self._jac(grad_y, xs, i, j, retain_graph, create_graph)
        if component is not None:  # every traced field is [N, 1]  (ad.py:214-218)
            raise ValueError(f"component{component} should be set to None when dim_y(1)=1.")
        key = (id(ys), id(xs), component)
        if key not in self.Hs:
            if grad_y is None:
                grad_y = self._jac(ys, xs, i=0, j=None)
            self.Hs[key] = grad_y
        return self._jac(self.Hs[key], xs, i, j)

    def _clear(self):
        self.Hs = {}


jacobian = Jacobians()
hessian = Hessians(jacobian)


def clear():
    """Drop the cached Jacobians / Hessians (ad.py:326-341)."""
    jacobian._clear()
    hessian._clear()
