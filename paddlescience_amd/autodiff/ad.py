"""ppsci.autodiff surface (/root/reference/ppsci/autodiff/ad.py) on traced expressions.

In the reference `jacobian(ys, xs)` is one `paddle.grad(ys, xs, create_graph=True)` reverse sweep and
`hessian` two of them (ad.py:56-77, 181-236), cached per (ys, xs) object pair until `clear()`
(ad.py:326-341).  Here ys / xs are `graph.Sym` proxies: the derivative is taken symbolically down to
network-output leaves, which the Taylor-mode HIP kernel then evaluates in its forward pass.  The
call signatures, the cache-by-identity behaviour and the error conditions follow the reference."""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple, Union


from ..graph import Sym, diff


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
        if not isinstance(ys, Sym):
            raise TypeError("jacobian: `ys` must be a traced expression (output of model(...) or an expression of it); numeric "
                            "tensors carry no derivative graph here -- derivatives are streams of the Taylor-mode kernels")
        if not isinstance(xs, (list, tuple)):
            return self._one(ys, xs, i, j)
        return [self._one(ys, x, i, j) for x in xs]

    def _clear(self):
        self.Js = {}


class Hessians:
    def __init__(self, jac: Jacobians):
        self.Hs: Dict[Tuple[int, int, Optional[int]], Sym] = {}
        self._jac = jac

    def __call__(self, ys: Sym, xs: Sym, component: Optional[int] = None, i: int = 0, j: int = 0,
                 grad_y: Optional[Sym] = None, retain_graph: Optional[bool] = None, create_graph: bool = True) -> Sym:
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
