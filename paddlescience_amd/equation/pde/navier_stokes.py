"""ppsci.equation.NavierStokes (/root/reference/ppsci/equation/pde/navier_stokes.py:27-151):
continuity and momentum residuals as sympy expressions; nu / rho may be numbers or strings parsed by
sympy (extra input variables when they are symbols)."""
from typing import Optional, Tuple, Union

import sympy as sp
from sympy.parsing import sympy_parser as sp_parser

from .base import PDE


class NavierStokes(PDE):
    def __init__(self, nu: Union[float, str], rho: Union[float, str], dim: int, time: bool,
                 detach_keys: Optional[Tuple[str, ...]] = None):
        super().__init__()
        self.detach_keys = detach_keys
        self.dim = dim
        self.time = time
        t, x, y, z = self.create_symbols("t x y z")
        invars = (x, y)
        if time:
            invars = (t,) + invars
        if dim == 3:
            invars += (z,)
        if isinstance(nu, str):
            nu = sp_parser.parse_expr(nu)
            if isinstance(nu, sp.Symbol):
                invars += (nu,)
        if isinstance(rho, str):
            rho = sp_parser.parse_expr(rho)
            if isinstance(rho, sp.Symbol):
                invars += (rho,)
        self.nu, self.rho = nu, rho
        u = self.create_function("u", invars)
        v = self.create_function("v", invars)
        w = self.create_function("w", invars) if dim == 3 else sp.Number(0)
        p = self.create_function("p", invars)

        def momentum(k, xk):
            return (k.diff(t) + u * k.diff(x) + v * k.diff(y) + w * k.diff(z)
                    - ((nu * k.diff(x)).diff(x) + (nu * k.diff(y)).diff(y) + (nu * k.diff(z)).diff(z))
                    + 1 / rho * p.diff(xk))

        self.add_equation("continuity", u.diff(x) + v.diff(y) + w.diff(z))
        self.add_equation("momentum_x", momentum(u, x))
        self.add_equation("momentum_y", momentum(v, y))
        if self.dim == 3:
            self.add_equation("momentum_z", momentum(w, z))
        self._apply_detach()
