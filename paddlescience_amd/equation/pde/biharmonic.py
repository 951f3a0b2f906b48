"""ppsci.equation.Biharmonic (/root/reference/ppsci/equation/pde/biharmonic.py:27-78):
nabla^4 u - q / D = sum_i sum_j u_{ii jj} - q / D   (euler_beam.py: dim 1, u_xxxx + 1; biharmonic2d.py: dim 2).

The fourth derivatives are carried by the Taylor kernels' third / fourth-order streams (taylor_fwd.inc N3, N4); the mixed
term u_xxyy of dim >= 2 by polarisation over the directions x + y and x - y (graph.lower)."""
from typing import Optional, Tuple, Union

import sympy

from .base import PDE


class Biharmonic(PDE):
    def __init__(self, dim: int, q: Union[float, str, sympy.Basic], D: Union[float, str],
                 detach_keys: Optional[Tuple[str, ...]] = None):
        super().__init__()
        self.detach_keys = detach_keys
        invars = self.create_symbols("x y z")[:dim]
        u = self.create_function("u", invars)
        if isinstance(q, str):
            q = self.create_function("q", invars)
        if isinstance(D, str):
            D = self.create_function("D", invars)
        self.dim, self.q, self.D = dim, q, D
        biharmonic = -self.q / self.D
        for invar_i in invars:
            for invar_j in invars:
                biharmonic += u.diff(invar_i, 2).diff(invar_j, 2)
        self.add_equation("biharmonic", biharmonic)
        self._apply_detach()
