"""ppsci.equation.AllenCahn (/root/reference/ppsci/equation/pde/allen_cahn.py:25-64): a Python
closure on the data dict, u_t - eps^2 u_xx + 5u^3 - 5u with u*u*u spelled out (allen_cahn.py:55,62)."""
from typing import Optional, Tuple

from ...autodiff import jacobian
from .base import PDE


class AllenCahn(PDE):
    def __init__(self, eps: float, detach_keys: Optional[Tuple[str, ...]] = None):
        super().__init__()
        self.detach_keys = detach_keys
        self.eps = eps

        def allen_cahn(out):
            t, x = out["t"], out["x"]
            u = out["u"]
            u__t, u__x = jacobian(u, [t, x])
            u__x__x = jacobian(u__x, x)
            return u__t - (self.eps**2) * u__x__x + 5 * u * u * u - 5 * u

        self.add_equation("allen_cahn", allen_cahn)
