"""ppsci.equation.Poisson (/root/reference/ppsci/equation/pde/poisson.py:40-53)."""
from typing import Optional, Tuple

from .base import PDE


class Poisson(PDE):
    def __init__(self, dim: int, detach_keys: Optional[Tuple[str, ...]] = None):
        super().__init__()
        self.detach_keys = detach_keys
        invars = self.create_symbols("x y z")[:dim]
        p = self.create_function("p", invars)
        self.dim = dim
        poisson = 0
        for invar in invars:
            poisson += p.diff(invar, 2)
        self.add_equation("poisson", poisson)
        self._apply_detach()
