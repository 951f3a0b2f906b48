"""ppsci.equation.Laplace (/root/reference/ppsci/equation/pde/laplace.py:40-55)."""
from typing import Optional, Tuple

from .base import PDE


class Laplace(PDE):
    def __init__(self, dim: int, detach_keys: Optional[Tuple[str, ...]] = None):
        super().__init__()
        self.detach_keys = detach_keys
        invars = self.create_symbols("x y z")[:dim]
        u = self.create_function("u", invars)
        self.dim = dim
        laplace = 0
        for invar in invars:
            laplace += u.diff(invar, 2)
        self.add_equation("laplace", laplace)
        self._apply_detach()
