"""ppsci.equation.Vibration (/root/reference/ppsci/equation/pde/viv.py:25-64): vortex-induced vibration,
rho * eta_tt + exp(k1) * eta_t + exp(k2) * eta = f with LEARNABLE k1, k2 (an inverse problem: the two exponents are
trained together with the network)."""
from __future__ import annotations

import sympy as sp

from .base import PDE, EqParam


class Vibration(PDE):
    def __init__(self, rho: float, k1: float, k2: float):
        super().__init__()
        self.rho = rho
        self.k1 = EqParam(k1)
        self.k2 = EqParam(k2)
        self.learnable_parameters.append(self.k1)
        self.learnable_parameters.append(self.k2)
        t_f = self.create_symbols("t_f")
        eta = self.create_function("eta", (t_f,))
        k1 = self.create_symbols(self.k1.name)
        k2 = self.create_symbols(self.k2.name)
        f = self.rho * eta.diff(t_f, 2) + sp.exp(k1) * eta.diff(t_f) + sp.exp(k2) * eta
        self.add_equation("f", f)
        self._apply_detach()
