"""ppsci.equation.NLSMB (/root/reference/ppsci/equation/pde/nls_m_b.py:24-101): the nonlinear Schroedinger-Maxwell-Bloch system
    E_x = i a1 E_tt - i a2 |E|^2 E + 2 p,     p_t = 2 i w0 p + 2 E eta,     eta_t = -(E p* + E* p)
for E = Eu + i Ev, p = pu + i pv, split into its real and imaginary parts (five residuals of five network outputs)."""
from typing import Optional, Tuple, Union

from .base import PDE


class NLSMB(PDE):
    def __init__(self, alpha_1: Union[float, str], alpha_2: Union[float, str], omega_0: Union[float, str], time: bool,
                 detach_keys: Optional[Tuple[str, ...]] = None):
        super().__init__()
        self.detach_keys = detach_keys
        self.time = time
        self.alpha_1, self.alpha_2, self.omega_0 = alpha_1, alpha_2, omega_0
        t, x = self.create_symbols("t x")
        invars = ((t,) if time else ()) + (x,)
        Eu, Ev, pu, pv, eta = (self.create_function(name, invars) for name in ("Eu", "Ev", "pu", "pv", "eta"))
        intensity = Eu ** 2 + Ev ** 2

        def schrodinger(field, polarisation, other_x, sign):
            """real / imaginary part of  i a1 E_tt - i a2 |E|^2 E + 2 p - E_x"""
            return alpha_1 * field.diff(t).diff(t) - alpha_2 * field * intensity + sign * 2 * polarisation - sign * other_x

        self.add_equation("Schrodinger_1", schrodinger(Eu, pv, Ev.diff(x), 1))
        self.add_equation("Schrodinger_2", schrodinger(Ev, pu, Eu.diff(x), -1))
        self.add_equation("Maxwell_1", 2 * Ev * eta - pv.diff(t) + 2 * pu * omega_0)
        self.add_equation("Maxwell_2", -2 * Eu * eta + pu.diff(t) + 2 * pv * omega_0)
        self.add_equation("Bloch", 2 * pv * Ev + 2 * pu * Eu + eta.diff(t))
        self._apply_detach()
