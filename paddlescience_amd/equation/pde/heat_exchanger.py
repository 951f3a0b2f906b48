"""ppsci.equation.HeatExchanger (/root/reference/ppsci/equation/pde/heat_exchanger.py:24-94): energy balances of the hot fluid,
the cold fluid (counter-flow) and the wall between them; the mass flow rates qm_h / qm_c are inputs of the fluid networks."""
from typing import Union

from .base import PDE


class HeatExchanger(PDE):
    def __init__(self, alpha_h: Union[float, str], alpha_c: Union[float, str], v_h: Union[float, str], v_c: Union[float, str],
                 w_h: Union[float, str], w_c: Union[float, str]):
        super().__init__()
        x, t, qm_h, qm_c = self.create_symbols("x t qm_h qm_c")
        T_h = self.create_function("T_h", (x, t, qm_h))
        T_c = self.create_function("T_c", (x, t, qm_c))
        T_w = self.create_function("T_w", (x, t))

        # beta = alpha v / qm with the positive speed on both sides; the cold side flows against x
        self.add_equation("heat_boundary", T_h.diff(t) + v_h * T_h.diff(x) - (alpha_h * v_h) / qm_h * (T_w - T_h))
        self.add_equation("cold_boundary", T_c.diff(t) - v_c * T_c.diff(x) - (alpha_c * v_c) / qm_c * (T_w - T_c))
        self.add_equation("wall", T_w.diff(t) - w_h * (T_h - T_w) - w_c * (T_c - T_w))
        self._apply_detach()
