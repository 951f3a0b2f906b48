"""ppsci.equation.Helmholtz (/root/reference/ppsci/equation/pde/helmholtz.py:44-95): k^2 u + u_xx + u_yy + u_zz
on a separable net.  The reference obtains each u_ii with `hvp_revrev` -- two nested
`paddle.incubate.autograd.jvp` through `model.forward_tensor` with unit tangents (:27-41, :86-88); here the
same quantity is the second-derivative stream of the corresponding branch net (SPINN.second_derivative)."""
from typing import Dict, Optional, Tuple

from .base import PDE


class Helmholtz(PDE):
    def __init__(self, dim: int, k: float, detach_keys: Optional[Tuple[str, ...]] = None):
        super().__init__()
        self.dim, self.k, self.detach_keys = dim, k, detach_keys
        self.model = None  # set by the user script (examples/spinn/helmholtz3d.py:122)

        def helmholtz(data_dict: Dict[str, object]):
            u = data_dict["u"]
            keys = ("x", "y", "z")[: self.dim]
            out = (self.k**2) * u
            for key in keys:
                out = out + self.model.second_derivative(key)
            return out

        self.add_equation("helmholtz", helmholtz)
        self._apply_detach()
