"""ppsci.equation.LinearElasticity (/root/reference/ppsci/equation/pde/linear_elasticity.py:25-184): stress-displacement relations
(Hooke), momentum balance and boundary tractions for displacement (u, v, w) and stress (sigma_ij) networks.  Lame parameters from
(E, nu) or given directly; a string names a network output / input column instead of a constant."""
from typing import Optional, Tuple, Union

import sympy as sp

from .base import PDE


class LinearElasticity(PDE):
    def __init__(self, E: Optional[Union[float, str]] = None, nu: Optional[Union[float, str]] = None,
                 lambda_: Optional[Union[float, str]] = None, mu: Optional[Union[float, str]] = None, rho: Union[float, str] = 1,
                 dim: int = 3, time: bool = False, detach_keys: Optional[Tuple[str, ...]] = None):
        super().__init__()
        self.detach_keys = detach_keys
        self.dim, self.time = dim, time
        t, x, y, z = self.create_symbols("t x y z")
        normal = self.create_symbols("normal_x normal_y normal_z")
        invars = ((t,) if time else ()) + (x, y) + ((z,) if dim == 3 else ())
        coords = (x, y, z)

        def field(name, three_d_only=False):
            return sp.Number(0) if (three_d_only and dim != 3) else self.create_function(name, invars)

        def material(value):
            return self.create_function(value, invars) if isinstance(value, str) else value

        disp = (field("u"), field("v"), field("w", True))
        # symmetric stress tensor: sigma[i][j] for i <= j
        sig = {(0, 0): field("sigma_xx"), (1, 1): field("sigma_yy"), (0, 1): field("sigma_xy"), (2, 2): field("sigma_zz", True),
               (0, 2): field("sigma_xz", True), (1, 2): field("sigma_yz", True)}
        sigma = lambda i, j: sig[(min(i, j), max(i, j))]  # noqa: E731
        if lambda_ is None:
            nu, E = material(nu), material(E)
            lambda_ = nu * E / ((1 + nu) * (1 - 2 * nu))
            mu = E / (2 * (1 + nu))
        else:
            lambda_, mu = material(lambda_), material(mu)
        rho = material(rho)
        self.E, self.nu, self.lambda_, self.mu, self.rho = E, nu, lambda_, mu, rho

        divergence = sum(d.diff(c) for d, c in zip(disp, coords))
        names = "xyz"
        ndim = 3 if dim == 3 else 2
        # Hooke: lambda div(u) + 2 mu u_i,i - sigma_ii   and   mu (u_i,j + u_j,i) - sigma_ij
        for i in range(ndim):
            self.add_equation(f"stress_disp_{names[i]}{names[i]}", lambda_ * divergence + 2 * mu * disp[i].diff(coords[i]) - sigma(i, i))
            if i == 1:  # (the reference registers xx, yy, xy first, then the z components)
                self.add_equation("stress_disp_xy", mu * (disp[0].diff(y) + disp[1].diff(x)) - sigma(0, 1))
        if dim == 3:
            self.add_equation("stress_disp_xz", mu * (disp[0].diff(z) + disp[2].diff(x)) - sigma(0, 2))
            self.add_equation("stress_disp_yz", mu * (disp[1].diff(z) + disp[2].diff(y)) - sigma(1, 2))
        # momentum balance: rho u_i,tt - sigma_ij,j
        for i in range(ndim):
            self.add_equation(f"equilibrium_{names[i]}",
                              rho * disp[i].diff(t).diff(t) - sum(sigma(i, j).diff(coords[j]) for j in range(3)))
        # tractions: n_j sigma_ij
        for i in range(ndim):
            self.add_equation(f"traction_{names[i]}", sum(normal[j] * sigma(i, j) for j in range(3)))
        self._apply_detach()
