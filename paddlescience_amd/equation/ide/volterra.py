"""ppsci.equation.Volterra (/root/reference/ppsci/equation/ide/volterra.py:27-135): a Volterra integral equation of the
second kind,  x(t) - f(t) = int_a^t K(t, s) x(s) ds,  with Gauss-Legendre quadrature.

The batch of the constraint is  [the N collocation points | Q quadrature points of point 1 | ... | of point N]  (the example's
dataset transform builds it with `get_quad_points`, examples/ide/volterra_ide.py:66-94); residual i < N is
    func(out)_i - sum_q  w_q(t_i) K(t_i, s_iq) u(s_iq),
i.e. `lhs[:N] - int_mat @ u` with a constant [N, N + N Q] matrix: a batch-COUPLED residual.  It is traced as graph.couple and
runs as per-point programs around two small matrix-vector launches (engine.FusedConstraint._forward_couplings); the matrix is
built from the VALUES of the (fixed) batch at trace time, as the reference builds it from `float(x[i])` on every call."""
from __future__ import annotations

from typing import Callable

import numpy as np

from ..pde.base import PDE


class Volterra(PDE):
    dtype = "float32"

    def __init__(self, bound: float, num_points: int, quad_deg: int, kernel_func: Callable, func: Callable):
        super().__init__()
        self.bound = bound
        self.num_points = num_points
        self.quad_deg = quad_deg
        self.kernel_func = kernel_func
        self.func = func
        qx, qw = np.polynomial.legendre.leggauss(quad_deg)
        self.quad_x = qx.astype(Volterra.dtype).reshape(-1, 1)  # [Q, 1]
        self.quad_w = qw.astype(Volterra.dtype)                 # [Q]

        def compute_volterra_func(out):
            from ... import graph

            x, u = out["x"], out["u"]
            lhs = self.func(out)
            if isinstance(x, graph.Sym):
                xv = graph.concrete_values(x, "Volterra's quadrature matrix").astype(Volterra.dtype).reshape(-1, 1)
                graph._TRACE.concretized.append(f"volterra_matrix({x!r}) crc {_crc(xv):08x}")
                return graph.couple(lhs, self._get_int_matrix(xv), u)
            # real tensors / arrays (evaluating the equation outside a compiled constraint)
            import torch

            xv = x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)
            mat = self._get_int_matrix(np.asarray(xv, dtype=Volterra.dtype).reshape(-1, 1))
            if isinstance(u, torch.Tensor):
                rhs = torch.as_tensor(mat, dtype=u.dtype, device=u.device) @ u
            else:
                rhs = mat @ np.asarray(u)
            return lhs[:len(rhs)] - rhs

        self.add_equation("volterra", compute_volterra_func)

    def get_quad_points(self, t):
        """volterra.py:79-93: Gauss points of [a, t_i] for every upper bound: [N, 1] -> [N, Q] (arrays or tensors)."""
        import torch

        a = self.bound
        if isinstance(t, torch.Tensor):
            qx = torch.as_tensor(self.quad_x, dtype=t.dtype, device=t.device)
            return ((t - a) / 2) @ qx.T + (t + a) / 2
        t = np.asarray(t, dtype=Volterra.dtype)
        return ((t - a) / 2) @ self.quad_x.T + (t + a) / 2

    def _get_quad_weights(self, t: float) -> np.ndarray:
        """volterra.py:95-107: Gauss weights scaled to [a, t]."""
        return (t - self.bound) / 2 * self.quad_w

    def _get_int_matrix(self, x: np.ndarray) -> np.ndarray:
        """volterra.py:109-135: row i holds w_q(x_i) K(x_i, s_iq) at the columns of point i's quadrature points."""
        n, q = self.num_points, self.quad_deg
        if len(x) != n + n * q:
            raise ValueError(f"Volterra(num_points={n}, quad_deg={q}) expects a batch of {n + n * q} points "
                             f"([points | their quadrature points]), got {len(x)}")
        mat = np.zeros((n, n + n * q), dtype=Volterra.dtype)
        for i in range(n):
            xi = float(np.asarray(x[i]).reshape(-1)[0])
            beg, end = n + q * i, n + q * (i + 1)
            k = np.ravel(self.kernel_func(np.full((q, 1), xi), np.asarray(x[beg:end])))
            mat[i, beg:end] = self._get_quad_weights(xi) * k
        return mat


def _crc(a: np.ndarray) -> int:
    import zlib

    return zlib.crc32(np.ascontiguousarray(a).tobytes())
