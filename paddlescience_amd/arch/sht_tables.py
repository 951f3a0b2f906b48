"""Quadrature nodes / weights and normalised associated Legendre functions of the spherical-harmonic transform pair SFNO runs on
(/root/reference/ppsci/arch/paddle_harmonics/{quadrature,legendre,sht}.py), as the two tables the kernels of csrc/sht.hip read.

The pair, for a real field x[k][j] on nlat x nlon points (colatitude theta_k from the north pole to the south pole, longitude j):

    analysis   X[l][m] = sum_k A[m][l][k] * ( sum_j x[k][j] e^{-2 pi i j m / nlon} )        RealSHT.forward          sht.py:118-150
    synthesis  y[k][j] = sum_m Re( ( sum_l B[m][l][k] Z[l][m] ) e^{+2 pi i j m / nlon} )    InverseRealSHT.forward   sht.py:216-232

with  A[m][l][k] = (2 pi / nlon) w_k Pbar_l^m(cos theta_k)     (rfft(norm="forward") * 2 pi, then the quadrature, sht.py:124-147)
      B[m][l][k] = c_m Pbar_l^m(cos theta_k)                   (irfft(n=nlon, norm="forward") of mmax <= nlon/2 + 1 columns: the
                                                                Hermitian weights c_0 = 1, c_m = 2, c_{nlon/2} = 1 folded in)

Both are linear and real-in / real-out, so each one's adjoint is the OTHER kernel on its own table: the adjoint of the synthesis is
the analysis kernel reading B, the adjoint of the analysis is the synthesis kernel reading A (derivation in csrc/sht.hip).

Pbar: fully normalised ("ortho": integral of |Y_l^m|^2 over the sphere = 1) with the Condon-Shortley phase, by the standard
three-term recurrence in l started from the sectoral values (legendre.py:48-110).  Quadrature: "equiangular" = Clenshaw-Curtis on
theta_k = k pi / (nlat - 1) (quadrature.py:88-121; the closed-form cosine series here instead of its FFT construction),
"legendre-gauss" = numpy's Gauss-Legendre rule, "lobatto" = Gauss-Lobatto from the roots of P'_{n-1}."""
from __future__ import annotations

import numpy as np


def quadrature(grid: str, nlat: int):
    """(theta_k ascending from the north pole, w_k): nodes in colatitude and the weights of the integral over cos(theta)."""
    if grid == "equiangular":
        n1 = nlat - 1
        k = np.arange(nlat)
        theta = np.pi * k / n1
        w = np.ones(nlat)
        for j in range(1, n1 // 2 + 1):
            b = 1.0 if 2 * j == n1 else 2.0
            w -= b / (4.0 * j * j - 1.0) * np.cos(2.0 * j * theta)
        c = np.where((k == 0) | (k == n1), 1.0, 2.0)
        return theta, c * w / n1
    if grid == "legendre-gauss":
        x, w = np.polynomial.legendre.leggauss(nlat)  # x ascending: theta descending -> reverse both (the rule is symmetric)
        return np.arccos(x)[::-1].copy(), w[::-1].copy()
    if grid == "lobatto":  # Gauss-Lobatto: the end points and the roots of P'_{n-1}; w = 2 / (n (n-1) P_{n-1}(x)^2)  (quadrature.py:45-85)
        Pn = np.polynomial.legendre.Legendre.basis(nlat - 1)
        x = np.concatenate(([-1.0], np.sort(Pn.deriv().roots().real), [1.0]))
        w = 2.0 / (nlat * (nlat - 1) * Pn(x) ** 2)
        return np.arccos(np.clip(x, -1.0, 1.0))[::-1].copy(), w[::-1].copy()
    raise NotImplementedError(f"SHT grid {grid!r} (built: 'equiangular', 'legendre-gauss', 'lobatto')")


def legendre(mmax: int, lmax: int, theta: np.ndarray) -> np.ndarray:
    """Pbar[m][l][k] = (-1)^m Nbar_l^m P_l^m(cos theta_k), zero for l < m."""
    x = np.cos(theta)
    s2 = (1.0 + x) * (1.0 - x)
    n = max(mmax, lmax)
    P = np.zeros((n, n, len(x)))
    P[0, 0] = 1.0 / np.sqrt(4.0 * np.pi)
    for m in range(n):
        if m > 0:  # sectoral: from the one before
            P[m, m] = np.sqrt((2 * m + 1) * s2 / (2.0 * m)) * P[m - 1, m - 1]
        if m + 1 < n:
            P[m, m + 1] = np.sqrt(2.0 * m + 3.0) * x * P[m, m]
        for l in range(m + 2, n):
            a = np.sqrt((2.0 * l - 1.0) * (2.0 * l + 1.0) / ((l - m) * (l + m)))
            b = np.sqrt((2.0 * l + 1.0) * (l + m - 1.0) * (l - m - 1.0) / ((2.0 * l - 3.0) * (l - m) * (l + m)))
            P[m, l] = a * x * P[m, l - 1] - b * P[m, l - 2]
    P = P[:mmax, :lmax].copy()
    P[1::2] *= -1.0  # Condon-Shortley
    return P


def tables(nlat: int, nlon: int, lmax: int, mmax: int, grid: str = "equiangular", norm: str = "ortho"):
    """(tw [nlon][mmax][2] = (cos, sin)(2 pi j m / nlon), A [mmax][lmax][nlat], B [mmax][lmax][nlat]) in float64."""
    if norm != "ortho":
        raise NotImplementedError(f"SHT norm {norm!r} (built: 'ortho', the reference's SFNO default)")
    if mmax > nlon // 2 + 1:
        raise ValueError(f"mmax = {mmax} orders do not exist on {nlon} longitudes")
    theta, w = quadrature(grid, nlat)
    P = legendre(mmax, lmax, theta)
    A = P * w[None, None, :] * (2.0 * np.pi / nlon)
    c = np.full(mmax, 2.0)
    c[0] = 1.0
    if nlon % 2 == 0 and mmax > nlon // 2:
        c[nlon // 2] = 1.0
    B = P * c[:, None, None]
    j, m = np.arange(nlon)[:, None], np.arange(mmax)[None, :]
    ang = 2.0 * np.pi * ((j * m) % nlon) / nlon
    tw = np.stack([np.cos(ang), np.sin(ang)], axis=-1)
    return tw, A, B


def kernel_layouts(T: np.ndarray):
    """A table [m][l][k] in the two storage orders of csrc/sht.hip: ([k][l][m] for the analysis kernel, [l][k][m] for the synthesis
    kernel) -- the threads of a wave then read consecutive addresses."""
    return np.ascontiguousarray(T.transpose(2, 1, 0)), np.ascontiguousarray(T.transpose(1, 2, 0))
