"""Point samplers in [0,1]^d (/root/reference/ppsci/geometry/sampler.py:30-92).

"pseudo" is numpy's global RNG exactly as the reference calls it (bit-exact, pinned by the geometry fixtures).

The quasi-random ones (LHS / Halton / Hammersley / Sobol) come from a THIRD-PARTY dependency of the reference that is in
neither tree: scikit-optimize (`skopt.sampler.*`; /root/reference/requirements.txt:14 names it WITHOUT a version).  They
are restated here from the published definitions of the sequences, with the reference's own call-site rules
(sampler.py:60-92: `generate(space, n + skip)[skip:]` -- Halton starts at index 1, Hammersley drops its first point, Sobol
drops [0, ...] and from three dimensions on also [0.5, ...]):
  Halton      van der Corput radical inverses phi_b(i) = sum_k d_k(i) b^(-k-1) in the first d primes, i = 1 .. n
              (Halton 1960) -- own implementation, PINNED against the independent scipy.stats.qmc.Halton(scramble=False)
              (tests/test_geometry.py)
  Hammersley  d = 1: Halton; d > 1: (i / N, phi_2(i), phi_3(i), ...), i = 1 .. n, N = n + 1 (Hammersley 1960, the
              textbook arrangement) -- skopt's own arrangement of the equidistant coordinate cannot be checked offline:
              for THIS sampler parity with the reference's dependency stays unpinned
  Sobol       unscrambled Sobol' points with the Joe-Kuo (2008) direction numbers as shipped in scipy.stats.qmc.Sobol;
              pinned by the published first points of the 2-D sequence (tests/test_geometry.py)
  LHS         classic Latin hypercube (McKay 1979): one stratified draw per cell, independently permuted per dimension,
              from numpy's global RNG (so `set_random_seed` controls it); skopt's draw order cannot be checked offline"""
import warnings

import numpy as np

from ..utils.misc import DEFAULT_DTYPE


def pseudorandom(n_samples: int, ndim: int) -> np.ndarray:
    # numpy's *global* RNG, exactly one call, then the cast (sampler.py:49-57): bit-exact with the reference
    return np.random.random(size=(n_samples, ndim)).astype(dtype=DEFAULT_DTYPE)


_PRIMES = (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53)


def radical_inverse(idx: np.ndarray, base: int) -> np.ndarray:
    """van der Corput: mirror the base-`base` digits of every index about the radix point (float64)."""
    idx = np.asarray(idx, dtype=np.int64).copy()
    out = np.zeros(idx.shape, np.float64)
    scale = 1.0 / base
    while np.any(idx > 0):
        out += (idx % base) * scale
        idx //= base
        scale /= base
    return out


def _halton(n: int, ndim: int, skip: int) -> np.ndarray:
    if ndim > len(_PRIMES):
        raise NotImplementedError(f"Halton / Hammersley points in {ndim} dimensions (built: up to {len(_PRIMES)})")
    i = np.arange(skip, n + skip)
    return np.stack([radical_inverse(i, _PRIMES[d]) for d in range(ndim)], axis=1)


def quasirandom(n_samples: int, ndim: int, method: str) -> np.ndarray:
    """sampler.py:60-92."""
    if method == "LHS":
        cells = np.stack([np.random.permutation(n_samples) for _ in range(ndim)], axis=1)
        pts = (cells + np.random.random(size=(n_samples, ndim))) / n_samples
    elif method == "Halton":
        pts = _halton(n_samples, ndim, 1)  # 1st point: [0, 0, ...]
    elif method == "Hammersley":
        if ndim == 1:
            pts = _halton(n_samples, 1, 1)
        else:
            total = n_samples + 1
            first = (np.arange(total, dtype=np.float64) / total)[1:, None]
            pts = np.concatenate([first, _halton(n_samples, ndim - 1, 1)], axis=1)
    elif method == "Sobol":
        from scipy.stats import qmc

        skip = 1 if ndim < 3 else 2  # 1st point: [0, 0, ...], 2nd point: [0.5, 0.5, ...]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", UserWarning)  # "n is not a power of two": balance is not needed here
            pts = qmc.Sobol(d=ndim, scramble=False).random(n_samples + skip)[skip:]
    else:
        raise ValueError(f"Sampling method({method}) is not available.")
    return np.asarray(pts, dtype=DEFAULT_DTYPE)


def sample(n_samples: int, ndim: int, method: str = "pseudo") -> np.ndarray:
    if method == "pseudo":
        return pseudorandom(n_samples, ndim)
    if method in ("LHS", "Halton", "Hammersley", "Sobol"):
        return quasirandom(n_samples, ndim, method)
    raise ValueError(f"Sampling method({method}) is not available.")
