"""Point samplers in [0,1]^d (/root/reference/ppsci/geometry/sampler.py:30-92).

"pseudo" is numpy's global RNG exactly as the reference calls it (bit-exact, pinned by the geometry fixtures).

The quasi-random ones (LHS / Halton / Hammersley / Sobol) come from scikit-optimize in the reference
(`skopt.sampler.*`), which is neither in the reference tree nor in this image: they are restated here from the
sequences' definitions with the reference's skip rules (sampler.py:60-92: drop the all-zero first point, and for Sobol
also [0.5, ...]) -- PARITY UNPINNED for these four (no stored vectors anywhere in the reference, skopt's direction
numbers / LHS optimisation cannot be checked):
  Halton      van der Corput radical inverses in the first d primes, indices 1 .. n        (scipy.stats.qmc.Halton)
  Hammersley  d = 1: Halton; d > 1: (i / N, Halton_{d-1}(i)), i = 1 .. n, N = n + 1
  Sobol       unscrambled Sobol' points (scipy.stats.qmc.Sobol, Joe-Kuo direction numbers), first 1 (d < 3) or 2 dropped
  LHS         classic Latin hypercube: one stratified draw per cell, independently permuted per dimension, from numpy's
              global RNG (so `set_random_seed` controls it)"""
import warnings

import numpy as np

from ..utils.misc import DEFAULT_DTYPE


def pseudorandom(n_samples: int, ndim: int) -> np.ndarray:
    # numpy's *global* RNG, exactly one call, then the cast (sampler.py:49-57): bit-exact with the reference
    return np.random.random(size=(n_samples, ndim)).astype(dtype=DEFAULT_DTYPE)


def _halton(n: int, ndim: int, skip: int) -> np.ndarray:
    from scipy.stats import qmc

    return qmc.Halton(d=ndim, scramble=False).random(n + skip)[skip:]


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
