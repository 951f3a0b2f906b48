"""Point samplers in [0,1]^d (/root/reference/ppsci/geometry/sampler.py:30-92).  Only the pseudo-random
sampler is available: the quasi-random ones (LHS / Halton / Hammersley / Sobol) come from scikit-optimize,
which is not in this image."""
import numpy as np

from ..utils.misc import DEFAULT_DTYPE


def pseudorandom(n_samples: int, ndim: int) -> np.ndarray:
    # numpy's *global* RNG, exactly one call, then the cast (sampler.py:49-57): bit-exact with the reference
    return np.random.random(size=(n_samples, ndim)).astype(dtype=DEFAULT_DTYPE)


def sample(n_samples: int, ndim: int, method: str = "pseudo") -> np.ndarray:
    if method == "pseudo":
        return pseudorandom(n_samples, ndim)
    if method in ("LHS", "Halton", "Hammersley", "Sobol"):
        raise NotImplementedError(f"quasi-random sampler {method!r} needs scikit-optimize, which is not installed")
    raise ValueError(f"Sampling method({method}) is not available.")
