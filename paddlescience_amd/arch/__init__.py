from .base import Arch  # noqa: F401
from .fno import SpectralConv2d, spectral_contract  # noqa: F401
from .mlp import MLP  # noqa: F401
from .spinn import SPINN  # noqa: F401

__all__ = ["Arch", "MLP", "SpectralConv2d", "spectral_contract", "SPINN", "build_model"]


def build_model(cfg):
    """ppsci/arch/__init__.py build_model: cfg is {ClassName: kwargs} or a list of such dicts."""
    cfg = dict(cfg) if not isinstance(cfg, (list, tuple)) else cfg
    if isinstance(cfg, (list, tuple)):
        raise NotImplementedError("ModelList is not supported on the fused HIP path yet")
    (name, kwargs), = cfg.items()
    return {"MLP": MLP, "SPINN": SPINN}[name](**kwargs)
