from .base import Arch  # noqa: F401
from .fno import FNONet, SpectralConv2d, TFNO1dNet, TFNO2dNet, TFNO3dNet, spectral_contract  # noqa: F401
from .mlp import MLP  # noqa: F401
from .spinn import SPINN  # noqa: F401

__all__ = ["Arch", "MLP", "SpectralConv2d", "spectral_contract", "SPINN", "FNONet", "TFNO1dNet", "TFNO2dNet", "TFNO3dNet",
           "build_model"]


def build_model(cfg):
    """ppsci/arch/__init__.py build_model: cfg is {ClassName: kwargs} or a list of such dicts."""
    cfg = dict(cfg) if not isinstance(cfg, (list, tuple)) else cfg
    if isinstance(cfg, (list, tuple)):
        raise NotImplementedError("ModelList is not supported on the fused HIP path yet")
    (name, kwargs), = cfg.items()
    return {"MLP": MLP, "SPINN": SPINN, "FNONet": FNONet, "TFNO2dNet": TFNO2dNet}[name](**kwargs)
