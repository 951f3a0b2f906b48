from . import mtl  # noqa: F401
from .base import Loss  # noqa: F401
from .func import FunctionalLoss  # noqa: F401
from .mse import MSELoss  # noqa: F401

__all__ = ["Loss", "MSELoss", "FunctionalLoss", "mtl", "build_loss"]


def build_loss(cfg):
    cfg = dict(cfg)
    cls = cfg.pop("name")
    return {"MSELoss": MSELoss, "FunctionalLoss": FunctionalLoss}[cls](**cfg)
