from . import mtl  # noqa: F401
from .base import Loss  # noqa: F401
from .func import FunctionalLoss  # noqa: F401
from .lp_h1 import H1Loss, H1Loss_train, LpLoss, LpLoss_train  # noqa: F401
from .l1l2 import L1Loss, L2Loss, L2RelLoss, MAELoss  # noqa: F401
from .mse import CausalMSELoss, MSELoss  # noqa: F401
from .periodic import PeriodicL1Loss, PeriodicL2Loss, PeriodicMSELoss  # noqa: F401

__all__ = ["Loss", "MSELoss", "CausalMSELoss", "FunctionalLoss", "L1Loss", "L2Loss", "L2RelLoss", "MAELoss", "PeriodicMSELoss", "PeriodicL1Loss",
           "PeriodicL2Loss", "LpLoss", "LpLoss_train", "H1Loss", "H1Loss_train", "mtl", "build_loss"]


def build_loss(cfg):
    cfg = dict(cfg)
    cls = cfg.pop("name")
    return {"MSELoss": MSELoss, "CausalMSELoss": CausalMSELoss, "FunctionalLoss": FunctionalLoss, "L1Loss": L1Loss, "L2Loss": L2Loss,
            "L2RelLoss": L2RelLoss, "MAELoss": MAELoss, "PeriodicMSELoss": PeriodicMSELoss, "PeriodicL1Loss": PeriodicL1Loss,
            "PeriodicL2Loss": PeriodicL2Loss}[cls](**cfg)
