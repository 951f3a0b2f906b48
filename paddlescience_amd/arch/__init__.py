from .base import Arch  # noqa: F401
from .fno import FNONet, SFNONet, SpectralConv2d, TFNO1dNet, TFNO2dNet, TFNO3dNet  # noqa: F401
from .uno import UNONet  # noqa: F401
from .mlp import MLP  # noqa: F401
from .model_list import ModelList  # noqa: F401
from .piratenet import PirateNet  # noqa: F401
from .modified_mlp import ModifiedMLP  # noqa: F401
from .spinn import SPINN  # noqa: F401

__all__ = ["Arch", "MLP", "PirateNet", "ModifiedMLP", "ModelList", "SpectralConv2d", "SPINN", "FNONet", "TFNO1dNet", "TFNO2dNet", "TFNO3dNet",
           "UNONet", "SFNONet", "build_model"]


def build_model(cfg):
    """ppsci/arch/__init__.py build_model: cfg is {ClassName: kwargs} or a list of such dicts."""
    cfg = dict(cfg) if not isinstance(cfg, (list, tuple)) else cfg
    classes = {"MLP": MLP, "PirateNet": PirateNet, "ModifiedMLP": ModifiedMLP, "SPINN": SPINN, "FNONet": FNONet, "TFNO2dNet": TFNO2dNet, "UNONet": UNONet, "SFNONet": SFNONet}
    if isinstance(cfg, (list, tuple)):  # a list of {ClassName: kwargs} -> ModelList (arch/__init__.py build_model)
        return ModelList(tuple(classes[name](**kwargs) for item in cfg for name, kwargs in dict(item).items()))
    (name, kwargs), = cfg.items()
    return classes[name](**kwargs)
