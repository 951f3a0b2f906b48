"""ppsci.equation (/root/reference/ppsci/equation/__init__.py:55-76)."""
import copy

from . import ide, pde  # noqa: F401
from .ide import Volterra  # noqa: F401
from .pde import (DETACH_FUNC_NAME, NLSMB, PDE, AllenCahn, Biharmonic, HeatExchanger, Helmholtz, Laplace,  # noqa: F401
                  LinearElasticity, NavierStokes, NormalDotVec, Poisson, Vibration)

__all__ = ["PDE", "DETACH_FUNC_NAME", "AllenCahn", "Biharmonic", "HeatExchanger", "Helmholtz", "Laplace", "LinearElasticity",
           "NavierStokes", "NormalDotVec", "Poisson", "Vibration", "Volterra", "NLSMB", "build_equation"]


def build_equation(cfg):
    if cfg is None:
        return None
    cfg = copy.deepcopy(cfg)
    eq_dict = {}
    for item in cfg:
        cls = next(iter(item.keys()))
        kwargs = item[cls]
        eq_dict[cls] = globals()[cls](**kwargs)
    return eq_dict
