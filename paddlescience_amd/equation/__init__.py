"""ppsci.equation (/root/reference/ppsci/equation/__init__.py:55-76)."""
import copy

from . import ide, pde  # noqa: F401
from .ide import Volterra  # noqa: F401
from .pde import PDE, AllenCahn, Biharmonic, Helmholtz, Laplace, NavierStokes, Poisson, Vibration  # noqa: F401

__all__ = ["PDE", "AllenCahn", "Biharmonic", "Helmholtz", "Laplace", "NavierStokes", "Poisson", "Vibration", "Volterra", "build_equation"]


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
