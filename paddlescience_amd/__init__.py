"""paddlescience_amd: the MI355X-native PINN training core behind the ppsci.* Python surface.

`import ppsci` (top-level alias package) gives the reference's import names."""
from . import (arch, autodiff, constraint, data, equation, functional, geometry, loss, metric, optimizer,  # noqa: F401
               solver, utils, validate, visualize)
from .utils.symbolic import lambdify  # noqa: F401

__version__ = "0.1.0"
__all__ = ["arch", "autodiff", "constraint", "data", "equation", "functional", "geometry", "loss", "metric", "optimizer", "solver",
           "utils", "validate", "visualize", "lambdify"]
