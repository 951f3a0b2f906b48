from .allen_cahn import AllenCahn  # noqa: F401
from .base import PDE  # noqa: F401
from .biharmonic import Biharmonic  # noqa: F401
from .helmholtz import Helmholtz  # noqa: F401
from .laplace import Laplace  # noqa: F401
from .navier_stokes import NavierStokes  # noqa: F401
from .poisson import Poisson  # noqa: F401
from .viv import Vibration  # noqa: F401
