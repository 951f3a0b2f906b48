from . import logger, misc  # noqa: F401
from .symbolic import lambdify  # noqa: F401
