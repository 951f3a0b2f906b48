from . import logger, misc  # noqa: F401
from .symbolic import lambdify  # noqa: F401
from . import expression  # noqa: F401,E402
from .expression import ExpressionSolver  # noqa: F401,E402
