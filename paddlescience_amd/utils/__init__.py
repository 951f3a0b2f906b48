from . import logger, misc  # noqa: F401
from .symbolic import lambdify  # noqa: F401
from . import expression  # noqa: F401,E402
from .expression import ExpressionSolver  # noqa: F401,E402
from . import reader, save_load  # noqa: F401,E402
from .misc import set_random_seed  # noqa: F401,E402
from .save_load import load_checkpoint, load_pretrain, save_checkpoint  # noqa: F401,E402
from . import initializer  # noqa: F401,E402
