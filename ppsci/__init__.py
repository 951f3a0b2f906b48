"""Drop-in alias: `import ppsci` resolves to the MI355X-native implementation (paddlescience_amd), so
example scripts written against /root/reference/ppsci keep their import lines."""
import sys

import paddlescience_amd as _impl
from paddlescience_amd import *  # noqa: F401,F403
from paddlescience_amd import (arch, autodiff, constraint, data, equation, functional, geometry, loss, metric, optimizer,  # noqa: F401
                               solver, utils, validate)

for _name in ("arch", "autodiff", "constraint", "data", "equation", "functional", "geometry", "loss", "metric", "optimizer", "solver",
              "utils", "validate"):
    sys.modules[f"ppsci.{_name}"] = getattr(_impl, _name)
sys.modules["ppsci.loss.mtl"] = _impl.loss.mtl
sys.modules["ppsci.data.dataset"] = _impl.data.dataset
sys.modules["ppsci.data.dataset.darcyflow_dataset"] = _impl.data.dataset.darcyflow_dataset
sys.modules["ppsci.optimizer.lr_scheduler"] = _impl.optimizer.lr_scheduler
sys.modules["ppsci.utils.misc"] = _impl.utils.misc
sys.modules["ppsci.utils.logger"] = _impl.utils.logger
sys.modules["ppsci.utils.expression"] = _impl.utils.expression
sys.modules["ppsci.utils.reader"] = _impl.utils.reader
sys.modules["ppsci.utils.save_load"] = _impl.utils.save_load
sys.modules["ppsci.utils.save_load"] = _impl.utils.save_load
sys.modules["ppsci.utils.symbolic"] = _impl.utils.symbolic
lambdify = _impl.lambdify
