"""Drop-in alias: `import ppsci` resolves to the MI355X-native implementation (paddlescience_amd), so
example scripts written against /root/reference/ppsci keep their import lines.

Every `ppsci.<x>[.<y>...]` import -- `from ppsci.utils import logger`, `import ppsci.visualize.vtu`,
`from ppsci.optimizer import lr_scheduler` -- is answered with the module object of the same dotted name under
`paddlescience_amd` (one module, two names: no second copy of any state such as the logger or the autodiff
cache).  A name the native package does not have raises ModuleNotFoundError like any missing module."""
import importlib
import importlib.abc
import importlib.util
import sys

import paddlescience_amd as _impl
from paddlescience_amd import *  # noqa: F401,F403
from paddlescience_amd import (arch, autodiff, constraint, data, equation, functional, geometry, loss, metric, optimizer,  # noqa: F401
                               solver, utils, validate, visualize)

_PREFIX, _TARGET = "ppsci.", "paddlescience_amd."


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, target: str):
        self.target = target

    def create_module(self, spec):
        module = importlib.import_module(self.target)
        self._own_spec = getattr(module, "__spec__", None)
        return module

    def exec_module(self, module):
        # already executed under its own name; the import system has just pointed __spec__ at the alias: put it back, so
        # that relative imports inside the module keep seeing __package__ == __spec__.parent
        if self._own_spec is not None:
            module.__spec__ = self._own_spec


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(_PREFIX):
            return None
        real = _TARGET + fullname[len(_PREFIX):]
        try:
            found = importlib.util.find_spec(real)
        except (ImportError, AttributeError, ValueError):
            found = None
        if found is None:
            return None
        spec = importlib.util.spec_from_loader(fullname, _AliasLoader(real), is_package=found.submodule_search_locations is not None)
        return spec


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())

# the subpackages imported above are already in sys.modules under their native names: publish them under the alias too
for _name, _mod in list(sys.modules.items()):
    if _name.startswith(_TARGET):
        sys.modules.setdefault(_PREFIX + _name[len(_TARGET):], _mod)
lambdify = _impl.lambdify
