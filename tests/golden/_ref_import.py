"""TEST INFRASTRUCTURE, container-only: import pure-numpy / pure-sympy parts of the reference
(/root/reference/ppsci) WITHOUT PaddlePaddle, to generate golden vectors.

`paddle` (and other absent third-party packages) are replaced by permissive dummy modules; `ppsci`,
`ppsci.utils`, ... are registered as bare packages so that their heavy `__init__.py` files do not run
and only the requested sub-modules are executed from the reference tree.  Nothing here ships: the
generated fixtures under tests/golden/*.npz / *.json are what the tests read, because /root/reference
does not exist on the GPU box."""
import importlib
import importlib.abc
import importlib.machinery
import os
import sys
import types

REF = "/root/reference"
DUMMY_ROOTS = ("paddle", "hydra", "omegaconf", "colorlog", "pysdf", "skopt", "pymesh", "open3d", "warp", "h5py",
               "meshio", "pyevtk", "visualdl", "wandb", "tensorboardX", "pgl", "sklearn", "scipy", "matplotlib",
               "tqdm", "requests", "imageio", "seaborn", "pandas", "xarray", "cftime", "netCDF4")


class _Dummy(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        full = f"{self.__name__}.{name}"
        if full in sys.modules:
            return sys.modules[full]

        class _Anything:
            def __init__(self, *a, **k):
                pass

            def __call__(self, *a, **k):
                return _Anything()

            def __getattr__(self, n):
                return _Anything()

            def __iter__(self):
                return iter(())

            def __mro_entries__(self, bases):
                return (object,)

        _Anything.__name__ = name
        return _Anything


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in DUMMY_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _Dummy(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        if module.__name__ == "paddle":
            module.get_default_dtype = lambda: "float32"


def install(force_dummy=("scipy", "sklearn", "matplotlib", "pandas", "tqdm", "requests")):
    # packages that exist here but are irrelevant and slow/fragile to import through the reference
    for k in list(sys.modules):
        if k.split(".")[0] in DUMMY_ROOTS and not isinstance(sys.modules[k], _Dummy) and k.split(".")[0] not in force_dummy:
            pass
    sys.meta_path.insert(0, _Finder())
    for pkg in ("ppsci", "ppsci.utils", "ppsci.geometry", "ppsci.equation", "ppsci.equation.pde", "ppsci.autodiff",
                "ppsci.arch", "ppsci.loss", "ppsci.data"):
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(REF, *pkg.split("."))]
        sys.modules[pkg] = m
    # the few helpers the geometry code takes from ppsci.utils
    logger = types.ModuleType("ppsci.utils.logger")
    for fn in ("info", "warning", "message", "debug", "error"):
        setattr(logger, fn, lambda *a, **k: None)
    sys.modules["ppsci.utils.logger"] = logger
    sys.modules["ppsci.utils"].logger = logger
    misc = importlib.import_module("ppsci.utils.misc")
    sys.modules["ppsci.utils"].misc = misc
    return misc


def ref_module(name: str):
    return importlib.import_module(name)
