"""The C-ABI library loads on a GPU-less host and exports every function include/ppsci_hip.h declares
(no compute calls here).  Also: the product loader refuses the emulator build and has no CPU fallback."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "ppsci_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(ppsci_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


@pytest.fixture(scope="module")
def libpath():
    import __graft_entry__ as g

    if not os.path.exists(g.LIB):
        g.build()
    return g.LIB


def test_header_functions_are_exported(libpath):
    lib = ctypes.CDLL(libpath)
    names = _declared_functions()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ppsci_hip.h but not exported"
    lib.ppsci_is_device_build.restype = ctypes.c_int
    assert lib.ppsci_is_device_build() == 1


def test_python_binding_table_matches_header():
    from paddlescience_amd import _lib

    assert sorted(_lib.EXPORTED_SYMBOLS) == _declared_functions()


def test_struct_sizes_match_header_constants():
    from paddlescience_amd import _lib as L

    assert ctypes.sizeof(L.MlpDesc) == 4 * 8 + 4 * L.MAX_IN + 4 * L.MAX_IN + 4 * L.MAX_DIRS * L.MAX_IN + 8 + 8  # + n3, n4
    assert ctypes.sizeof(L.Instr) == 16 and ctypes.sizeof(L.Residual) == 24
    assert ctypes.sizeof(L.EpilogueDesc) == 20 + 16 * L.MAX_PROG + 24 * L.MAX_RES


def test_no_cpu_fallback():
    """Without the emulator injected, CPU tensors are refused and a missing device library is a loud error."""
    import torch

    from paddlescience_amd import _lib, hotpath as hp

    _lib._inject_for_tests(None)
    with pytest.raises(RuntimeError):
        hp.adam_step(torch.zeros(4), torch.zeros(4), torch.zeros(4), torch.zeros(4), 1e-3, 1)
    from tests.emu import build_emu

    emu = build_emu.build()
    lib = _lib._bind(emu)
    assert lib.ppsci_is_device_build() == 0  # the product loader (`_lib.lib()`) rejects such a build
