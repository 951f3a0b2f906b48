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
    assert ctypes.sizeof(L.Instr) == 16 and ctypes.sizeof(L.Residual) == 28
    assert ctypes.sizeof(L.EpilogueDesc) == 20 + 16 * L.MAX_PROG + 28 * L.MAX_RES


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


def test_invalid_arguments_are_reported_not_executed(libpath):
    """Error behaviour of the entry points added in round 2 (no device work: every call is rejected by its argument check):
    a non-zero status and a message behind ppsci_last_error(), as for the round-1 entry points."""
    from paddlescience_amd import _lib as L

    lib = L._bind(libpath)

    def bad(rc, needle):
        assert rc != 0
        msg = lib.ppsci_last_error().decode()
        assert needle in msg, msg

    d = L.PirateEmbedDesc()
    bad(lib.ppsci_pirate_embed_fwd(ctypes.byref(d), None, None, None, None), "pirate_embed")  # d_raw = 0
    d.d_raw, d.d0, d.half, d.n1, d.n2, d.N, d.NP = 2, 2, 8, 2, 1, 100, 100  # NP not a multiple of 16
    bad(lib.ppsci_pirate_embed_fwd(ctypes.byref(d), None, None, None, None), "pirate_embed")
    bad(lib.ppsci_pirate_act_fwd(7, L.ACT["tanh"], 16, 32, 32, 1, 1, None, None, None, None, None, None, None, None), "pirate_act")
    bad(lib.ppsci_pirate_act_fwd(L.PIRATE_ACT, L.ACT["relu"], 16, 32, 32, 1, 1, None, None, None, None, None, None, None, None),
        "no PirateNet kernel")
    bad(lib.ppsci_pirate_act_fwd(L.PIRATE_GATE, L.ACT["tanh"], 16, 32, 32, 2, 3, None, None, None, None, None, None, None, None),
        "pirate_act")  # n2 > n1
    bad(lib.ppsci_pirate_out_fwd(0, 1, 8, 16, None, None, None, None), "pirate_out_fwd")
    bad(lib.ppsci_pirate_out_bwd(4, 1, 8, 4, None, None, None), "pirate_out_bwd")  # NP < N
    md = L.ModMlpDesc()
    bad(lib.ppsci_modmlp_fwd_batch(ctypes.byref(md), 3, None, None, None, None, None, None), "modmlp")
    bad(lib.ppsci_linear_pad(4, 4, 2, 8, None, None, None, None, None), "linear_pad")  # destination smaller than the source
    bad(lib.ppsci_pw_conv(1, 8, 8, 6, None, None, 0, None, None, 0, None, None, None), "pw_conv")  # P not a multiple of 4
