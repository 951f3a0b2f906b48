"""Compile-time residual programs (csrc/epi_static.h, epi_static_programs.h, tools/gen_static_programs.py).

The fused tile kernel evaluates a residual program that IS one of the generated tables as straight-line code on every
wave; any other program runs on the epilogue VM.  Checked here:
  * the committed header is what the generator produces from the package's own lowering (so the tables cannot drift from
    what `ppsci.equation.*` lowers to), and the generator's Python port of the pre-decoder equals the C one;
  * static vs VM vs opcode interpreter: the same gradients, loss terms, residuals and dL/dU (same operations in the same
    order: bit for bit on the emulator; on the GPU up to the compiler's choice of which product of a sum it fuses);
  * the API path (Solver-style constraint on `ppsci.equation.AllenCahn`) lands on its table; a program that is in no table
    lands on the VM and still agrees with the separate launches.
Reference: /root/reference/ppsci/equation/pde/allen_cahn.py:56-64, laplace.py:40-55, navier_stokes.py:96-160, loss/mse.py:82-105."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from paddlescience_amd import _lib as L
from paddlescience_amd import device
from paddlescience_amd import hotpath as hp
from paddlescience_amd.engine import Engine
from tests.common import make_dev_fixture, rel
from tests.test_one_launch import _constraint, _program, _weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = make_dev_fixture()


def test_committed_header_is_what_the_generator_produces():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_static_programs.py"), "--check"], capture_output=True,
                       text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr


def test_python_predecoder_equals_the_c_one(dev):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_static_programs as G

    for kind in ("allen_cahn", "laplace", "value", "streams:2,1", "streams:2,2"):
        ed, _, _ = _program(kind, 1000)
        out = (C.c_uint32 * 256)()
        nl = C.c_int(0)
        ns = L.lib().ppsci_epilogue_predecode(C.byref(ed), out, C.byref(nl))
        steps, loads, terms = G.predecode(ed)
        assert ns == len(steps) and nl.value == len(loads)
        assert list(out[:ns]) == steps and list(out[64:64 + nl.value]) == loads and list(out[128:128 + ed.n_res]) == terms


def _run(d, lay, kind, n, flat, steps, static, fast=1, fused=True):
    lib = L.lib()
    lib.ppsci_set_static_program(static)
    lib.ppsci_set_fast_program(fast)
    lib.ppsci_set_step_tail(1)
    try:
        params = torch.tensor(flat, device=d)
        eng = Engine(lay, params)
        eng.one_launch = fused
        c = _constraint(d, kind, lay, n, 101)
        c.step_outputs = True
        grads, losses = [], []
        for _ in range(steps):
            eng.train_step([c], 1e-2)
            grads.append(eng.grad.detach().cpu().numpy().copy())
            losses.append(c.loss_terms.detach().cpu().numpy().copy())
        name = c._step_plan.static_program if fused else None
        return dict(p=params.detach().cpu().numpy(), g=grads, l=losses, r=c.resid.detach().cpu().numpy().copy(),
                    U=c.U.detach().cpu().numpy().copy(), Ub=c.Ubar.detach().cpu().numpy().copy(), name=name)
    finally:
        lib.ppsci_set_static_program(1)
        lib.ppsci_set_fast_program(1)
        lib.ppsci_set_step_tail(-1)


@pytest.mark.parametrize("kind,table,act,depth,width", [
    ("allen_cahn", "allen_cahn_handbuilt", "tanh", 4, 64),   # BASELINE configs[1]'s net and stream set
    ("laplace", "test_laplace", "tanh", 3, 50),              # two terms, label + weight arrays, an input operand
    ("value", "test_value", "silu", 2, 40),                  # S = 1, a label
    ("allen_cahn", "allen_cahn_handbuilt", "sin", 5, 33),
])
def test_static_program_equals_the_vm_and_the_interpreter(dev, kind, table, act, depth, width):
    d = device.get_device()
    lay = hp.NetLayout(2, depth, width, 1, act)
    flat = _weights(lay, 13)
    n = 90 if dev != "gpu" else 3000  # (a ragged last tile)
    a = _run(d, lay, kind, n, flat, 2, static=1)
    b = _run(d, lay, kind, n, flat, 2, static=0)
    c = _run(d, lay, kind, n, flat, 2, static=0, fast=0)
    assert a["name"] == table and b["name"] == "" and c["name"] == ""
    # the VM and the interpreter are one kernel binary: bit-identical.  The compile-time program is another instantiation of
    # the kernel: the same operations in the same order, but hipcc is free to pick WHICH product of `w0 h0 + w1 h1` it fuses
    # into the add (observed on MI355X: U differs in the last bit at a few points, everything else is bit-identical); the
    # emulator build does not contract at all, so there the three agree bit for bit.
    same = np.array_equal if dev != "gpu" else (lambda x, y: rel(x, y) < 5e-7)
    assert np.array_equal(b["p"], c["p"]) and all(np.array_equal(x, y) for x, y in zip(b["g"], c["g"]))
    for o in (b, c):
        assert same(a["p"], o["p"])
        for k in ("g", "l"):
            assert all(same(x, y) for x, y in zip(a[k], o[k])), k
        for k in ("r", "U", "Ub"):
            assert same(a[k], o[k]), k
    assert np.abs(a["g"][0]).max() > 0 and np.isfinite(a["p"]).all()
    # ... and the separate launches (pinned to the reference by the golden tests) up to the summation order
    a1 = _run(d, lay, kind, n, flat, 1, static=1)
    s = _run(d, lay, kind, n, flat, 1, static=0, fused=False)
    assert rel(a1["g"][0], s["g"][0]) < 3e-6 and rel(a1["r"], s["r"]) < 1e-6 and rel(a1["Ub"], s["Ub"]) < 1e-6


def test_program_outside_the_tables_runs_on_the_vm(dev):
    d = device.get_device()
    lay = hp.NetLayout(2, 3, 64, 1, "tanh")
    flat = _weights(lay, 5)
    n = 70 if dev != "gpu" else 2000
    a = _run(d, lay, "streams:2,1", n, flat, 1, static=1)
    s = _run(d, lay, "streams:2,1", n, flat, 1, static=0, fused=False)
    assert a["name"] == ""
    assert rel(a["g"][0], s["g"][0]) < 3e-6


def test_api_constraint_on_allen_cahn_lands_on_its_table(dev):
    """ppsci.equation.AllenCahn through the constraint compiler: the lowered program is table `allen_cahn`."""
    import paddlescience_amd as ppsci
    from paddlescience_amd.compile import CompiledConstraint

    d = device.get_device()
    torch.manual_seed(0)
    model = ppsci.arch.MLP(("t", "x"), ("u",), 4, 64, "tanh")
    eq = ppsci.equation.AllenCahn(0.01 ** 2)
    n = 80 if dev != "gpu" else 4096
    rng = np.random.default_rng(0)
    inp = {"t": rng.random((n, 1), dtype=np.float32), "x": rng.random((n, 1), dtype=np.float32) * 2 - 1}
    lab = {"allen_cahn": np.zeros((n, 1), np.float32)}

    def grads(static):
        L.lib().ppsci_set_static_program(static)
        try:
            cc = CompiledConstraint("EQ", model, eq.equations, ("t", "x"), ["allen_cahn"], [], ppsci.loss.MSELoss("mean"), n, n, d)
            cc.bind(inp, lab, None)
            eng = Engine(model.layout, model.flat_params.detach().clone())
            eng.forward_backward([cc.fused])
            return eng.grad.detach().cpu().numpy().copy(), cc.fused._step_plan.static_program
        finally:
            L.lib().ppsci_set_static_program(1)

    g1, name1 = grads(1)
    g0, name0 = grads(0)
    assert name1 == "allen_cahn" and name0 == ""
    assert (np.array_equal(g1, g0) if dev != "gpu" else rel(g1, g0) < 5e-7) and np.abs(g1).max() > 0
