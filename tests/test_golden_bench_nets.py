"""Parity of the EXACT networks bench.py times (BASELINE configs[0..2]: Laplace2D 3x20, Allen-Cahn 4x64 without
periods, LDC NavierStokes 5x128; and Allen-Cahn at the reference yaml's own shape, 4x256 with the period embedding of x:
examples/allen_cahn/conf/allen_cahn.yaml:38-42) on the first 2 048 points of their bench batches, against
tests/golden/bench_nets.npz -- produced by executing the REFERENCE's own hot-path code in float64 under the
torch-backed paddle shim (tests/golden/make_bench_nets_golden.py).

1. the CPU oracle (oracle/ref_torch.py) reproduces residuals, losses and parameter gradients to ~1e-9;
2. the HIP path through the ppsci API matches them within the fp32 tolerance of the north star:
   residual rel-L2 <= 1e-5, gradient rel-L2 <= 1e-4, loss rel <= 5e-5 (MI355X: all 2 048 points; the CPU
   emulator: the residuals of the first 48 points, which is what it finishes in seconds at width 128)."""
import os

import numpy as np
import pytest
import torch

import ppsci
from oracle import ref_torch as R
from oracle import taylor_np as T
from tests.common import make_dev_fixture, rel
from tests.golden.make_bench_nets_golden import CASES, bench_weights

dev = make_dev_fixture()
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bench_nets.npz"))


def _keys(name):
    return [k.split("/")[-1] for k in GOLD.files if k.startswith(f"{name}/res/")]


def _flat(name, c):
    flat = bench_weights(c.get("d_in", len(c["inputs"])), c["hidden"], len(c["outputs"]))
    cs = GOLD[f"{name}/param_checksum"]
    assert float(flat.astype(np.float64).sum()) == cs[0] and float(np.abs(flat.astype(np.float64)).sum()) == cs[1]
    return flat


def _equation(c):
    return {"laplace": lambda: ppsci.equation.Laplace(2), "allen_cahn": lambda: ppsci.equation.AllenCahn(0.01),
            "navier_stokes": lambda: ppsci.equation.NavierStokes(0.01, 1.0, 2, False)}[c["eq"]]()


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_reproduces_reference_run_on_bench_nets(name):
    c = CASES[name]
    flat = _flat(name, c).astype(np.float64)
    # (periods: {key: [period, trainable]} of the reference -> the oracle's {input index: omega = 2 pi / period in fp32})
    per = {c["inputs"].index(k): float(np.float32(2 * np.pi / float(p))) for k, (p, _) in (c.get("periods") or {}).items()}
    net = T.make_net(len(c["inputs"]), c["hidden"], len(c["outputs"]), periods=per)
    off = 0
    for i in range(len(net.weights)):
        n = net.weights[i].size
        net.weights[i] = flat[off:off + n].reshape(net.weights[i].shape)
        off += n
        n = net.biases[i].size
        net.biases[i] = flat[off:off + n]
        off += n
    X = GOLD[f"{name}/X"].astype(np.float64)
    model = R.MLP(c["inputs"], c["outputs"], net)
    if c["eq"] == "allen_cahn":
        exprs = {"allen_cahn": R.allen_cahn_fn(0.01)}
    else:
        sym = R.laplace_exprs(2) if c["eq"] == "laplace" else R.navier_stokes_exprs(0.01, 1.0, 2, False)
        exprs = {k: R.lambdify(e, model) for k, e in sym.items()}
    keys = _keys(name)
    n = X.shape[0]
    w = None if not c.get("weight") else {k: np.full((n, 1), float(np.float32(c["weight"]))) for k in keys}
    cst = dict(name="EQ", input={k: X[:, j:j + 1] for j, k in enumerate(c["inputs"])}, exprs=exprs,
               label={k: np.zeros((n, 1)) for k in keys}, weight=w, reduction=c["reduction"])
    total, losses, g, outs = R.loss_and_grads(model, [cst])
    for k in keys:
        assert rel(outs[0][k].detach().numpy()[:, 0], GOLD[f"{name}/res/{k}"]) < 1e-9
        assert losses[k] == pytest.approx(float(GOLD[f"{name}/loss/{k}"]), rel=1e-9)
    assert rel(g, GOLD[f"{name}/grad"]) < 1e-8


@pytest.mark.parametrize("name", list(CASES))
def test_hip_path_matches_reference_run_on_bench_nets(name, dev, tmp_path):
    c = CASES[name]
    keys = _keys(name)
    n = GOLD[f"{name}/X"].shape[0] if dev == "gpu" else 48
    X = GOLD[f"{name}/X"][:n]
    model = ppsci.arch.MLP(c["inputs"], c["outputs"], len(c["hidden"]), c["hidden"][0], "tanh", periods=c.get("periods"))
    model.flat_params.copy_(torch.tensor(_flat(name, c)).to(model.flat_params.device))
    eq = _equation(c)
    inp = {k: X[:, j:j + 1] for j, k in enumerate(c["inputs"])}
    if dev == "gpu":
        w = None if not c.get("weight") else {k: np.full((n, 1), c["weight"], np.float32) for k in keys}
        cfg = {"dataset": {"name": "IterableNamedArrayDataset", "input": inp,
                           "label": {k: np.zeros((n, 1), np.float32) for k in keys}, "weight": w}}
        cst = ppsci.constraint.SupervisedConstraint(cfg, ppsci.loss.MSELoss(c["reduction"]), eq.equations, name="EQ")
        solver = ppsci.solver.Solver(model, {"EQ": cst}, str(tmp_path), ppsci.optimizer.Adam(1e-3)(model), epochs=1,
                                     iters_per_epoch=1)
        solver.engine.forward_backward([solver._compiled["EQ"].fused])
        losses = solver._compiled["EQ"].fused.losses()
        for k in keys:
            assert losses[k] == pytest.approx(float(GOLD[f"{name}/loss/{k}"]), rel=5e-5), k
        assert rel(solver.engine.grad.cpu().numpy(), GOLD[f"{name}/grad"]) < 1e-4
    else:
        solver = ppsci.solver.Solver(model, None, str(tmp_path))
    res = solver.predict(inp, eq.equations, batch_size=None, return_numpy=True)
    for k in keys:
        assert rel(res[k][:, 0], GOLD[f"{name}/res/{k}"][:n]) < 1e-5, k
