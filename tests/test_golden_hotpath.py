"""Parity against tests/golden/hotpath.npz -- produced by executing the REFERENCE's own hot-path code
(ppsci/arch/mlp.py, autodiff/ad.py, utils/symbolic.py with fuse_derivative=True, equation/pde/*.py,
loss/mse.py) in float64 with PaddlePaddle replaced by a torch-backed shim
(tests/golden/make_hotpath_golden.py, tests/golden/_paddle_shim.py).

1. the CPU oracle (oracle/ref_torch.py) reproduces residuals, losses and parameter gradients to ~1e-10:
   this pins the oracle;
2. the HIP path through the ppsci API (emulator here, MI355X with -m gpu) matches them within the fp32
   tolerance: residual rel-L2 <= 1e-5 (north-star bar), gradient rel-L2 <= 1e-4, loss rel <= 5e-5."""
import os

import numpy as np
import pytest
import torch

import ppsci
from oracle import ref_torch as R
from oracle import taylor_np as T
from tests.common import make_dev_fixture, rel, set_model_weights

dev = make_dev_fixture()
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hotpath.npz"))

CASES = {
    "laplace2d_3x20": dict(eq="laplace", inputs=("x", "y"), outputs=("u",), hidden=[20, 20, 20], act="tanh", reduction="sum"),
    "laplace2d_5x20_skip": dict(eq="laplace", inputs=("x", "y"), outputs=("u",), hidden=[20] * 5, act="tanh", reduction="sum", skip=True),
    "allen_cahn_4x64_period": dict(eq="allen_cahn", inputs=("t", "x"), outputs=("u",), hidden=[64] * 4, act="tanh",
                                   reduction="mean", periods={"x": (2.0, False)}),
    "ns2d_3x20_detach": dict(eq="navier_stokes", inputs=("x", "y"), outputs=("u", "v", "p"), hidden=[20] * 3, act="tanh",
                             reduction="sum", detach=("u", "v__y")),
    "ns2d_3x32_silu": dict(eq="navier_stokes", inputs=("x", "y"), outputs=("u", "v", "p"), hidden=[32] * 3, act="silu",
                           reduction="mean"),
    "poisson2d_2x24_sin": dict(eq="poisson", inputs=("x", "y"), outputs=("p",), hidden=[24, 24], act="sin", reduction="mean"),
}


def _keys(name):
    return [k.split("/")[-1] for k in GOLD.files if k.startswith(f"{name}/res/")]


def _net(name, c):
    flat = GOLD[f"{name}/params"]
    pidx = {c["inputs"].index(k): float(np.float32(2 * np.pi / p[0])) for k, p in (c.get("periods") or {}).items()}
    net = T.make_net(len(c["inputs"]), c["hidden"], len(c["outputs"]), activation=c["act"], periods=pidx,
                     skip_connection=c.get("skip", False))
    off = 0
    for i in range(len(net.weights)):
        n = net.weights[i].size
        net.weights[i] = flat[off:off + n].reshape(net.weights[i].shape)
        off += n
        n = net.biases[i].size
        net.biases[i] = flat[off:off + n]
        off += n
    assert off == flat.size
    return net


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_is_pinned_by_reference_run(name, dev):
    if dev != "emu":
        pytest.skip("CPU-only check")
    c = CASES[name]
    net = _net(name, c)
    X = GOLD[f"{name}/X"]
    model = R.MLP(c["inputs"], c["outputs"], net)
    if c["eq"] == "allen_cahn":
        exprs = {"allen_cahn": R.allen_cahn_fn(0.01)}
    else:
        sym = {"laplace": lambda: R.laplace_exprs(2), "poisson": lambda: R.poisson_exprs(2),
               "navier_stokes": lambda: R.navier_stokes_exprs(0.01, 1.0, 2, False)}[c["eq"]]()
        if c.get("detach"):
            eq = ppsci.equation.NavierStokes(0.01, 1.0, 2, False, detach_keys=c["detach"])
            sym = eq.equations
        exprs = {k: R.lambdify(e, model) for k, e in sym.items()}
    keys = _keys(name)
    cst = dict(name="EQ", input={k: X[:, j:j + 1] for j, k in enumerate(c["inputs"])}, exprs=exprs,
               label={k: GOLD[f"{name}/label/{k}"][:, None] for k in keys},
               weight={k: GOLD[f"{name}/weight/{k}"][:, None] for k in keys} if f"{name}/weight/{keys[0]}" in GOLD.files else None,
               reduction=c["reduction"])
    total, losses, g, outs = R.loss_and_grads(model, [cst])
    for k in keys:
        assert rel(outs[0][k].detach().numpy()[:, 0], GOLD[f"{name}/res/{k}"]) < 1e-10
        assert losses[k] == pytest.approx(float(GOLD[f"{name}/loss/{k}"]), rel=1e-10)
    assert rel(g, GOLD[f"{name}/grad"]) < 1e-9


@pytest.mark.parametrize("name", list(CASES))
def test_hip_path_matches_reference_run(name, tmp_path):
    c = CASES[name]
    X = GOLD[f"{name}/X"].astype(np.float32)
    keys = _keys(name)
    model = ppsci.arch.MLP(c["inputs"], c["outputs"], len(c["hidden"]), c["hidden"][0], c["act"],
                           skip_connection=c.get("skip", False), periods=c.get("periods"))
    model.flat_params.copy_(torch.tensor(GOLD[f"{name}/params"], dtype=torch.float32).to(model.flat_params.device))
    eq = {"laplace": lambda: ppsci.equation.Laplace(2), "poisson": lambda: ppsci.equation.Poisson(2),
          "allen_cahn": lambda: ppsci.equation.AllenCahn(0.01),
          "navier_stokes": lambda: ppsci.equation.NavierStokes(0.01, 1.0, 2, False, detach_keys=c.get("detach"))}[c["eq"]]()
    has_w = f"{name}/weight/{keys[0]}" in GOLD.files
    cfg = {"dataset": {"name": "IterableNamedArrayDataset",
                       "input": {k: X[:, j:j + 1] for j, k in enumerate(c["inputs"])},
                       "label": {k: GOLD[f"{name}/label/{k}"][:, None].astype(np.float32) for k in keys},
                       "weight": {k: GOLD[f"{name}/weight/{k}"][:, None].astype(np.float32) for k in keys} if has_w else None}}
    cst = ppsci.constraint.SupervisedConstraint(cfg, ppsci.loss.MSELoss(c["reduction"]), eq.equations, name="EQ")
    solver = ppsci.solver.Solver(model, {"EQ": cst}, str(tmp_path), ppsci.optimizer.Adam(1e-3)(model), epochs=1,
                                 iters_per_epoch=1)
    solver.engine.forward_backward([solver._compiled["EQ"].fused])
    losses = solver._compiled["EQ"].fused.losses()
    for k in keys:
        assert losses[k] == pytest.approx(float(GOLD[f"{name}/loss/{k}"]), rel=5e-5), k
    assert rel(solver.engine.grad.cpu().numpy(), GOLD[f"{name}/grad"]) < 1e-4
    # residual values per point (what BASELINE.json calls "L2 residual vs ref")
    res = solver.predict({k: X[:, j:j + 1] for j, k in enumerate(c["inputs"])}, eq.equations, batch_size=None, return_numpy=True)
    for k in keys:
        assert rel(res[k][:, 0], GOLD[f"{name}/res/{k}"]) < 1e-5, k
