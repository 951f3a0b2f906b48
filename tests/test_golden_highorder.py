"""Third / fourth-order derivatives (Biharmonic dim 1 and 2, KdV-type u_xxx, the boundary quantities of
examples/euler_beam) and the relu / leaky_relu / elu / selu / identity activations against tests/golden/highorder.npz
-- produced by executing the REFERENCE's own code (arch/mlp.py + activation.py, autodiff/ad.py, utils/symbolic.py
DerivativeNode, equation/pde/biharmonic.py, loss/mse.py) in float64 under the torch-backed paddle shim
(tests/golden/make_highorder_golden.py).

1. the CPU oracle (oracle/ref_torch.py, reverse-over-reverse autograd) reproduces them to ~1e-9;
2. the HIP path through the ppsci API matches within fp32 tolerance.  Fourth derivatives amplify the fp32 rounding of
   the streams (cancellation in Faa di Bruno's sums): residual rel-L2 <= 2e-5 for order 4, 1e-5 otherwise; gradient
   rel-L2 <= 2e-4; loss rel <= 1e-4."""
import os

import numpy as np
import pytest
import torch

import ppsci
from oracle import ref_torch as R
from oracle import taylor_np as T
from tests.common import make_dev_fixture, rel
from tests.golden.make_highorder_golden import CASES

dev = make_dev_fixture()
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "highorder.npz"))


def _keys(name):
    return [k.split("/")[-1] for k in GOLD.files if k.startswith(f"{name}/res/")]


def _sympy_eqs(c):
    import sympy as sp

    if c["eq"] in ("biharmonic1", "biharmonic2"):
        return ppsci.equation.Biharmonic(int(c["eq"][-1]), -1.0, 1.0).equations
    syms = sp.symbols(" ".join(c["inputs"]))
    syms = syms if isinstance(syms, tuple) else (syms,)
    u = sp.Function("u")(*syms)
    if c["eq"] == "kdv":
        t, x = syms
        return {"kdv": u.diff(t) + u * u.diff(x) + 0.0025 * u.diff(x, 3)}
    if c["eq"] == "beam_bc":
        (x,) = syms
        return {"u__x": u.diff(x), "u__x__x": u.diff(x, 2), "u__x__x__x": u.diff(x, 3)}
    x, y = syms
    return {"r": u.diff(x) + u * u.diff(y) + 0.5 * u.diff(x, 2) + u.diff(y, 2)}


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_is_pinned_by_reference_run(name):
    c = CASES[name]
    flat = GOLD[f"{name}/params"]
    net = T.make_net(len(c["inputs"]), c["hidden"], len(c["outputs"]), activation=c["act"])
    off = 0
    for i in range(len(net.weights)):
        n = net.weights[i].size
        net.weights[i] = flat[off:off + n].reshape(net.weights[i].shape)
        off += n
        n = net.biases[i].size
        net.biases[i] = flat[off:off + n]
        off += n
    X = GOLD[f"{name}/X"]
    model = R.MLP(c["inputs"], c["outputs"], net)
    keys = _keys(name)
    cst = dict(name="EQ", input={k: X[:, j:j + 1] for j, k in enumerate(c["inputs"])},
               exprs={k: R.lambdify(e, model) for k, e in _sympy_eqs(c).items()},
               label={k: GOLD[f"{name}/label/{k}"][:, None] for k in keys}, reduction=c["reduction"])
    total, losses, g, outs = R.loss_and_grads(model, [cst])
    for k in keys:
        assert rel(outs[0][k].detach().numpy()[:, 0], GOLD[f"{name}/res/{k}"]) < 1e-9
        assert losses[k] == pytest.approx(float(GOLD[f"{name}/loss/{k}"]), rel=1e-9)
    assert rel(g, GOLD[f"{name}/grad"]) < 1e-8


@pytest.mark.parametrize("name", list(CASES))
def test_hip_path_matches_reference_run(name, dev, tmp_path):
    c = CASES[name]
    X = GOLD[f"{name}/X"].astype(np.float32)
    keys = _keys(name)
    model = ppsci.arch.MLP(c["inputs"], c["outputs"], len(c["hidden"]), c["hidden"][0], c["act"])
    model.flat_params.copy_(torch.tensor(GOLD[f"{name}/params"], dtype=torch.float32).to(model.flat_params.device))
    eqs = _sympy_eqs(c)
    inp = {k: X[:, j:j + 1] for j, k in enumerate(c["inputs"])}
    cfg = {"dataset": {"name": "IterableNamedArrayDataset", "input": inp,
                       "label": {k: GOLD[f"{name}/label/{k}"][:, None].astype(np.float32) for k in keys}}}
    cst = ppsci.constraint.SupervisedConstraint(cfg, ppsci.loss.MSELoss(c["reduction"]), eqs, name="EQ")
    solver = ppsci.solver.Solver(model, {"EQ": cst}, str(tmp_path), ppsci.optimizer.Adam(1e-3)(model), epochs=1,
                                 iters_per_epoch=1)
    solver.engine.forward_backward([solver._compiled["EQ"].fused])
    losses = solver._compiled["EQ"].fused.losses()
    fourth = c["eq"].startswith("biharmonic")
    for k in keys:
        assert losses[k] == pytest.approx(float(GOLD[f"{name}/loss/{k}"]), rel=1e-4), k
    assert rel(solver.engine.grad.cpu().numpy(), GOLD[f"{name}/grad"]) < 1e-4
    res = solver.predict(inp, eqs, batch_size=None, return_numpy=True)
    for k in keys:
        assert rel(res[k][:, 0], GOLD[f"{name}/res/{k}"]) < (2e-5 if fourth else 1e-5), k
