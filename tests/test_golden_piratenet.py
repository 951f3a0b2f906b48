"""ppsci.arch.PirateNet (layer-by-layer HIP path: csrc/pirate.hip + the MFMA 1x1-conv GEMMs) against tests/golden/piratenet.npz
-- produced by executing the REFERENCE's own PirateNet / PeriodEmbedding / FourierEmbedding / RandomWeightFactorization
(ppsci/arch/mlp.py:28-137, :530-820), autodiff/ad.py, utils/symbolic.py and loss/mse.py in float64 under the torch-backed
paddle shim (tests/golden/make_piratenet_golden.py): network outputs, per-point residuals (u_t, u_xx, products of two
outputs ...), loss terms and the gradient w.r.t. every named parameter incl. alpha, the Fourier kernel and the factorised
weights.  Tolerances: fp32 kernels against fp64 reference values -- residual rel-L2 <= 2e-5, gradient rel-L2 <= 2e-4."""
import os

import numpy as np
import pytest
import torch

import ppsci
from tests.common import make_dev_fixture, rel
from tests.golden.make_piratenet_golden import CASES, equations

dev = make_dev_fixture()
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "piratenet.npz"))


def _model(name):
    c = CASES[name]
    model = ppsci.arch.PirateNet(c["inputs"], c["outputs"], c["blocks"], c["hidden"], c["act"], periods=c["periods"],
                                 fourier=c["fourier"], random_weight=c["rwf"])
    state = {k.split("/", 2)[2]: GOLD[k] for k in GOLD.files if k.startswith(f"{name}/param/")}
    assert [n for n, _ in model.named_parameters()] == list(state)  # the reference's parameters() order and names
    missing, unexpected = model.set_state_dict(state)
    assert not missing and not unexpected
    return c, model


@pytest.mark.parametrize("name", list(CASES))
def test_piratenet_matches_reference_run(name, dev, tmp_path):
    c, model = _model(name)
    X = GOLD[f"{name}/X"].astype(np.float32)
    keys = [k.split("/")[-1] for k in GOLD.files if k.startswith(f"{name}/res/")]
    eqs = equations(c)
    inp = {k: X[:, j:j + 1] for j, k in enumerate(c["inputs"])}
    # eager model call (no derivative streams)
    out = model({k: torch.as_tensor(v) for k, v in inp.items()})
    for k in c["outputs"]:
        assert rel(out[k].cpu().numpy()[:, 0], GOLD[f"{name}/out/{k}"]) < 5e-6, k
    cfg = {"dataset": {"name": "IterableNamedArrayDataset", "input": inp,
                       "label": {k: GOLD[f"{name}/label/{k}"][:, None].astype(np.float32) for k in keys}}}
    cst = ppsci.constraint.SupervisedConstraint(cfg, ppsci.loss.MSELoss(c["reduction"]), eqs, name="EQ")
    solver = ppsci.solver.Solver(model, {"EQ": cst}, str(tmp_path), ppsci.optimizer.Adam(1e-3)(model), epochs=1,
                                 iters_per_epoch=1)
    solver.engine.forward_backward([solver._compiled["EQ"].fused])
    losses = solver._compiled["EQ"].fused.losses()
    for k in keys:
        assert losses[k] == pytest.approx(float(GOLD[f"{name}/loss/{k}"]), rel=1e-4), k
    g = solver.engine.grad.cpu().numpy()
    gref = np.concatenate([GOLD[f"{name}/grad/{n}"].ravel() for n, _ in model.named_parameters()])
    off = 0
    for n, p in model.named_parameters():
        k = p.numel()
        ref = GOLD[f"{name}/grad/{n}"].ravel()
        if np.linalg.norm(ref) > 1e-6 * np.linalg.norm(gref):
            assert rel(g[off:off + k], ref) < 5e-4, n
        off += k
    assert rel(g, gref) < 2e-4
    res = solver.predict(inp, eqs, batch_size=None, return_numpy=True)
    for k in keys:
        assert rel(res[k][:, 0], GOLD[f"{name}/res/{k}"]) < 2e-5, k


def test_piratenet_trains(dev, tmp_path):
    c, model = _model("three_blocks_silu")
    rng = np.random.default_rng(0)
    X = rng.uniform(c["lo"], c["hi"], (64, 3)).astype(np.float32)
    inp = {k: X[:, j:j + 1] for j, k in enumerate(c["inputs"])}
    cfg = {"dataset": {"name": "IterableNamedArrayDataset", "input": inp, "label": {"heat": np.zeros((64, 1), np.float32)}}}
    cst = ppsci.constraint.SupervisedConstraint(cfg, ppsci.loss.MSELoss("mean"), equations(c), name="EQ")
    solver = ppsci.solver.Solver(model, {"EQ": cst}, str(tmp_path), ppsci.optimizer.Adam(2e-3)(model), epochs=1,
                                 iters_per_epoch=30, log_freq=30)
    solver.engine.forward_backward([solver._compiled["EQ"].fused])
    l0 = solver._compiled["EQ"].fused.losses()["heat"]
    solver.train()
    assert solver.last_losses["loss"] < 0.5 * l0
