"""ppsci.arch.ModifiedMLP (layer-by-layer HIP path shared with PirateNet: csrc/pirate.hip + the MFMA 1x1-conv GEMMs) against
tests/golden/modified_mlp.npz -- the REFERENCE's own ModifiedMLP (ppsci/arch/mlp.py:318-528) executed in float64 under the
paddle shim (tests/golden/make_modified_mlp_golden.py): with / without Fourier embedding (dim != hidden_size), periods, random
weight factorisation, 1-3 inputs, 1-4 gated layers.  The fp64 oracle restatement is pinned by the same file."""
import os

import numpy as np
import pytest
import torch

import ppsci
from tests.common import make_dev_fixture, rel
from tests.golden.make_modified_mlp_golden import CASES
from tests.golden.make_piratenet_golden import equations

dev = make_dev_fixture()
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "modified_mlp.npz"))


def _state(name):
    return {k.split("/", 2)[2]: GOLD[k] for k in GOLD.files if k.startswith(f"{name}/param/")}


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_is_pinned_by_reference_run(name):
    from oracle import ref_torch as R

    c = CASES[name]
    model = R.ModifiedMLPN(c["inputs"], c["outputs"], _state(name), c["act"], c["periods"])
    X = GOLD[f"{name}/X"]
    keys = [k.split("/")[-1] for k in GOLD.files if k.startswith(f"{name}/res/")]
    cst = dict(name="EQ", input={k: X[:, j:j + 1] for j, k in enumerate(c["inputs"])},
               exprs={k: R.lambdify(e, model) for k, e in equations(c).items()},
               label={k: GOLD[f"{name}/label/{k}"][:, None] for k in keys}, reduction=c["reduction"])
    total, losses, g, outs = R.loss_and_grads(model, [cst])
    for k in keys:
        assert rel(outs[0][k].detach().numpy()[:, 0], GOLD[f"{name}/res/{k}"]) < 1e-9
        assert losses[k] == pytest.approx(float(GOLD[f"{name}/loss/{k}"]), rel=1e-9)
    assert rel(g, np.concatenate([GOLD[f"{name}/grad/{n}"].ravel() for n in _state(name)])) < 1e-8


@pytest.mark.parametrize("name", list(CASES))
def test_modified_mlp_matches_reference_run(name, dev, tmp_path):
    c = CASES[name]
    model = ppsci.arch.ModifiedMLP(c["inputs"], c["outputs"], c["layers"], c["hidden"], c["act"], periods=c["periods"],
                                   fourier=c["fourier"], random_weight=c["rwf"])
    state = _state(name)
    assert [n for n, _ in model.named_parameters()] == list(state)  # the reference's parameters() order and names
    missing, unexpected = model.set_state_dict(state)
    assert not missing and not unexpected
    X = GOLD[f"{name}/X"].astype(np.float32)
    keys = [k.split("/")[-1] for k in GOLD.files if k.startswith(f"{name}/res/")]
    eqs = equations(c)
    inp = {k: X[:, j:j + 1] for j, k in enumerate(c["inputs"])}
    out = model({k: torch.as_tensor(v) for k, v in inp.items()})
    for k in c["outputs"]:
        assert rel(out[k].cpu().numpy()[:, 0], GOLD[f"{name}/out/{k}"]) < 5e-6, k
    cfg = {"dataset": {"name": "IterableNamedArrayDataset", "input": inp,
                       "label": {k: GOLD[f"{name}/label/{k}"][:, None].astype(np.float32) for k in keys}}}
    cst = ppsci.constraint.SupervisedConstraint(cfg, ppsci.loss.MSELoss(c["reduction"]), eqs, name="EQ")
    solver = ppsci.solver.Solver(model, {"EQ": cst}, str(tmp_path), ppsci.optimizer.Adam(1e-3)(model), epochs=2, iters_per_epoch=1)
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
            assert rel(g[off:off + k], ref) < 1e-4, n
        off += k
    assert rel(g, gref) < 1e-4
    res = solver.predict(inp, eqs, batch_size=None, return_numpy=True)
    for k in keys:
        assert rel(res[k][:, 0], GOLD[f"{name}/res/{k}"]) < 1e-5, k
    before = model.flat_params.clone()
    solver.train()
    assert not torch.equal(before, model.flat_params)
