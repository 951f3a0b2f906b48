"""Row a17 beyond MSELoss: L1Loss / MAELoss / L2Loss / L2RelLoss as epilogue loss kinds, through the ppsci API on a
Laplace residual + a data term with per-point weights, against the oracle's restatement of
/root/reference/ppsci/loss/{l1,mae,l2}.py (fp64 autograd)."""
import numpy as np
import pytest
import torch

import ppsci
from oracle import ref_torch as R
from oracle import taylor_np as T
from tests.common import make_dev_fixture, rel, set_model_weights

dev = make_dev_fixture()

KINDS = {"l1": ppsci.loss.L1Loss, "mae": ppsci.loss.MAELoss, "l2": ppsci.loss.L2Loss, "l2rel": ppsci.loss.L2RelLoss}


@pytest.mark.parametrize("kind", list(KINDS))
@pytest.mark.parametrize("reduction", ["mean", "sum"])
def test_point_losses_match_oracle(kind, reduction, dev, tmp_path):
    model = ppsci.arch.MLP(("x", "y"), ("u",), 2, 16, "tanh")
    net = T.make_net(2, [16, 16], 1, bias_scale=0.1)
    set_model_weights(model, net)
    N = 37
    rng = np.random.default_rng(9)
    X = rng.uniform(0, 1, (N, 2)).astype(np.float32)
    lab = {"laplace": rng.standard_normal((N, 1)).astype(np.float32) + 3.0,  # away from 0 for the relative loss
           "u": (np.cos(X[:, :1]) * np.cosh(X[:, 1:])).astype(np.float32)}
    wts = {"laplace": rng.uniform(0.5, 2.0, (N, 1)).astype(np.float32), "u": rng.uniform(0.5, 2.0, (N, 1)).astype(np.float32)}
    eq = ppsci.equation.Laplace(dim=2)
    exprs = {**eq.equations, "u": lambda out: out["u"]}
    cfg = {"dataset": {"name": "IterableNamedArrayDataset", "input": {"x": X[:, :1], "y": X[:, 1:]}, "label": lab,
                       "weight": wts}}
    loss = KINDS[kind](reduction, weight={"u": 0.7})
    cst = ppsci.constraint.SupervisedConstraint(cfg, loss, exprs, name="EQ")
    opt = ppsci.optimizer.Adam(1e-3)(model)
    solver = ppsci.solver.Solver(model, {"EQ": cst}, str(tmp_path), opt, epochs=1, iters_per_epoch=1)
    solver.engine.forward_backward([c.fused for c in solver._compiled.values()])
    g = solver.engine.grad.cpu().numpy().astype(np.float64)
    got = solver._compiled["EQ"].fused.losses()
    omodel = R.MLP(("x", "y"), ("u",), net.astype(np.float32).astype(np.float64))
    oc = dict(name="EQ", input={"x": X[:, :1].astype(np.float64), "y": X[:, 1:].astype(np.float64)},
              exprs={"laplace": R.laplace_fn(2) if hasattr(R, "laplace_fn") else None, "u": lambda d: d["u"]},
              label={k: v.astype(np.float64) for k, v in lab.items()}, weight={k: v.astype(np.float64) for k, v in wts.items()},
              reduction=reduction, loss_weight={"u": 0.7}, loss_kind=kind)
    if oc["exprs"]["laplace"] is None:
        def lap(d):
            ux = torch.autograd.grad(d["u"].sum(), d["x"], create_graph=True)[0]
            uy = torch.autograd.grad(d["u"].sum(), d["y"], create_graph=True)[0]
            return torch.autograd.grad(ux.sum(), d["x"], create_graph=True)[0] + torch.autograd.grad(uy.sum(), d["y"], create_graph=True)[0]
        oc["exprs"]["laplace"] = lap
    total, losses, gref, _ = R.loss_and_grads(omodel, [oc])
    for k in ("laplace", "u"):
        assert got[k] == pytest.approx(float(losses[k]), rel=3e-5)
    assert rel(g, gref) < 5e-5
    # the host-side torch forward of the loss class is the same arithmetic
    out = {k: torch.tensor(lab[k]) * 0.9 + 0.05 for k in lab}
    host = loss(out, {k: torch.tensor(v) for k, v in lab.items()}, {k: torch.tensor(v) for k, v in wts.items()})
    ref = R.point_loss(kind, out, {k: torch.tensor(v) for k, v in lab.items()}, {k: torch.tensor(v) for k, v in wts.items()},
                       reduction, {"u": 0.7})
    for k in host:
        assert float(host[k]) == pytest.approx(float(ref[k]), rel=1e-6)


def test_causal_mse_known_answer():
    """The doctest value of CausalMSELoss (mse.py:125-134)."""
    out = {"u": torch.tensor([[0.5, 0.9, 1.0], [1.1, -1.3, 0.0]])}
    lab = {"u": torch.tensor([[-1.8, 1.0, -0.1], [-0.2, 2.5, 2.0]])}
    for f in (ppsci.loss.CausalMSELoss(n_chunks=3), lambda o, l: R.causal_mse_loss(o, l, n_chunks=3)):
        assert float(f(out, lab)["u"]) == pytest.approx(0.96841478, rel=1e-6)


@pytest.mark.parametrize("reduction", ["mean", "sum"])
def test_causal_mse_on_the_fused_path(reduction, dev, tmp_path):
    """CausalMSELoss (mse.py:109-189) on an Allen-Cahn residual + a data term: the windows' weights are constants
    of the reverse sweep; loss terms and parameter gradient against the oracle."""
    model = ppsci.arch.MLP(("t", "x"), ("u",), 2, 16, "tanh")
    net = T.make_net(2, [16, 16], 1, bias_scale=0.1)
    set_model_weights(model, net)
    n_chunks, per = 8, 6
    N = n_chunks * per
    rng = np.random.default_rng(31)
    t = np.repeat(np.linspace(0.0, 1.0, n_chunks), per)[:, None].astype(np.float32)  # time-major batch
    x = rng.uniform(-1, 1, (N, 1)).astype(np.float32)
    lab = {"allen_cahn": np.zeros((N, 1), np.float32), "u": rng.standard_normal((N, 1)).astype(np.float32)}
    wts = {"u": rng.uniform(0.5, 2.0, (N, 1)).astype(np.float32)}
    eq = ppsci.equation.AllenCahn(eps=0.05)
    exprs = {**eq.equations, "u": lambda out: out["u"]}
    cfg = {"dataset": {"name": "IterableNamedArrayDataset", "input": {"t": t, "x": x}, "label": lab, "weight": wts}}
    loss = ppsci.loss.CausalMSELoss(n_chunks, reduction, weight={"u": 0.7}, tol=1.5)
    cst = ppsci.constraint.SupervisedConstraint(cfg, loss, exprs, name="EQ")
    opt = ppsci.optimizer.Adam(1e-3)(model)
    solver = ppsci.solver.Solver(model, {"EQ": cst}, str(tmp_path), opt, epochs=1, iters_per_epoch=1)
    for _ in range(2):  # twice: the factors are recomputed every step
        solver.engine.forward_backward([c.fused for c in solver._compiled.values()])
    g = solver.engine.grad.cpu().numpy().astype(np.float64)
    got = solver._compiled["EQ"].fused.losses()
    omodel = R.MLP(("t", "x"), ("u",), net.astype(np.float32).astype(np.float64))
    oc = dict(name="EQ", input={"t": t.astype(np.float64), "x": x.astype(np.float64)},
              exprs={"allen_cahn": R.allen_cahn_fn(0.05), "u": lambda d: d["u"]},
              label={k: v.astype(np.float64) for k, v in lab.items()}, weight={k: v.astype(np.float64) for k, v in wts.items()},
              reduction=reduction, loss_weight={"u": 0.7}, loss_kind="causal_mse", n_chunks=n_chunks, tol=1.5)
    total, losses, gref, _ = R.loss_and_grads(omodel, [oc])
    for k in ("allen_cahn", "u"):
        assert got[k] == pytest.approx(float(losses[k]), rel=3e-5)
    assert rel(g, gref) < 5e-5
    with pytest.raises(ValueError):
        bad = {"dataset": {"name": "IterableNamedArrayDataset", "input": {"t": t[:-1], "x": x[:-1]},
                           "label": {k: v[:-1] for k, v in lab.items()}}}
        c2 = ppsci.constraint.SupervisedConstraint(bad, ppsci.loss.CausalMSELoss(n_chunks), exprs, name="EQ")
        ppsci.solver.Solver(model, {"EQ": c2}, str(tmp_path), opt, epochs=1, iters_per_epoch=1)
