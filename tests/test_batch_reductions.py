"""Batch reductions inside user expressions (`out["u"].mean()`, `.sum()`: the reference runs arbitrary tensor code,
/root/reference/ppsci/utils/expression.py:96-102) lowered as a three-launch residual program (graph.lower,
engine.FusedConstraint._forward_reductions): values, loss and parameter gradient against torch's reverse-over-reverse autograd in
float64 on the same weights."""
import numpy as np
import pytest
import torch

import ppsci
from ppsci.autodiff import hessian, jacobian
from tests.common import make_dev_fixture, rel

dev = make_dev_fixture()


def _ref_forward(model, inp):
    ps = [p.detach().double().cpu() for p in model.parameters()]
    for p in ps:
        p.requires_grad_(True)
    X = {k: torch.tensor(v.astype(np.float64), requires_grad=True) for k, v in inp.items()}
    h = torch.cat([X[k] for k in model.input_keys], 1)
    for i in range(0, len(ps) - 2, 2):
        h = torch.tanh(h @ ps[i] + ps[i + 1])
    return ps, X, h @ ps[-2] + ps[-1]


def _g(f, v):
    return torch.autograd.grad(f.sum(), v, create_graph=True)[0]


def _solver(tmp_path, model, exprs, inp, reduction):
    n = len(next(iter(inp.values())))
    cfg = {"dataset": {"name": "NamedArrayDataset", "input": inp, "label": {k: np.zeros((n, 1), np.float32) for k in exprs}},
           "batch_size": n, "sampler": {"name": "BatchSampler", "shuffle": False, "drop_last": True}}
    cst = ppsci.constraint.SupervisedConstraint(cfg, ppsci.loss.MSELoss(reduction), exprs, name="EQ")
    opt = ppsci.optimizer.Adam(1e-3)(model)
    return ppsci.solver.Solver(model, {"EQ": cst}, str(tmp_path), opt, epochs=1, iters_per_epoch=1)


@pytest.mark.parametrize("reduction", ["mean", "sum"])
def test_mean_and_sum_inside_expressions(dev, tmp_path, reduction):
    torch.manual_seed(5)
    np.random.seed(5)
    n = 52 if dev != "gpu" else 5000  # (a ragged last tile / block)
    model = ppsci.arch.MLP(("x", "y"), ("u", "v"), 2, 24, "tanh")
    rng = np.random.default_rng(9)
    inp = {"x": rng.uniform(-1, 1, (n, 1)).astype(np.float32), "y": rng.uniform(-1, 1, (n, 1)).astype(np.float32)}
    exprs = {
        # a field measured against its own batch mean, scaled by a batch sum of a derivative quantity
        "centred": lambda d: (d["u"] - d["u"].mean()) * (1.0 + (jacobian(d["v"], d["x"]) * d["y"]).sum() / 100.0),
        # a PDE residual normalised by the mean square of the field
        "normed": lambda d: (hessian(d["u"], d["x"]) + hessian(d["u"], d["y"])) / ((d["u"] * d["u"]).mean() + 0.5),
    }
    solver = _solver(tmp_path, model, exprs, inp, reduction)
    cc = solver._compiled["EQ"]
    assert cc.low.reductions["k"] == 3 and not cc.fused.one_launch_ready()
    solver.engine.forward_backward([cc.fused])
    losses = cc.fused.losses()
    grad = solver.engine.grad.detach().cpu().numpy().astype(np.float64)

    ps, X, out = _ref_forward(model, inp)
    u, v = out[:, :1], out[:, 1:]
    ref = {"centred": (u - u.mean()) * (1.0 + (_g(v, X["x"]) * X["y"]).sum() / 100.0),
           "normed": (_g(_g(u, X["x"]), X["x"]) + _g(_g(u, X["y"]), X["y"])) / ((u * u).mean() + 0.5)}
    red = (lambda t: t.mean()) if reduction == "mean" else (lambda t: t.sum())
    terms = {k: red(r * r) for k, r in ref.items()}
    total = sum(terms.values())
    gref = np.concatenate([g.detach().numpy().ravel() for g in torch.autograd.grad(total, ps, retain_graph=True)])
    for k in exprs:
        assert abs(losses[k] / float(terms[k].detach()) - 1.0) < 2e-5, (k, losses[k], float(terms[k].detach()))
    assert rel(grad, gref) < 5e-5, rel(grad, gref)
    # without the third launch (the reductions' adjoints) the gradient is visibly different: the test can tell
    partial = np.concatenate([g.detach().numpy().ravel() for g in torch.autograd.grad(
        sum(red(((u - u.mean().detach()) * (1.0 + (_g(v, X["x"]) * X["y"]).sum().detach() / 100.0)) ** 2
                if k == "centred" else (ref["normed"].detach() * 0 + (_g(_g(u, X["x"]), X["x"]) + _g(_g(u, X["y"]), X["y"]))
                                        / ((u * u).mean().detach() + 0.5)) ** 2) for k in exprs), ps)])
    assert rel(partial, gref) > 1e-3
    # values through predict (eval mode: two launches, no adjoints)
    got = solver.predict(inp, exprs, batch_size=None, return_numpy=True)
    for k in exprs:
        assert rel(got[k][:, 0], ref[k].detach().numpy()[:, 0]) < 2e-5, k


def test_training_with_a_reduction_decreases_the_loss(dev, tmp_path):
    torch.manual_seed(1)
    n = 64
    model = ppsci.arch.MLP(("x",), ("u",), 2, 16, "tanh")
    inp = {"x": np.linspace(-1, 1, n, dtype=np.float32).reshape(n, 1)}
    exprs = {"fit": lambda d: d["u"] - d["u"].mean() - d["x"]}  # u = x + const: the mean removes the constant
    cfg = {"dataset": {"name": "NamedArrayDataset", "input": inp, "label": {"fit": np.zeros((n, 1), np.float32)}},
           "batch_size": n, "sampler": {"name": "BatchSampler", "shuffle": False, "drop_last": True}}
    cst = ppsci.constraint.SupervisedConstraint(cfg, ppsci.loss.MSELoss("mean"), exprs, name="EQ")
    opt = ppsci.optimizer.Adam(5e-3)(model)
    solver = ppsci.solver.Solver(model, {"EQ": cst}, str(tmp_path), opt, epochs=30, iters_per_epoch=1, log_freq=1000)
    cc = solver._compiled["EQ"]
    solver.engine.forward_backward([cc.fused])
    first = cc.fused.losses()["fit"]
    solver.train()
    solver.engine.forward_backward([cc.fused])
    assert cc.fused.losses()["fit"] < 0.5 * first


def test_what_is_refused_says_why(dev):
    from paddlescience_amd import compile as cp
    from paddlescience_amd import graph

    model = ppsci.arch.MLP(("x",), ("u",), 2, 16, "tanh")

    def lower(exprs):
        outs = cp.trace_exprs(model, ("x",), exprs, (), None, [])
        return graph.lower(outs, [dict(key=k, label=None, weight=None, area=None, scale=1.0) for k in exprs], n_global=10)

    with pytest.raises(NotImplementedError, match="inside a batch reduction"):
        lower({"var": lambda d: ((d["u"] - d["u"].mean()) ** 2).mean()})
    with pytest.raises(NotImplementedError, match="differentiate first"):
        lower({"bad": lambda d: jacobian(d["u"].mean() * d["x"], d["x"])})
    low = lower({"ok": lambda d: d["u"] * d["x"].mean()})  # (a reduction of an input column is still a reduction: 1 slot)
    assert low.reductions["k"] == 1


def test_volterra_coupled_residual_matches_autograd(dev, tmp_path):
    """ppsci.equation.Volterra (volterra.py:66-77: lhs[:N] - int_mat @ u on the batch [points | quadrature points]) through the
    kernels: loss and parameter gradient against float64 autograd of the same expression."""
    torch.manual_seed(2)
    N, Q = 6, 5
    model = ppsci.arch.MLP(("x",), ("u",), 2, 16, "tanh")
    eq = ppsci.equation.Volterra(0.0, N, Q, lambda x, s: np.exp(s - x), lambda out: jacobian(out["u"], out["x"]) + out["u"])
    x = np.linspace(0.3, 2.0, N, dtype=np.float32).reshape(-1, 1)
    xq = np.concatenate([x, eq.get_quad_points(x).reshape(-1, 1)], axis=0).astype(np.float32)
    lab = np.linspace(-0.2, 0.3, N, dtype=np.float32).reshape(-1, 1)
    cfg = {"dataset": {"name": "IterableNamedArrayDataset", "input": {"x": xq}, "label": {"volterra": lab}},
           "batch_size": len(xq), "iters_per_epoch": 1}
    cst = ppsci.constraint.SupervisedConstraint(cfg, ppsci.loss.MSELoss("mean"), eq.equations, name="EQ")
    opt = ppsci.optimizer.Adam(1e-3)(model)
    solver = ppsci.solver.Solver(model, {"EQ": cst}, str(tmp_path), opt, epochs=1, iters_per_epoch=1)
    cc = solver._compiled["EQ"]
    assert cc.low.couplings and cc.specialised_to  # (the matrix was built from the values of this batch)
    solver.engine.forward_backward([cc.fused])
    loss = cc.fused.losses()["volterra"]
    grad = solver.engine.grad.detach().cpu().numpy().astype(np.float64)

    ps, X, u = _ref_forward(model, {"x": xq})
    lhs = _g(u, X["x"]) + u
    M = torch.tensor(eq._get_int_matrix(xq).astype(np.float64))
    r = lhs[:N] - M @ u - torch.tensor(lab.astype(np.float64))
    total = (r * r).mean()
    gref = np.concatenate([g.detach().numpy().ravel() for g in torch.autograd.grad(total, ps)])
    assert abs(loss / float(total.detach()) - 1.0) < 2e-5, (loss, float(total.detach()))
    assert rel(grad, gref) < 5e-5, rel(grad, gref)
    # a few optimizer steps move the loss down (the whole step runs: Taylor forward, five small launches, reverse, Adam)
    solver.epochs = 25
    solver.train()
    solver.engine.forward_backward([cc.fused])
    assert cc.fused.losses()["volterra"] < loss
