"""API-level parity on the structure of /root/reference/examples/laplace/laplace2d.py (BASELINE config 1,
scaled down): ppsci.arch.MLP + ppsci.equation.Laplace + Interior/Boundary constraints + MSELoss("sum") +
Adam, through ppsci.solver.Solver.  Checked against the reverse-over-reverse oracle (oracle/ref_torch.py)
evaluated in fp64 on the same fp32 weights and points:
  * the per-constraint losses at step 0,
  * the parameters after 3 Adam steps (covers the flat gradient, i.e. taylor_bwd of both constraints),
  * solver.predict and solver.eval (MSE metric)."""
import numpy as np
import pytest
import torch

import ppsci
from oracle import ref_torch as R
from oracle import taylor_np as T
from tests.common import make_dev_fixture, rel, set_model_weights

dev = make_dev_fixture()


def u_solution_func(out):
    x, y = out["x"], out["y"]
    return np.cos(x) * np.cosh(y)


def build(tmp_path, n_int=49, n_bc=16, steps=3):
    np.random.seed(2024)
    model = ppsci.arch.MLP(("x", "y"), ("u",), 3, 20, "tanh")
    net = T.make_net(2, [20, 20, 20], 1, seed=1234, bias_scale=0.05)
    set_model_weights(model, net)
    equation = {"laplace": ppsci.equation.Laplace(dim=2)}
    geom = {"rect": ppsci.geometry.Rectangle((0.0, 0.0), (1.0, 1.0))}
    cfg = {"dataset": "IterableNamedArrayDataset", "iters_per_epoch": 1}
    pde = ppsci.constraint.InteriorConstraint(equation["laplace"].equations, {"laplace": 0}, geom["rect"],
                                              {**cfg, "batch_size": n_int}, ppsci.loss.MSELoss("sum"), evenly=True, name="EQ")
    bc = ppsci.constraint.BoundaryConstraint({"u": lambda out: out["u"]}, {"u": u_solution_func}, geom["rect"],
                                             {**cfg, "batch_size": n_bc}, ppsci.loss.MSELoss("sum"), name="BC")
    constraint = {pde.name: pde, bc.name: bc}
    optimizer = ppsci.optimizer.Adam(learning_rate=1e-3)(model)
    val = ppsci.validate.GeometryValidator({"u": lambda out: out["u"]}, {"u": u_solution_func}, geom["rect"],
                                           {"dataset": "IterableNamedArrayDataset", "total_size": n_int},
                                           ppsci.loss.MSELoss(), evenly=True, metric={"MSE": ppsci.metric.MSE()},
                                           with_initial=True, name="MSE_Metric")
    solver = ppsci.solver.Solver(model, constraint, str(tmp_path), optimizer, epochs=steps, iters_per_epoch=1,
                                 equation=equation, geom=geom, validator={val.name: val}, log_freq=1)
    return solver, model, net, pde, bc


def oracle_constraints(omodel, pde, bc):
    ds_p, ds_b = pde.data_loader, bc.data_loader
    lap = R.lambdify(R.laplace_exprs(2)["laplace"], omodel)
    c1 = dict(name="EQ", input={k: ds_p.input[k].astype(np.float64) for k in ("x", "y")}, exprs={"laplace": lap},
              label={"laplace": ds_p.label["laplace"].astype(np.float64)}, reduction="sum")
    c2 = dict(name="BC", input={k: ds_b.input[k].astype(np.float64) for k in ("x", "y")}, exprs={},
              label={"u": ds_b.label["u"].astype(np.float64)}, reduction="sum")
    return [c1, c2]


def test_losses_and_three_adam_steps_match_oracle(tmp_path):
    solver, model, net, pde, bc = build(tmp_path)
    net32 = net.astype(np.float32).astype(np.float64)
    omodel = R.MLP(("x", "y"), ("u",), net32)
    csts = oracle_constraints(omodel, pde, bc)
    total0, losses0, g0, _ = R.loss_and_grads(omodel, csts)

    # --- step-0 losses through the engine, without updating
    solver.engine.forward_backward([c.fused for c in solver._compiled.values()])
    l_eq = solver._compiled["EQ"].fused.losses()["laplace"]
    l_bc = solver._compiled["BC"].fused.losses()["u"]
    assert l_eq == pytest.approx(losses0["laplace"], rel=2e-5)
    assert l_bc == pytest.approx(losses0["u"], rel=2e-5)
    assert rel(solver.engine.grad.cpu().numpy(), g0) < 2e-5

    # --- three optimizer steps
    solver.train()
    p = T.flat_params(net32)
    adam = R.Adam(p.size, 1e-3)
    for _ in range(3):
        off = 0
        with torch.no_grad():
            for t in omodel.parameters():
                n = t.numel()
                t.copy_(torch.tensor(p[off:off + n].reshape(t.shape)))
                off += n
        _, _, g, _ = R.loss_and_grads(omodel, csts)
        p = adam.step(p, g)
    got = model.flat_params.cpu().numpy().astype(np.float64)
    # Adam's first steps move every parameter by ~lr regardless of gradient scale: compare the updates
    upd_ref = p - T.flat_params(net32)
    upd_got = got - T.flat_params(net32)
    assert rel(upd_got, upd_ref) < 2e-3
    assert np.abs(got - p).max() < 5e-6


def test_predict_and_eval(tmp_path):
    solver, model, net, pde, bc = build(tmp_path)
    net32 = net.astype(np.float32).astype(np.float64)
    X = np.random.default_rng(0).uniform(0, 1, (37, 2)).astype(np.float32)
    pred = solver.predict({"x": X[:, :1], "y": X[:, 1:]}, batch_size=16, return_numpy=True)
    ref = T.taylor_forward(net32, X.astype(np.float64), np.zeros((0, 2)), 0)[0, 0]
    assert rel(pred["u"][:, 0], ref) < 2e-6
    # derivative expressions through predict (expr_dict with jacobian), like visu/eval expressions
    from ppsci.autodiff import jacobian

    pred2 = solver.predict({"x": X[:, :1], "y": X[:, 1:]}, {"u__x": lambda out: jacobian(out["u"], out["x"])},
                           batch_size=64, return_numpy=True)
    ux = T.taylor_forward(net32, X.astype(np.float64), np.array([[1.0, 0.0]]), 0)[0, 1]
    assert rel(pred2["u__x"][:, 0], ux) < 5e-6
    metric, group = solver.eval()
    val = solver.validator["MSE_Metric"]
    xs = val.data_loader.input
    Xv = np.concatenate([xs["x"], xs["y"]], 1).astype(np.float64)
    uv = T.taylor_forward(net32, Xv, np.zeros((0, 2)), 0)[0, 0]
    mse = np.mean((uv - val.data_loader.label["u"][:, 0].astype(np.float64)) ** 2)
    assert group["MSE_Metric"]["MSE.u"] == pytest.approx(mse, rel=1e-4)
    assert metric == pytest.approx(mse, rel=1e-4)


def test_update_freq_accumulates_gradients(tmp_path):
    """train.py:141-142, :163-180: losses divided by update_freq, gradients accumulated, optimizer steps every
    update_freq-th iteration and at the last iteration of the epoch (5 iterations, update_freq 2: steps after
    iterations 2, 4 and 5 with accumulated gradients g, g and g/2 of the then-current parameters)."""
    np.random.seed(2024)
    model = ppsci.arch.MLP(("x", "y"), ("u",), 3, 20, "tanh")
    net = T.make_net(2, [20, 20, 20], 1, seed=1234, bias_scale=0.05)
    set_model_weights(model, net)
    equation = {"laplace": ppsci.equation.Laplace(dim=2)}
    geom = {"rect": ppsci.geometry.Rectangle((0.0, 0.0), (1.0, 1.0))}
    cfg = {"dataset": "IterableNamedArrayDataset", "iters_per_epoch": 5}
    pde = ppsci.constraint.InteriorConstraint(equation["laplace"].equations, {"laplace": 0}, geom["rect"],
                                              {**cfg, "batch_size": 49}, ppsci.loss.MSELoss("sum"), evenly=True, name="EQ")
    bc = ppsci.constraint.BoundaryConstraint({"u": lambda out: out["u"]}, {"u": u_solution_func}, geom["rect"],
                                             {**cfg, "batch_size": 16}, ppsci.loss.MSELoss("sum"), name="BC")
    optimizer = ppsci.optimizer.Adam(learning_rate=1e-3)(model)
    solver = ppsci.solver.Solver(model, {pde.name: pde, bc.name: bc}, str(tmp_path), optimizer, epochs=1,
                                 iters_per_epoch=5, update_freq=2, equation=equation, geom=geom, log_freq=1)
    solver.train()
    net32 = net.astype(np.float32).astype(np.float64)
    omodel = R.MLP(("x", "y"), ("u",), net32)
    csts = oracle_constraints(omodel, pde, bc)
    p = T.flat_params(net32)
    adam = R.Adam(p.size, 1e-3)
    for n_acc in (2, 2, 1):
        off = 0
        with torch.no_grad():
            for t in omodel.parameters():
                n = t.numel()
                t.copy_(torch.tensor(p[off:off + n].reshape(t.shape)))
                off += n
        _, _, g, _ = R.loss_and_grads(omodel, csts)
        p = adam.step(p, g * n_acc / 2.0)  # same batch every iteration: n_acc identical gradients, each / 2
    got = model.flat_params.cpu().numpy().astype(np.float64)
    assert np.abs(got - p).max() < 5e-6
    assert optimizer.t == 3


def test_expression_solver_surface(tmp_path):
    """ppsci.utils.expression.ExpressionSolver (expression.py:60-222): train_forward / eval_forward / visu_forward with
    the reference's arguments and return structure, against the oracle's restatement of the same loop."""
    from ppsci.utils.expression import ExpressionSolver

    net = T.make_net(2, [20, 20, 20], 1, bias_scale=0.1)
    model = ppsci.arch.MLP(("x", "y"), ("u",), 3, 20, "tanh")
    set_model_weights(model, net)
    eq = ppsci.equation.Laplace(2)
    rng = np.random.default_rng(3)
    X = rng.uniform(0, 1, (40, 2)).astype(np.float32)
    Xb = rng.uniform(0, 1, (12, 2)).astype(np.float32)
    lab_b = rng.standard_normal((12, 1)).astype(np.float32)
    inputs = ({"x": X[:, :1], "y": X[:, 1:]}, {"x": Xb[:, :1], "y": Xb[:, 1:]})
    labels = ({"laplace": np.zeros((40, 1), np.float32)}, {"u": lab_b})
    weights = ({}, {})
    exprs = (eq.equations, {"u": lambda out: out["u"]})

    class C:
        def __init__(self, loss):
            self.loss = loss

    csts = {"EQ": C(ppsci.loss.MSELoss("sum")), "BC": C(ppsci.loss.MSELoss("mean"))}
    es = ExpressionSolver()
    with pytest.raises(NotImplementedError):
        es.forward()
    losses_all, losses_cst = es.train_forward(exprs, inputs, model, csts, labels, weights)
    grad = es.backward()
    omodel = R.MLP(("x", "y"), ("u",), net.astype(np.float32).astype(np.float64))
    ocs = [dict(name="EQ", input={k: v.astype(np.float64) for k, v in inputs[0].items()},
                exprs={k: R.lambdify(e, omodel) for k, e in R.laplace_exprs(2).items()},
                label={"laplace": np.zeros((40, 1))}, reduction="sum"),
           dict(name="BC", input={k: v.astype(np.float64) for k, v in inputs[1].items()}, exprs={"u": lambda d: d["u"]},
                label={"u": lab_b.astype(np.float64)}, reduction="mean")]
    total, olosses, og, outs = R.loss_and_grads(omodel, ocs)
    assert set(losses_all) == {"laplace", "u"} and set(losses_cst) == {"EQ", "BC"}
    assert float(losses_all["laplace"]) == pytest.approx(olosses["laplace"], rel=3e-5)
    assert float(losses_all["u"]) == pytest.approx(olosses["u"], rel=3e-5)
    assert losses_cst["EQ"] == pytest.approx(olosses["laplace"], rel=3e-5)
    assert rel(grad.cpu().numpy(), og) < 5e-5
    # second call, same signature: the compiled constraint is reused with the new batch
    inputs2 = ({"x": X[::-1, :1].copy(), "y": X[::-1, 1:].copy()}, inputs[1])
    l2, _ = es.train_forward(exprs, inputs2, model, csts, labels, weights)
    assert float(l2["laplace"]) == pytest.approx(float(losses_all["laplace"]), rel=1e-5) and len(es._cache) == 2
    out, vloss = es.eval_forward(exprs[1], inputs[1], model, C(ppsci.loss.MSELoss("mean")), labels[1], {})
    assert rel(out["u"].cpu().numpy(), outs[1]["u"].detach().numpy()) < 5e-6
    assert float(vloss["u"]) == pytest.approx(olosses["u"], rel=3e-5)
    vis = es.visu_forward(eq.equations, inputs[0], model)
    assert rel(vis["laplace"].cpu().numpy(), outs[0]["laplace"].detach().numpy()) < 1e-5
    assert set(es.visu_forward(None, inputs[0], model)) == {"u"}
