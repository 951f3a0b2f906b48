"""API-level parity for the other equation classes of the hot path:
AllenCahn (Python closure calling jacobian, period embedding), NavierStokes 2-D steady (sympy, three
outputs, per-point weights, detach_keys) and Poisson; plus symbolic-differentiation paths
(output transform, mixed second derivative by polarisation).  Oracle: oracle/ref_torch.py in fp64."""
import numpy as np
import pytest
import sympy as sp
import torch

import ppsci
from oracle import ref_torch as R
from oracle import taylor_np as T
from ppsci.autodiff import hessian, jacobian
from tests.common import make_dev_fixture, rel, set_model_weights

dev = make_dev_fixture()


def _solver(tmp_path, model, constraint, lr=1e-3):
    opt = ppsci.optimizer.Adam(learning_rate=lr)(model)
    return ppsci.solver.Solver(model, constraint, str(tmp_path), opt, epochs=1, iters_per_epoch=1, log_freq=1)


def _sup_constraint(inputs, labels, exprs, loss, weights=None, name="EQ"):
    cfg = {"dataset": {"name": "IterableNamedArrayDataset", "input": inputs, "label": labels, "weight": weights}}
    return ppsci.constraint.SupervisedConstraint(cfg, loss, exprs, name=name)


def _run(solver):
    solver.engine.forward_backward([c.fused for c in solver._compiled.values()])
    return solver.engine.grad.cpu().numpy().astype(np.float64)


def test_allen_cahn_closure_with_period_embedding(tmp_path):
    w = float(np.float32(2 * np.pi / 2.0))
    model = ppsci.arch.MLP(("t", "x"), ("u",), 3, 24, "tanh", periods={"x": (2.0, False)})
    net = T.make_net(2, [24, 24, 24], 1, periods={1: w}, bias_scale=0.05)
    set_model_weights(model, net)
    N = 50
    X = np.random.default_rng(42).uniform([0, -1], [1, 1], (N, 2)).astype(np.float32)
    eq = ppsci.equation.AllenCahn(eps=0.01)
    cst = _sup_constraint({"t": X[:, :1], "x": X[:, 1:]}, {"allen_cahn": np.zeros((N, 1), np.float32)}, eq.equations,
                          ppsci.loss.MSELoss("mean"))
    solver = _solver(tmp_path, model, {"EQ": cst})
    g = _run(solver)
    omodel = R.MLP(("t", "x"), ("u",), net.astype(np.float32).astype(np.float64))
    oc = dict(name="EQ", input={"t": X[:, :1].astype(np.float64), "x": X[:, 1:].astype(np.float64)},
              exprs={"allen_cahn": R.allen_cahn_fn(0.01)}, label={"allen_cahn": np.zeros((N, 1))}, reduction="mean")
    total, losses, gref, _ = R.loss_and_grads(omodel, [oc])
    assert solver._compiled["EQ"].fused.losses()["allen_cahn"] == pytest.approx(total, rel=2e-5)
    assert rel(g, gref) < 3e-5


def test_allen_cahn_reference_yaml_model_4x256(tmp_path):
    """The model of /root/reference/examples/allen_cahn/conf/allen_cahn.yaml:38-42 (4 x 256 tanh, periods on x):
    padded width 256 -> the feature-split kernels with four waves per tile."""
    w = float(np.float32(2 * np.pi / 2.0))
    model = ppsci.arch.MLP(("t", "x"), ("u",), 4, 256, "tanh", periods={"x": (2.0, False)})
    net = T.make_net(2, [256] * 4, 1, periods={1: w}, bias_scale=0.05)
    set_model_weights(model, net)
    N = 24
    X = np.random.default_rng(4).uniform([0, -1], [1, 1], (N, 2)).astype(np.float32)
    eq = ppsci.equation.AllenCahn(eps=0.01)
    cst = _sup_constraint({"t": X[:, :1], "x": X[:, 1:]}, {"allen_cahn": np.zeros((N, 1), np.float32)}, eq.equations,
                          ppsci.loss.MSELoss("mean"))
    solver = _solver(tmp_path, model, {"EQ": cst})
    g = _run(solver)
    omodel = R.MLP(("t", "x"), ("u",), net.astype(np.float32).astype(np.float64))
    oc = dict(name="EQ", input={"t": X[:, :1].astype(np.float64), "x": X[:, 1:].astype(np.float64)},
              exprs={"allen_cahn": R.allen_cahn_fn(0.01)}, label={"allen_cahn": np.zeros((N, 1))}, reduction="mean")
    total, losses, gref, _ = R.loss_and_grads(omodel, [oc])
    assert solver._compiled["EQ"].fused.losses()["allen_cahn"] == pytest.approx(total, rel=5e-5)
    assert rel(g, gref) < 1e-4


@pytest.mark.parametrize("detach_keys", [None, ("u", "v__y")])
def test_navier_stokes_2d_sum_loss_with_weights(tmp_path, detach_keys):
    model = ppsci.arch.MLP(("x", "y"), ("u", "v", "p"), 3, 20, "tanh")
    net = T.make_net(2, [20, 20, 20], 3, bias_scale=0.05)
    set_model_weights(model, net)
    N = 40
    X = np.random.default_rng(42).uniform(-0.05, 0.05, (N, 2)).astype(np.float32)
    eq = ppsci.equation.NavierStokes(0.01, 1.0, 2, False, detach_keys=detach_keys)
    keys = ("continuity", "momentum_x", "momentum_y")
    lab = {k: np.zeros((N, 1), np.float32) for k in keys}
    wts = {k: np.full((N, 1), 1e-4 * (i + 1), np.float32) for i, k in enumerate(keys)}
    cst = _sup_constraint({"x": X[:, :1], "y": X[:, 1:]}, lab, eq.equations, ppsci.loss.MSELoss("sum"), wts)
    solver = _solver(tmp_path, model, {"EQ": cst})
    g = _run(solver)
    omodel = R.MLP(("x", "y"), ("u", "v", "p"), net.astype(np.float32).astype(np.float64))
    oex = {k: R.lambdify(e, omodel) for k, e in eq.equations.items()}
    oc = dict(name="EQ", input={"x": X[:, :1].astype(np.float64), "y": X[:, 1:].astype(np.float64)}, exprs=oex,
              label={k: v.astype(np.float64) for k, v in lab.items()}, weight={k: v.astype(np.float64) for k, v in wts.items()},
              reduction="sum")
    total, losses, gref, _ = R.loss_and_grads(omodel, [oc])
    mine = solver._compiled["EQ"].fused.losses()
    for k in keys:
        assert mine[k] == pytest.approx(losses[k], rel=3e-5), k
    assert rel(g, gref) < 5e-5


def test_ns_expression_strings_match_reference_doctest():
    """equation/pde/base.py:99-111."""
    ns = ppsci.equation.NavierStokes(1.0, 1.0, 2, False)
    assert str(ns).splitlines()[1].strip() == "continuity: Derivative(u(x, y), x) + Derivative(v(x, y), y)"
    ns = ppsci.equation.NavierStokes(1.0, 1.0, 2, False, detach_keys=("u", "v__y"))
    lines = [l.strip() for l in str(ns).splitlines()]
    assert lines[1] == "continuity: detach(Derivative(v(x, y), y)) + Derivative(u(x, y), x)"
    assert lines[2] == ("momentum_x: detach(u(x, y))*Derivative(u(x, y), x) + v(x, y)*Derivative(u(x, y), y) + "
                        "1.0*Derivative(p(x, y), x) - 1.0*Derivative(u(x, y), (x, 2)) - 1.0*Derivative(u(x, y), (y, 2))")


def test_poisson_and_mixed_derivative_and_output_transform(tmp_path):
    """Symbolic differentiation of a transformed output (hard boundary constraint u = x(1-x) * net) and a
    mixed second derivative obtained by polarisation with the direction (x+y)."""
    model = ppsci.arch.MLP(("x", "y"), ("p",), 2, 20, "silu")
    net = T.make_net(2, [20, 20], 1, activation="silu", bias_scale=0.05)
    set_model_weights(model, net)
    model.register_output_transform(lambda inp, out: {"p": inp["x"] * (1.0 - inp["x"]) * out["p"]})
    N = 33
    X = np.random.default_rng(1).uniform(0, 1, (N, 2)).astype(np.float32)

    def mixed(out):
        return hessian(out["p"], out["x"]) + 2.0 * jacobian(jacobian(out["p"], out["x"]), out["y"]) + out["p"] * out["y"]

    cst = _sup_constraint({"x": X[:, :1], "y": X[:, 1:]}, {"r": np.full((N, 1), 0.3, np.float32)}, {"r": mixed},
                          ppsci.loss.MSELoss("mean", weight=2.5))
    solver = _solver(tmp_path, model, {"EQ": cst})
    g = _run(solver)
    # oracle with plain torch autograd
    net32 = net.astype(np.float32).astype(np.float64)
    omodel = R.MLP(("x", "y"), ("p",), net32)
    x = torch.tensor(X[:, :1].astype(np.float64), requires_grad=True)
    y = torch.tensor(X[:, 1:].astype(np.float64), requires_grad=True)
    p = x * (1.0 - x) * omodel({"x": x, "y": y})["p"]
    px = torch.autograd.grad(p.sum(), x, create_graph=True)[0]
    pxx = torch.autograd.grad(px.sum(), x, create_graph=True)[0]
    pxy = torch.autograd.grad(px.sum(), y, create_graph=True)[0]
    r = pxx + 2.0 * pxy + p * y
    loss = 2.5 * ((r - float(np.float32(0.3))) ** 2).mean()
    gref = torch.autograd.grad(loss, omodel.parameters(), allow_unused=True)
    gref = np.concatenate([(torch.zeros_like(q) if gg is None else gg).numpy().ravel() for gg, q in zip(gref, omodel.parameters())])
    assert solver._compiled["EQ"].fused.losses()["r"] == pytest.approx(float(loss), rel=3e-5)
    assert rel(g, gref) < 5e-5


def test_siren_mlp_laplace_residual_and_initialisation(tmp_path):
    """activation "siren" (activation.py:91-136): sin(30 z) with the first-layer / hidden-layer uniform
    initialisation of mlp.py:256-260; residual and parameter gradient against torch autograd."""
    ppsci.utils.misc.set_random_seed(3)
    model = ppsci.arch.MLP(("x", "y"), ("u",), 3, 20, "siren")
    w = [l.weight.cpu().numpy() for l in model.linears]
    assert np.abs(w[0]).max() <= 1.0 / 2 and np.abs(w[0]).max() > 0.4
    lim = np.sqrt(6.0 / 20) / 30.0
    assert np.abs(w[1]).max() <= lim and np.abs(w[1]).max() > 0.9 * lim
    assert float(model.linears[1].bias.abs().max()) == 0.0
    net = T.make_net(2, [20, 20, 20], 1, activation="siren", bias_scale=0.05)
    set_model_weights(model, net)
    N = 40
    X = np.random.default_rng(2).uniform(-1, 1, (N, 2)).astype(np.float32)
    eq = ppsci.equation.Laplace(dim=2)
    cst = _sup_constraint({"x": X[:, :1], "y": X[:, 1:]}, {"laplace": np.zeros((N, 1), np.float32)}, eq.equations,
                          ppsci.loss.MSELoss("mean"))
    solver = _solver(tmp_path, model, {"EQ": cst})
    g = _run(solver)
    omodel = R.MLP(("x", "y"), ("u",), net.astype(np.float32).astype(np.float64))
    oc = dict(name="EQ", input={"x": X[:, :1].astype(np.float64), "y": X[:, 1:].astype(np.float64)},
              exprs={"laplace": R.lambdify(R.laplace_exprs(2)["laplace"], omodel)}, label={"laplace": np.zeros((N, 1))},
              reduction="mean")
    total, losses, gref, _ = R.loss_and_grads(omodel, [oc])
    # the factor 30 amplifies the fp32 rounding of every pre-activation: looser than the tanh cases
    assert solver._compiled["EQ"].fused.losses()["laplace"] == pytest.approx(total, rel=2e-4)
    assert rel(g, gref) < 2e-4


def test_lambdify_with_the_remaining_sympy_functions(tmp_path):
    """An expression over SYMPY_TO_PADDLE's less common entries (symbolic.py:79-108) around first and second
    derivatives of the net: loss and parameter gradient against the oracle's lambdify."""
    model = ppsci.arch.MLP(("x", "y"), ("u",), 2, 20, "tanh")
    net = T.make_net(2, [20, 20], 1, bias_scale=0.05)
    set_model_weights(model, net)
    N = 48
    X = np.random.default_rng(8).uniform(0.1, 0.9, (N, 2)).astype(np.float32)
    x, y = sp.symbols("x y")
    u = sp.Function("u")(x, y)
    expr = (sp.atan(u.diff(x)) + sp.erf(u) * sp.asinh(u.diff(y, 2)) + sp.atan2(u.diff(x, 2), 1 + x * x)
            + sp.asin(u / 4) * sp.acos(x / 2) + sp.atanh(y / 2) * u + sp.acosh(2 + u * u)
            + sp.loggamma(2 + y) * u.diff(y) + sp.floor(4 * x) * u + sp.ceiling(3 * y) * u.diff(x)
            + sp.Max(u, u.diff(x), 0.1) + sp.Min(u.diff(y), x) + sp.Heaviside(x - 0.5) * u + sp.sign(y - 0.4) * u)
    cst = _sup_constraint({"x": X[:, :1], "y": X[:, 1:]}, {"r": np.zeros((N, 1), np.float32)}, {"r": expr},
                          ppsci.loss.MSELoss("mean"))
    solver = _solver(tmp_path, model, {"EQ": cst})
    g = _run(solver)
    omodel = R.MLP(("x", "y"), ("u",), net.astype(np.float32).astype(np.float64))
    oc = dict(name="EQ", input={"x": X[:, :1].astype(np.float64), "y": X[:, 1:].astype(np.float64)},
              exprs={"r": R.lambdify(expr, omodel)}, label={"r": np.zeros((N, 1))}, reduction="mean")
    total, losses, gref, _ = R.loss_and_grads(omodel, [oc])
    assert solver._compiled["EQ"].fused.losses()["r"] == pytest.approx(total, rel=3e-5)
    assert rel(g, gref) < 5e-5


@pytest.mark.parametrize("variant", ["weight_norm", "random_weight", "fourier", "fourier_rwf_periods"])
def test_factored_and_fourier_mlp_variants(tmp_path, variant):
    """MLP(weight_norm=True) / MLP(random_weight=...) / MLP(fourier=...) (mlp.py:31-136, :233-249; the model of
    examples/allen_cahn/conf/allen_cahn_causal_fourier_rwf.yaml:35-48 at a smaller width): Allen-Cahn loss and the
    gradient w.r.t. the reference's trainable tensors (fourier_emb.kernel, weight_v, weight_g, bias) against torch
    autograd, then one Adam step on them."""
    H, nl = 32, 3
    rng = np.random.default_rng(12)
    kw, okw = {}, {}
    periods = None
    if variant in ("weight_norm", "random_weight"):
        kw = {"weight_norm": True} if variant == "weight_norm" else {"random_weight": {"mean": 0.5, "std": 0.1}}
        okw = {"factor": variant}
    elif variant == "fourier":
        kw = {"fourier": {"dim": H, "scale": 1.0}}
    else:
        kw = {"fourier": {"dim": H, "scale": 1.0}, "random_weight": {"mean": 0.5, "std": 0.1},
              "periods": {"x": (2.0, False)}}
        okw = {"factor": "random_weight"}
        periods = {1: float(np.float32(2 * np.pi / 2.0))}
    ppsci.utils.misc.set_random_seed(5)
    model = ppsci.arch.MLP(("t", "x"), ("u",), nl, H, "tanh", **kw)
    fourier = "fourier" in variant
    d0 = 2 + (1 if periods else 0)
    net = T.make_net(2, [H] * nl, 1, periods=periods, bias_scale=0.05)
    if fourier:
        B = rng.normal(0.0, 1.0, (d0, H // 2))
        net.weights[0] = rng.uniform(-0.3, 0.3, (H, H))  # the first Linear sees the embedding, not the raw inputs
        okw["fourier_kernel"] = B.astype(np.float32).astype(np.float64)
    if "factor" in okw:
        okw["weight_g"] = [rng.uniform(0.6, 1.6, H).astype(np.float32).astype(np.float64) for _ in range(nl)]
        if okw["factor"] == "random_weight":  # last_fc is factorised too (mlp.py:266-272)
            okw["weight_g"].append(rng.uniform(0.6, 1.6, 1).astype(np.float32).astype(np.float64))
    net32 = net.astype(np.float32).astype(np.float64)
    omodel = R.MLP(("t", "x"), ("u",), net32, **okw)
    # the model's trainable tensors are in the oracle's parameters() order
    assert [tuple(p.shape) for p in model.parameters()] == [tuple(p.shape) for p in omodel.parameters()]
    flat = np.concatenate([p.detach().numpy().ravel() for p in omodel.parameters()])
    model.flat_params.copy_(torch.tensor(flat, dtype=torch.float32).to(model.flat_params.device))
    names = [n for n, _ in model.named_parameters()]
    if fourier:
        assert names[0] == "fourier_emb.kernel"
    if "factor" in okw:
        assert "linears.0.weight_v" in names and "linears.0.weight_g" in names
    N = 45
    X = rng.uniform([0, -1], [1, 1], (N, 2)).astype(np.float32)
    eq = ppsci.equation.AllenCahn(eps=0.01)
    cst = _sup_constraint({"t": X[:, :1], "x": X[:, 1:]}, {"allen_cahn": np.zeros((N, 1), np.float32)}, eq.equations,
                          ppsci.loss.MSELoss("mean"))
    solver = _solver(tmp_path, model, {"EQ": cst})
    solver._materialize()
    solver.engine.forward_backward([c.fused for c in solver._compiled.values()])
    g = solver._train_grad().cpu().numpy().astype(np.float64)
    oc = dict(name="EQ", input={"t": X[:, :1].astype(np.float64), "x": X[:, 1:].astype(np.float64)},
              exprs={"allen_cahn": R.allen_cahn_fn(0.01)}, label={"allen_cahn": np.zeros((N, 1))}, reduction="mean")
    total, losses, gref, _ = R.loss_and_grads(omodel, [oc])
    assert solver._compiled["EQ"].fused.losses()["allen_cahn"] == pytest.approx(total, rel=5e-5)
    assert rel(g, gref) < 1e-4
    # numeric forward of the model (eager `model(dict)`)
    out = model({"t": X[:, :1], "x": X[:, 1:]})["u"].cpu().numpy()
    oref = omodel({"t": torch.tensor(X[:, :1].astype(np.float64)), "x": torch.tensor(X[:, 1:].astype(np.float64))})["u"]
    assert rel(out, oref.detach().numpy()) < 1e-5
    # one training step through Solver.train moves the trainable tensors like the oracle's Adam
    solver.train()
    adam = R.Adam(flat.size, lr=1e-3)
    pref = adam.step(flat, gref)
    assert rel(model.flat_params.cpu().numpy(), pref) < 1e-5


def test_autodiff_errors_follow_reference():
    from paddlescience_amd.graph import Sym

    model = ppsci.arch.MLP(("x",), ("u",), 1, 16)
    out = model({"x": Sym.input("x")})
    x = Sym.input("x")
    with pytest.raises(ValueError):
        jacobian(out["u"], x, i=1)
    with pytest.raises(ValueError):
        hessian(out["u"], x, component=0)
    third = jacobian(hessian(out["u"], x), x)  # orders 3 and 4 are carried by the kernels' higher-order streams
    assert repr(third) == "u__x__x__x" and repr(jacobian(third, x)) == "u__x__x__x__x"
    with pytest.raises(NotImplementedError):
        jacobian(jacobian(third, x), x)  # fifth order
    with pytest.raises(TypeError):
        jacobian(out["u"], torch.zeros(3, 1))
    ppsci.autodiff.clear()
    assert ppsci.autodiff.hessian.Hs == {} and ppsci.autodiff.jacobian.Js == {}


def test_learnable_equation_parameters_vibration(tmp_path):
    """ParameterNode (symbolic.py:471-485) through equation/pde/viv.py:41-62: rho*eta_tt + exp(k1)*eta_t + exp(k2)*eta
    with learnable k1, k2 trained together with the network by Adam((model,) + equations) (examples/fsi/viv.py:121):
    loss, network gradient, d loss / d k1, d loss / d k2 and one optimizer step against torch autograd."""
    from paddlescience_amd.equation.pde.base import EqParamStore

    EqParamStore.reset()
    model = ppsci.arch.MLP(("t_f",), ("eta",), 2, 20, "tanh")
    net = T.make_net(1, [20, 20], 1, bias_scale=0.1)
    set_model_weights(model, net)
    eq = ppsci.equation.Vibration(2.0, 0.7, -0.4)
    assert [p.name for p in eq.parameters()] == [eq.k1.name, eq.k2.name] and set(eq.state_dict()) == {"0", "1"}
    N = 41
    rng = np.random.default_rng(6)
    t = rng.uniform(0, 1, (N, 1)).astype(np.float32)
    f_lab = rng.standard_normal((N, 1)).astype(np.float32)
    eta_lab = rng.standard_normal((N, 1)).astype(np.float32) * 0.1
    cfg = {"dataset": {"name": "IterableNamedArrayDataset", "input": {"t_f": t}, "label": {"eta": eta_lab, "f": f_lab}}}
    cst = ppsci.constraint.SupervisedConstraint(cfg, ppsci.loss.MSELoss("mean"), {"eta": lambda out: out["eta"], **eq.equations},
                                                name="Sup")
    opt = ppsci.optimizer.Adam(1e-2)((model,) + (eq,))
    solver = ppsci.solver.Solver(model, {"Sup": cst}, str(tmp_path), opt, epochs=1, iters_per_epoch=1,
                                 equation={"VIV": eq})
    solver.engine.forward_backward([c.fused for c in solver._compiled.values()])
    g = solver.engine.grad.cpu().numpy().astype(np.float64)
    store = EqParamStore.get()
    gk = store.grad.cpu().numpy().astype(np.float64)[:2]
    # oracle
    omodel = R.MLP(("t_f",), ("eta",), net.astype(np.float32).astype(np.float64))
    k = {eq.k1.name: torch.tensor(float(np.float32(0.7)), dtype=torch.float64, requires_grad=True),
         eq.k2.name: torch.tensor(float(np.float32(-0.4)), dtype=torch.float64, requires_grad=True)}
    oc = dict(name="Sup", input={"t_f": t.astype(np.float64)},
              exprs={"eta": lambda d: d["eta"], "f": R.lambdify(eq.equations["f"], omodel, extra_parameters=k)},
              label={"eta": eta_lab.astype(np.float64), "f": f_lab.astype(np.float64)}, reduction="mean")
    losses_all, _, _ = R.train_forward(omodel, [oc])
    total = R.loss_sum(losses_all)
    grads = torch.autograd.grad(total, omodel.parameters() + list(k.values()), allow_unused=True)
    gref = np.concatenate([(torch.zeros_like(p) if gg is None else gg).numpy().ravel()
                           for gg, p in zip(grads[:-2], omodel.parameters())])
    got = solver._compiled["Sup"].fused.losses()
    for key in ("eta", "f"):
        assert got[key] == pytest.approx(float(losses_all[key]), rel=3e-5)
    assert rel(g, gref) < 5e-5
    assert gk[0] == pytest.approx(float(grads[-2]), rel=5e-5) and gk[1] == pytest.approx(float(grads[-1]), rel=5e-5)
    # one training step: network and the two exponents move by Adam's first step (lr * sign-like update)
    solver.train()
    adam = R.Adam(2, lr=1e-2)
    kref = adam.step(np.array([float(np.float32(0.7)), float(np.float32(-0.4))]), np.array([float(grads[-2]), float(grads[-1])]))
    assert eq.k1.item() == pytest.approx(kref[0], rel=1e-5) and eq.k2.item() == pytest.approx(kref[1], rel=1e-5)
    EqParamStore.reset()


def test_model_list_two_networks_in_one_residual(tmp_path):
    """ppsci.arch.ModelList (model_list.py:24-72): a velocity net (x, y) -> (u, v) and a pressure net (x, y) -> p with a
    different depth, width and activation, coupled in NS-like residuals; losses and the gradient of BOTH members against
    torch autograd, then Adam steps on the shared flat buffer."""
    net_a = T.make_net(2, [20, 20], 2, seed=3, bias_scale=0.05)
    net_b = T.make_net(2, [16, 16, 16], 1, seed=4, activation="silu", bias_scale=0.05)
    ma = ppsci.arch.MLP(("x", "y"), ("u", "v"), 2, 20, "tanh")
    mb = ppsci.arch.MLP(("x", "y"), ("p",), 3, 16, "silu")
    set_model_weights(ma, net_a)
    set_model_weights(mb, net_b)
    model = ppsci.arch.ModelList((ma, mb))
    assert model.output_keys == ("u", "v", "p") and set(model.input_keys) == {"x", "y"}
    assert "model_list.1.last_fc.bias" in model.state_dict()
    np.testing.assert_array_equal(mb.flat_params.cpu().numpy(), T.flat_params(net_b).astype(np.float32))
    N = 39
    X = np.random.default_rng(10).uniform(-1, 1, (N, 2)).astype(np.float32)
    x, y = sp.symbols("x y")
    u, v, p = (sp.Function(k)(x, y) for k in ("u", "v", "p"))
    nu = 0.05
    exprs = {"continuity": u.diff(x) + v.diff(y),
             "momentum_x": u * u.diff(x) + v * u.diff(y) - nu * (u.diff(x, 2) + u.diff(y, 2)) + p.diff(x),
             "momentum_y": u * v.diff(x) + v * v.diff(y) - nu * (v.diff(x, 2) + v.diff(y, 2)) + p.diff(y)}
    lab = {k: np.zeros((N, 1), np.float32) for k in exprs}
    cst = _sup_constraint({"x": X[:, :1], "y": X[:, 1:]}, lab, exprs, ppsci.loss.MSELoss("sum"))
    solver = _solver(tmp_path, model, {"EQ": cst})
    solver.engine.forward_backward([c.fused for c in solver._compiled.values()])
    g = solver.engine.grad.cpu().numpy().astype(np.float64)
    oa = R.MLP(("x", "y"), ("u", "v"), net_a.astype(np.float32).astype(np.float64))
    ob = R.MLP(("x", "y"), ("p",), net_b.astype(np.float32).astype(np.float64))
    om = R.ModelList((oa, ob))
    oc = dict(name="EQ", input={"x": X[:, :1].astype(np.float64), "y": X[:, 1:].astype(np.float64)},
              exprs={k: R.lambdify(e, om) for k, e in exprs.items()}, label={k: np.zeros((N, 1)) for k in exprs},
              reduction="sum")
    losses_all, _, _ = R.train_forward(om, [oc])
    total = R.loss_sum(losses_all)
    got = solver._compiled["EQ"].fused.losses()
    for k in exprs:
        assert got[k] == pytest.approx(float(losses_all[k]), rel=3e-5)
    for mem, omem in ((ma, oa), (mb, ob)):
        gr = torch.autograd.grad(total, omem.parameters(), allow_unused=True, retain_graph=True)
        gref = np.concatenate([(torch.zeros_like(q) if gg is None else gg).numpy().ravel() for gg, q in zip(gr, omem.parameters())])
        off, n = mem._param_offset, mem.flat_params.numel()
        assert rel(g[off:off + n], gref) < 5e-5
    # eager forward of the list and a training run
    out = model({"x": X[:, :1], "y": X[:, 1:]})
    ref = om({"x": torch.tensor(X[:, :1].astype(np.float64)), "y": torch.tensor(X[:, 1:].astype(np.float64))})
    for k in ("u", "v", "p"):
        assert rel(out[k].cpu().numpy(), ref[k].detach().numpy()) < 1e-5
    before = model.flat_params.clone()
    solver.train()
    assert float((model.flat_params - before).abs().max()) > 1e-4
    pred = solver.predict({"x": X[:5, :1], "y": X[:5, 1:]}, return_numpy=True)
    assert set(pred) >= {"u", "v", "p"}


def test_model_list_constraints_touching_different_members(tmp_path):
    """Two constraints, each on ONE member of a ModelList (the second member also takes fewer inputs): the gradient
    slice of a member comes only from the constraints that evaluate it, every step."""
    net_a = T.make_net(2, [20, 20], 1, seed=3, bias_scale=0.05)
    net_b = T.make_net(1, [16, 16], 1, seed=4, bias_scale=0.05)
    ma = ppsci.arch.MLP(("x", "y"), ("u",), 2, 20, "tanh")
    mb = ppsci.arch.MLP(("x",), ("q",), 2, 16, "tanh")
    set_model_weights(ma, net_a)
    set_model_weights(mb, net_b)
    model = ppsci.arch.ModelList((ma, mb))
    N = 30
    rng = np.random.default_rng(12)
    X = rng.uniform(-1, 1, (N, 2)).astype(np.float32)
    eq = ppsci.equation.Laplace(dim=2)
    c1 = _sup_constraint({"x": X[:, :1], "y": X[:, 1:]}, {"laplace": np.zeros((N, 1), np.float32)}, eq.equations,
                         ppsci.loss.MSELoss("mean"), name="A")
    labq = rng.standard_normal((N, 1)).astype(np.float32)
    c2 = _sup_constraint({"x": X[:, :1], "y": X[:, 1:]}, {"q": labq}, {"q": lambda out: out["q"] + jacobian(out["q"], out["x"])},
                         ppsci.loss.MSELoss("mean"), name="B")
    solver = _solver(tmp_path, model, {"A": c1, "B": c2})
    for _ in range(2):  # the second pass must not accumulate onto the first
        solver.engine.forward_backward([c.fused for c in solver._compiled.values()])
    g = solver.engine.grad.cpu().numpy().astype(np.float64)
    oa = R.MLP(("x", "y"), ("u",), net_a.astype(np.float32).astype(np.float64))
    ob = R.MLP(("x",), ("q",), net_b.astype(np.float32).astype(np.float64))
    # member A: Laplace residual
    ca = dict(name="A", input={"x": X[:, :1].astype(np.float64), "y": X[:, 1:].astype(np.float64)},
              exprs={"laplace": R.lambdify(R.laplace_exprs(2)["laplace"], oa)}, label={"laplace": np.zeros((N, 1))},
              reduction="mean")
    _, _, ga, _ = R.loss_and_grads(oa, [ca])
    xb = torch.tensor(X[:, :1].astype(np.float64), requires_grad=True)
    qb = ob({"x": xb})["q"]
    rb = qb + torch.autograd.grad(qb.sum(), xb, create_graph=True)[0]
    lb = ((rb - torch.tensor(labq.astype(np.float64))) ** 2).mean()
    gb = torch.autograd.grad(lb, ob.parameters(), allow_unused=True)
    gb = np.concatenate([(torch.zeros_like(q) if gg is None else gg).numpy().ravel() for gg, q in zip(gb, ob.parameters())])
    assert rel(g[ma._param_offset:ma._param_offset + ga.size], ga) < 5e-5
    assert rel(g[mb._param_offset:mb._param_offset + gb.size], gb) < 5e-5


@pytest.mark.parametrize("act", ["swish", "stan"])
def test_learnable_activation_mlp(tmp_path, act):
    """MLP(activation="swish" / "stan") (activation.py:28-58): the trainable tensors are in the reference's parameters()
    order (linears, acts.<l>.beta -- a scalar for Swish, [hidden] for Stan --, last_fc); Poisson-type loss and the
    gradient of every one of them against torch autograd; betas start at 1."""
    H, nl = 20, 3
    ppsci.utils.misc.set_random_seed(8)
    model = ppsci.arch.MLP(("x", "y"), ("u",), nl, H, act)
    names = [n for n, _ in model.named_parameters()]
    assert names[2 * nl:2 * nl + nl] == [f"acts.{l}.beta" for l in range(nl)] and names[-2:] == ["last_fc.weight", "last_fc.bias"]
    assert all(float(p.min()) == 1.0 for n, p in model.named_parameters() if n.startswith("acts."))
    rng = np.random.default_rng(14)
    net = T.make_net(2, [H] * nl, 1, activation=act, bias_scale=0.05)
    beta = [np.float32(rng.uniform(0.7, 1.3)) if act == "swish" else rng.uniform(0.7, 1.3, H).astype(np.float32) for _ in range(nl)]
    omodel = R.MLP(("x", "y"), ("u",), net.astype(np.float32).astype(np.float64),
                   act_beta=[np.asarray(b, dtype=np.float64) for b in beta])
    assert [tuple(p.shape) for p in model.parameters()] == [tuple(p.shape) for p in omodel.parameters()]
    flat = np.concatenate([p.detach().numpy().ravel() for p in omodel.parameters()])
    model.flat_params.copy_(torch.tensor(flat, dtype=torch.float32).to(model.flat_params.device))
    N = 33
    X = rng.uniform(-1, 1, (N, 2)).astype(np.float32)
    eq = ppsci.equation.Laplace(dim=2)
    cst = _sup_constraint({"x": X[:, :1], "y": X[:, 1:]}, {"laplace": np.full((N, 1), 0.2, np.float32)}, eq.equations,
                          ppsci.loss.MSELoss("mean"))
    solver = _solver(tmp_path, model, {"EQ": cst})
    solver._materialize()
    solver.engine.forward_backward([c.fused for c in solver._compiled.values()])
    g = solver._train_grad().cpu().numpy().astype(np.float64)
    oc = dict(name="EQ", input={"x": X[:, :1].astype(np.float64), "y": X[:, 1:].astype(np.float64)},
              exprs={"laplace": R.lambdify(R.laplace_exprs(2)["laplace"], omodel)},
              label={"laplace": np.full((N, 1), float(np.float32(0.2)))}, reduction="mean")
    total, losses, gref, _ = R.loss_and_grads(omodel, [oc])
    assert solver._compiled["EQ"].fused.losses()["laplace"] == pytest.approx(total, rel=5e-5)
    assert rel(g, gref) < 1e-4
    off = sum(int(np.prod(p.shape)) for p in model.parameters()[:2 * nl])
    nb = nl if act == "swish" else nl * H
    assert rel(g[off:off + nb], gref[off:off + nb]) < 1e-4  # the betas on their own
    solver.train()
    adam = R.Adam(flat.size, lr=1e-3)
    assert rel(model.flat_params.cpu().numpy(), adam.step(flat, gref)) < 1e-5


@pytest.mark.parametrize("width,mixed", [(20, 0.5), (128, 0.0)])
def test_registered_input_transform_nonlinear_features(tmp_path, width, mixed):
    """Arch.register_input_transform (base.py:150-183, MLP.forward mlp.py:299-300) in the style of
    examples/pipe/poiseuille_flow.py:76-83: the network takes (sin(bx+c), cos(bx+c), y, nu) computed from the raw
    (x, y, nu); the residual differentiates w.r.t. the RAW x and y (first, second and mixed), so the chain rule
    through the features is carried by the input streams (EMBED_STREAMS).  Width 128 uses the feature-split kernels."""
    import ppsci.functional as F

    b, c = 1.7, 0.3
    feats = ("sin(x)", "cos(x)", "y", "nu")

    def trans(d):
        return {"sin(x)": 0.8 * F.sin(b * d["x"] + c), "cos(x)": 0.8 * F.cos(b * d["x"] + c), "y": d["y"], "nu": d["nu"]}

    model = ppsci.arch.MLP(feats, ("u",), 2, width, "tanh")
    net = T.make_net(4, [width, width], 1, bias_scale=0.05)
    set_model_weights(model, net)
    model.register_input_transform(trans)
    model.register_output_transform(lambda d, out: {"u": out["u"] * (1.0 - d["y"] * d["y"])})
    N = 35
    rng = np.random.default_rng(21)
    X = rng.uniform(-1, 1, (N, 2)).astype(np.float32)
    nu = rng.uniform(0.01, 0.1, (N, 1)).astype(np.float32)

    def resid(out):
        u, x, y = out["u"], out["x"], out["y"]
        r = jacobian(u, x) * u - out["nu"] * (hessian(u, x) + hessian(u, y))
        return r + mixed * jacobian(jacobian(u, x), y) if mixed else r  # (the mixed term needs the (3, 3) stream set)

    cst = _sup_constraint({"x": X[:, :1], "y": X[:, 1:], "nu": nu}, {"r": np.zeros((N, 1), np.float32)}, {"r": resid},
                          ppsci.loss.MSELoss("mean"))
    solver = _solver(tmp_path, model, {"EQ": cst})
    g = _run(solver)
    # torch autograd on the same construction
    omodel = R.MLP(feats, ("u",), net.astype(np.float32).astype(np.float64))
    x = torch.tensor(X[:, :1].astype(np.float64), requires_grad=True)
    y = torch.tensor(X[:, 1:].astype(np.float64), requires_grad=True)
    tnu = torch.tensor(nu.astype(np.float64))
    k08, kb, kc = float(np.float32(0.8)), float(np.float32(b)), float(np.float32(c))
    fx = {"sin(x)": k08 * torch.sin(kb * x + kc), "cos(x)": k08 * torch.cos(kb * x + kc), "y": y, "nu": tnu}
    u = omodel(fx)["u"] * (1.0 - y * y)
    ux = torch.autograd.grad(u.sum(), x, create_graph=True)[0]
    uy = torch.autograd.grad(u.sum(), y, create_graph=True)[0]
    uxx = torch.autograd.grad(ux.sum(), x, create_graph=True)[0]
    uyy = torch.autograd.grad(uy.sum(), y, create_graph=True)[0]
    uxy = torch.autograd.grad(ux.sum(), y, create_graph=True)[0]
    r = ux * u - tnu * (uxx + uyy) + mixed * uxy
    loss = (r**2).mean()
    gref = torch.autograd.grad(loss, omodel.parameters(), allow_unused=True)
    gref = np.concatenate([(torch.zeros_like(q) if gg is None else gg).numpy().ravel() for gg, q in zip(gref, omodel.parameters())])
    assert solver._compiled["EQ"].fused.losses()["r"] == pytest.approx(float(loss.detach()), rel=5e-5)
    assert rel(g, gref) < 1e-4
    # eager forward runs the transform on tensors
    dev_ = model.flat_params.device
    out = model({"x": torch.tensor(X[:, :1], device=dev_), "y": torch.tensor(X[:, 1:], device=dev_),
                 "nu": torch.tensor(nu, device=dev_)})["u"].cpu().numpy()
    assert rel(out, u.detach().numpy()) < 1e-5


def test_model_list_with_a_learnable_activation_member(tmp_path):
    """A ModelList whose members have different parameter layouts: a swish net (trainable betas: kernel layout !=
    trainable layout) next to a plain tanh net.  Gradient w.r.t. the trainable tensors of both, and an Adam step."""
    rng = np.random.default_rng(15)
    net_a = T.make_net(2, [20, 20], 1, seed=3, activation="swish", bias_scale=0.05)
    net_b = T.make_net(2, [16, 16], 1, seed=4, bias_scale=0.05)
    ma = ppsci.arch.MLP(("x", "y"), ("u",), 2, 20, "swish")
    mb = ppsci.arch.MLP(("x", "y"), ("p",), 2, 16, "tanh")
    beta = [np.float32(rng.uniform(0.7, 1.3)) for _ in range(2)]
    oa = R.MLP(("x", "y"), ("u",), net_a.astype(np.float32).astype(np.float64), act_beta=[np.asarray(b, np.float64) for b in beta])
    ob = R.MLP(("x", "y"), ("p",), net_b.astype(np.float32).astype(np.float64))
    ma.flat_params.copy_(torch.tensor(np.concatenate([p.detach().numpy().ravel() for p in oa.parameters()]), dtype=torch.float32))
    set_model_weights(mb, net_b)
    model = ppsci.arch.ModelList((ma, mb))
    assert model.reparam and ma.flat_params.data_ptr() == model.flat_params.data_ptr()
    N = 31
    X = rng.uniform(-1, 1, (N, 2)).astype(np.float32)

    def resid(out):
        return hessian(out["u"], out["x"]) + jacobian(out["p"], out["y"]) * out["u"]

    cst = _sup_constraint({"x": X[:, :1], "y": X[:, 1:]}, {"r": np.zeros((N, 1), np.float32)}, {"r": resid},
                          ppsci.loss.MSELoss("mean"))
    solver = _solver(tmp_path, model, {"EQ": cst})
    solver._materialize()
    solver.engine.forward_backward([c.fused for c in solver._compiled.values()])
    g = solver._train_grad().cpu().numpy().astype(np.float64)
    x = torch.tensor(X[:, :1].astype(np.float64), requires_grad=True)
    y = torch.tensor(X[:, 1:].astype(np.float64), requires_grad=True)
    u = oa({"x": x, "y": y})["u"]
    p = ob({"x": x, "y": y})["p"]
    ux = torch.autograd.grad(u.sum(), x, create_graph=True)[0]
    uxx = torch.autograd.grad(ux.sum(), x, create_graph=True)[0]
    py = torch.autograd.grad(p.sum(), y, create_graph=True)[0]
    loss = ((uxx + py * u) ** 2).mean()
    for mem, om in ((ma, oa), (mb, ob)):
        gr = torch.autograd.grad(loss, om.parameters(), allow_unused=True, retain_graph=True)
        gref = np.concatenate([(torch.zeros_like(q) if gg is None else gg).numpy().ravel() for gg, q in zip(gr, om.parameters())])
        off = mem._train_offset
        assert rel(g[off:off + gref.size], gref) < 1e-4
    assert solver._compiled["EQ"].fused.losses()["r"] == pytest.approx(float(loss.detach()), rel=5e-5)
    before = model.flat_params.clone()
    solver.train()
    assert float((model.flat_params - before).abs().max()) > 1e-4


def test_unsteady_navier_stokes_and_laplace_3d(tmp_path):
    """NavierStokes(dim=2, time=True) (navier_stokes.py:70-151: first derivatives along t, x, y, second along x, y ->
    the (3, 3) stream kernels) and Laplace(dim=3) (laplace.py:40-55), through the sympy path against the oracle."""
    model = ppsci.arch.MLP(("t", "x", "y"), ("u", "v", "p"), 2, 24, "tanh")
    net = T.make_net(3, [24, 24], 3, bias_scale=0.05)
    set_model_weights(model, net)
    N = 29
    rng = np.random.default_rng(33)
    X = rng.uniform(-1, 1, (N, 3)).astype(np.float32)
    eq = ppsci.equation.NavierStokes(0.02, 1.3, 2, True)
    keys = ("continuity", "momentum_x", "momentum_y")
    cst = _sup_constraint({"t": X[:, :1], "x": X[:, 1:2], "y": X[:, 2:]}, {k: np.zeros((N, 1), np.float32) for k in keys},
                          eq.equations, ppsci.loss.MSELoss("sum"))
    solver = _solver(tmp_path, model, {"EQ": cst})
    g = _run(solver)
    omodel = R.MLP(("t", "x", "y"), ("u", "v", "p"), net.astype(np.float32).astype(np.float64))
    oc = dict(name="EQ", input={"t": X[:, :1].astype(np.float64), "x": X[:, 1:2].astype(np.float64), "y": X[:, 2:].astype(np.float64)},
              exprs={k: R.lambdify(eq.equations[k], omodel) for k in keys}, label={k: np.zeros((N, 1)) for k in keys},
              reduction="sum")
    total, losses, gref, _ = R.loss_and_grads(omodel, [oc])
    got = solver._compiled["EQ"].fused.losses()
    for k in keys:
        assert got[k] == pytest.approx(losses[k], rel=5e-5)
    assert rel(g, gref) < 5e-5
    # 3-D Laplace: three second-order streams
    m3 = ppsci.arch.MLP(("x", "y", "z"), ("u",), 2, 24, "silu")
    net3 = T.make_net(3, [24, 24], 1, activation="silu", bias_scale=0.05)
    set_model_weights(m3, net3)
    lap = ppsci.equation.Laplace(dim=3)
    c3 = _sup_constraint({"x": X[:, :1], "y": X[:, 1:2], "z": X[:, 2:]}, {"laplace": np.zeros((N, 1), np.float32)},
                         lap.equations, ppsci.loss.MSELoss("mean"))
    s3 = _solver(tmp_path, m3, {"EQ": c3})
    g3 = _run(s3)
    o3 = R.MLP(("x", "y", "z"), ("u",), net3.astype(np.float32).astype(np.float64))
    oc3 = dict(name="EQ", input={"x": X[:, :1].astype(np.float64), "y": X[:, 1:2].astype(np.float64), "z": X[:, 2:].astype(np.float64)},
               exprs={"laplace": R.lambdify(lap.equations["laplace"], o3)}, label={"laplace": np.zeros((N, 1))}, reduction="mean")
    t3, _, gref3, _ = R.loss_and_grads(o3, [oc3])
    assert s3._compiled["EQ"].fused.losses()["laplace"] == pytest.approx(t3, rel=5e-5)
    assert rel(g3, gref3) < 5e-5


def test_unsteady_navier_stokes_3d(tmp_path):
    """NavierStokes(dim=3, time=True) (navier_stokes.py:70-151): four outputs, first derivatives along t, x, y, z and second
    along x, y, z -> the (4, 3) stream set (S = 8), through the sympy path against the oracle's reverse-over-reverse."""
    model = ppsci.arch.MLP(("t", "x", "y", "z"), ("u", "v", "w", "p"), 2, 24, "tanh")
    net = T.make_net(4, [24, 24], 4, bias_scale=0.05)
    set_model_weights(model, net)
    N = 23
    X = np.random.default_rng(34).uniform(-1, 1, (N, 4)).astype(np.float32)
    eq = ppsci.equation.NavierStokes(0.02, 1.3, 3, True)
    keys = ("continuity", "momentum_x", "momentum_y", "momentum_z")
    inp = {k: X[:, j:j + 1] for j, k in enumerate(("t", "x", "y", "z"))}
    cst = _sup_constraint(inp, {k: np.zeros((N, 1), np.float32) for k in keys}, eq.equations, ppsci.loss.MSELoss("sum"))
    solver = _solver(tmp_path, model, {"EQ": cst})
    g = _run(solver)
    assert solver._compiled["EQ"].fused.streams.S == 8
    omodel = R.MLP(("t", "x", "y", "z"), ("u", "v", "w", "p"), net.astype(np.float32).astype(np.float64))
    oc = dict(name="EQ", input={k: v.astype(np.float64) for k, v in inp.items()},
              exprs={k: R.lambdify(eq.equations[k], omodel) for k in keys}, label={k: np.zeros((N, 1)) for k in keys},
              reduction="sum")
    total, losses, gref, _ = R.loss_and_grads(omodel, [oc])
    got = solver._compiled["EQ"].fused.losses()
    for k in keys:
        assert got[k] == pytest.approx(losses[k], rel=5e-5)
    assert rel(g, gref) < 5e-5


def test_periodic_constraint_batches_and_training_step(tmp_path):
    """PeriodicConstraint (periodic_constraint.py:60-166) on a time x rectangle domain: every batch is [half ; images
    of that half along the periodic key]; two training iterations (two different batches) against the oracle."""
    geom = ppsci.geometry.TimeXGeometry(ppsci.geometry.TimeDomain(0.0, 1.0, time_step=0.125),
                                        ppsci.geometry.Rectangle((-1.0, 0.0), (1.0, 2.0)))
    model = ppsci.arch.MLP(("t", "x", "y"), ("u",), 2, 20, "tanh")
    net = T.make_net(3, [20, 20], 1, bias_scale=0.05)
    set_model_weights(model, net)
    x, y, t = sp.symbols("x y t")
    u = sp.Function("u")(t, x, y)
    exprs = {"u": lambda out: out["u"], "u_x": u.diff(x)}
    np.random.seed(7)
    bs, iters = 16, 2
    cst = ppsci.constraint.PeriodicConstraint(
        exprs, {"u": 0, "u_x": 0}, geom, "x", {"dataset": "NamedArrayDataset", "batch_size": bs, "iters_per_epoch": iters,
                                        "sampler": {"name": "BatchSampler", "shuffle": False, "drop_last": True}},
        ppsci.loss.PeriodicMSELoss("mean", weight={"u_x": 0.5}), criteria=lambda t, x, y: np.isclose(x, -1.0), name="PBC")
    data = cst.data_loader.dataset.input
    assert data["x"].shape == (bs * iters, 1)
    h = bs // 2
    for i in range(iters):
        lo, hi = data["x"][i * bs:i * bs + h], data["x"][i * bs + h:(i + 1) * bs]
        assert np.all(lo == -1.0) and np.all(hi == 1.0)
        for k in ("t", "y"):
            assert np.array_equal(data[k][i * bs:i * bs + h], data[k][i * bs + h:(i + 1) * bs])
    assert np.all(data["normal_x"][:h] == -1.0) and np.all(data["normal_x"][h:bs] == 1.0)
    with pytest.raises(ValueError, match="even"):
        ppsci.constraint.PeriodicConstraint(exprs, {"u": 0}, geom, "x", {"dataset": "IterableNamedArrayDataset", "batch_size": 7,
                                                                          "iters_per_epoch": 1}, ppsci.loss.PeriodicMSELoss())

    opt = ppsci.optimizer.Adam(learning_rate=1e-3)(model)
    solver = ppsci.solver.Solver(model, {"PBC": cst}, str(tmp_path), opt, epochs=1, iters_per_epoch=iters, log_freq=1)
    solver.train()
    omodel = R.MLP(("t", "x", "y"), ("u",), net.astype(np.float32).astype(np.float64))
    oexprs = {"u": lambda d: d["u"], "u_x": R.lambdify(u.diff(x), omodel)}
    p = np.concatenate([q.detach().numpy().ravel() for q in omodel.parameters()])
    adam = R.Adam(p.size, 1e-3)
    for i in range(iters):
        off = 0
        with torch.no_grad():
            for q in omodel.parameters():
                q.copy_(torch.tensor(p[off:off + q.numel()].reshape(q.shape)))
                off += q.numel()
        oc = dict(name="PBC", input={k: data[k][i * bs:(i + 1) * bs].astype(np.float64) for k in ("t", "x", "y")}, exprs=oexprs,
                  label={k: np.zeros((bs, 1)) for k in ("u", "u_x")}, reduction="mean", loss_kind="periodic_mse",
                  loss_weight={"u_x": 0.5})
        total, losses, g, _ = R.loss_and_grads(omodel, [oc])
        p = adam.step(p, g)
    got = solver._compiled["PBC"].fused.losses()
    for k in ("u", "u_x"):
        assert got[k] == pytest.approx(losses[k], rel=5e-4), k
    assert rel(model.flat_params.cpu().numpy()[:p.size], p) < 1e-5


def test_eager_fallback_for_expressions_the_tracer_cannot_lower(tmp_path):
    """examples/euler_beam/euler_beam.py with a twist: the PDE u_xxxx + 1 = 0 runs on the fused kernels (fourth-order
    streams); its boundary terms pick ROWS of the batch, and `u0` picks the row by a data-dependent Python branch
    (`if float(d["x"][0]) == 0.0`) -- in the reference that is plain eager code on tensors.  (The name is historic: up to
    round 3 this constraint ran op by op through torch autograd.)  The boundary set is one fixed batch, so the trace is
    specialised to its values (graph.batch_values): the branch the reference takes on every step is traced, everything
    runs on the fused kernels, and nothing of the step is on an autograd tape.  Loss terms and the summed parameter
    gradient against the oracle's reverse-over-reverse restatement; a batch that changes every step is refused."""
    from ppsci.autodiff import hessian, jacobian

    model = ppsci.arch.MLP(("x",), ("u",), 3, 20, "tanh")
    net = T.make_net(1, [20, 20, 20], 1, bias_scale=0.1)
    set_model_weights(model, net)
    rng = np.random.default_rng(5)
    Xi = rng.uniform(0, 1, (32, 1)).astype(np.float32)
    Xb = np.asarray([[0.0], [0.0], [1.0], [1.0]], np.float32)
    eq = ppsci.equation.Biharmonic(1, -1.0, 1.0)
    pde = _sup_constraint({"x": Xi}, {"biharmonic": np.zeros((32, 1), np.float32)}, eq.equations, ppsci.loss.MSELoss("mean"), name="EQ")
    bc_exprs = {"u0": lambda d: d["u"][0:1] if float(d["x"][0]) == 0.0 else d["u"][3:4],
                "u__x": lambda d: jacobian(d["u"], d["x"])[1:2] if d["x"][1] < 0.5 else jacobian(d["u"], d["x"])[2:3],
                "u__x__x": lambda d: hessian(d["u"], d["x"])[2:3],
                "u__x__x__x": lambda d: jacobian(hessian(d["u"], d["x"]), d["x"])[3:4]}
    bc = _sup_constraint({"x": Xb}, {k: np.zeros((4, 1), np.float32) for k in bc_exprs}, bc_exprs, ppsci.loss.MSELoss("sum"),
                         name="BC")
    solver = _solver(tmp_path, model, {"EQ": pde, "BC": bc})
    cc = solver._compiled["BC"]
    assert cc.specialised_to == ["float(x[0:1]) = 0.0", "lt(x[1:2]) = True"] and not solver._compiled["EQ"].specialised_to
    assert cc._row_slices == {"u0": 0, "u__x": 1, "u__x__x": 2, "u__x__x__x": 3}
    p0 = model.flat_params.clone()
    solver.train()  # one Adam step
    assert torch.isfinite(model.flat_params).all() and not torch.equal(model.flat_params, p0)
    model.flat_params.copy_(p0)
    solver.engine.forward_backward([solver._compiled["EQ"].fused, cc.fused])
    g = solver.engine.grad.cpu().numpy().astype(np.float64)

    omodel = R.MLP(("x",), ("u",), net.astype(np.float32).astype(np.float64))

    def ograd(y, x):
        return torch.autograd.grad(y, x, torch.ones_like(y), create_graph=True)[0]

    def d1(d): return ograd(d["u"], d["x"])  # noqa: E704
    def d2(d): return ograd(d1(d), d["x"])  # noqa: E704
    def d3(d): return ograd(d2(d), d["x"])  # noqa: E704
    # the oracle runs the SAME branching code eagerly on tensors
    oc = [dict(name="EQ", input={"x": Xi.astype(np.float64)}, exprs={k: R.lambdify(e, omodel) for k, e in eq.equations.items()},
               label={"biharmonic": np.zeros((32, 1))}, reduction="mean"),
          dict(name="BC", input={"x": Xb.astype(np.float64)},
               exprs={"u0": lambda d: d["u"][0:1] if float(d["x"][0]) == 0.0 else d["u"][3:4],
                      "u__x": lambda d: d1(d)[1:2] if d["x"][1] < 0.5 else d1(d)[2:3], "u__x__x": lambda d: d2(d)[2:3],
                      "u__x__x__x": lambda d: d3(d)[3:4]},
               label={k: np.zeros((4, 1)) for k in bc_exprs}, reduction="sum")]
    total, losses, gref, _ = R.loss_and_grads(omodel, oc)
    mine = {**solver._compiled["EQ"].fused.losses(), **cc.fused.losses()}
    for k in losses:
        assert mine[k] == pytest.approx(losses[k], rel=2e-4, abs=1e-9), k
    assert rel(g, gref) < 2e-4

    # a constraint whose batch changes every iteration has no single answer: refused with the reason
    import ppsci.constraint as C
    moving = C.SupervisedConstraint(
        {"dataset": {"name": "NamedArrayDataset", "input": {"x": Xi}, "label": {"u0": np.zeros((32, 1), np.float32)}},
         "batch_size": 8, "sampler": {"name": "BatchSampler", "shuffle": True, "drop_last": True}},
        ppsci.loss.MSELoss("mean"), {"u0": bc_exprs["u0"]}, name="MOV")
    with pytest.raises(NotImplementedError, match="changes every iteration"):
        _solver(tmp_path, model, {"MOV": moving})
    # ... and so has a value that depends on the network
    netdep = _sup_constraint({"x": Xb}, {"u0": np.zeros((4, 1), np.float32)},
                             {"u0": lambda d: d["u"] if float(d["u"][0]) > 0 else -d["u"]}, ppsci.loss.MSELoss("sum"), name="ND")
    with pytest.raises(NotImplementedError, match="depends on the network"):
        _solver(tmp_path, model, {"ND": netdep})


@pytest.mark.parametrize("act", ["tanh", "sigmoid"])
def test_per_layer_widths(tmp_path, act):
    """MLP(hidden_size=(24, 16, 32)) (mlp.py:199-201): the kernels run the padded width 32, the trainable tensors keep the
    reference's shapes; loss, gradient (in the reference's parameter layout) and two Adam steps against the oracle.
    sigmoid: act(0) != 0 on the padded features -- they must still contribute nothing."""
    hidden = [24, 16, 32]
    model = ppsci.arch.MLP(("x", "y"), ("u",), None, tuple(hidden), act)
    assert [tuple(p.shape) for p in model.parameters()] == [(2, 24), (24,), (24, 16), (16,), (16, 32), (32,), (32, 1), (1,)]
    net = T.make_net(2, hidden, 1, activation=act, bias_scale=0.1)
    set_model_weights(model, net)
    N = 37
    X = np.random.default_rng(8).uniform(0, 1, (N, 2)).astype(np.float32)
    lab = np.random.default_rng(9).standard_normal((N, 1)).astype(np.float32) * 0.1
    eq = ppsci.equation.Laplace(2)
    cst = _sup_constraint({"x": X[:, :1], "y": X[:, 1:]}, {"laplace": lab}, eq.equations, ppsci.loss.MSELoss("mean"))
    solver = _solver(tmp_path, model, {"EQ": cst})
    solver._materialize()
    solver.engine.forward_backward([solver._compiled["EQ"].fused])
    g = solver._train_grad().cpu().numpy().astype(np.float64)
    omodel = R.MLP(("x", "y"), ("u",), net.astype(np.float32).astype(np.float64))
    oc = dict(name="EQ", input={"x": X[:, :1].astype(np.float64), "y": X[:, 1:].astype(np.float64)},
              exprs={k: R.lambdify(e, omodel) for k, e in R.laplace_exprs(2).items()}, label={"laplace": lab.astype(np.float64)},
              reduction="mean")
    total, losses, gref, _ = R.loss_and_grads(omodel, [oc])
    assert g.shape == gref.shape
    assert solver._compiled["EQ"].fused.losses()["laplace"] == pytest.approx(total, rel=5e-5)
    assert rel(g, gref) < 1e-4
    solver.epochs = 2
    solver.train()
    p = np.concatenate([q.detach().numpy().ravel() for q in omodel.parameters()])
    adam = R.Adam(p.size, 1e-3)
    for _ in range(2):
        off = 0
        with torch.no_grad():
            for q in omodel.parameters():
                q.copy_(torch.from_numpy(p[off:off + q.numel()].reshape(q.shape)))
                off += q.numel()
        _, _, gg, _ = R.loss_and_grads(omodel, [oc])
        p = adam.step(p, gg)
    assert rel(model.flat_params.cpu().numpy(), p) < 2e-5


@pytest.mark.parametrize("hidden,layers", [(20, 3), (50, 5), (100, 2)])
def test_unsteady_two_dimensional_stream_set(tmp_path, hidden, layers):
    """(t, x, y) inputs with first derivatives in all three and second derivatives in x and y only -- unsteady NavierStokes
    (cylinder2d_unsteady_Re100) and heat / Allen-Cahn in 2-D: the (3, 2) stream set (6 streams instead of the padded (3, 3)'s 7)
    on the single-wave kernels (width 20 / 50) and on the feature-split ones (width 100), against the fp64 oracle."""
    from paddlescience_amd import graph

    assert (3, 2, 0, 0) in graph._INSTANTIATED
    net = T.make_net(3, [hidden] * layers, 3, seed=21, bias_scale=0.05)
    model = ppsci.arch.MLP(("t", "x", "y"), ("u", "v", "p"), layers, hidden, "tanh")
    set_model_weights(model, net)
    eq = ppsci.equation.NavierStokes(0.02, 1.0, 2, True)
    N = 45
    X = np.random.default_rng(8).uniform([1, -1, -1], [2, 1, 1], (N, 3)).astype(np.float32)
    inp = {"t": X[:, :1], "x": X[:, 1:2], "y": X[:, 2:]}
    lab = {k: np.zeros((N, 1), np.float32) for k in eq.equations}
    cst = _sup_constraint(inp, lab, eq.equations, ppsci.loss.MSELoss("mean"))
    solver = _solver(tmp_path, model, {"EQ": cst})
    fused = solver._compiled["EQ"].fused
    assert (len(fused.streams.dirs), fused.streams.n2) == (3, 2)
    solver.engine.forward_backward([fused])
    om = R.MLP(("t", "x", "y"), ("u", "v", "p"), net.astype(np.float32).astype(np.float64))
    oc = dict(name="EQ", input={k: v.astype(np.float64) for k, v in inp.items()},
              exprs={k: R.lambdify(e, om) for k, e in R.navier_stokes_exprs(0.02, 1.0, 2, True).items()},
              label={k: np.zeros((N, 1)) for k in eq.equations}, reduction="mean")
    total, losses, gref, _ = R.loss_and_grads(om, [oc])
    got = fused.losses()
    for k in eq.equations:
        assert got[k] == pytest.approx(losses[k], rel=5e-5), k
    assert rel(solver.engine.grad.cpu().numpy(), gref) < 1e-4


@pytest.mark.parametrize("reduction", ["sum", "mean"])
def test_one_row_slices_are_lowered_to_the_fused_kernels(tmp_path, reduction):
    """`d["u"][0:1]`, `jacobian(...)[1:2]`, ... as whole output expressions (examples/euler_beam/euler_beam.py:49-54): the
    reference's loss broadcasts the [1, 1] value against the [n, 1] label and weight columns; that equals a per-point loss on the
    one row with the weighted mean label and the summed weight plus a constant, which is what the fused path runs.  Non-zero,
    non-constant labels and weights; loss terms and gradient against the oracle's literal restatement."""
    from ppsci.autodiff import hessian, jacobian

    model = ppsci.arch.MLP(("x",), ("u",), 3, 20, "tanh")
    net = T.make_net(1, [20, 20, 20], 1, bias_scale=0.1, seed=3)
    set_model_weights(model, net)
    rng = np.random.default_rng(9)
    Xb = np.asarray([[0.0], [0.3], [0.7], [1.0]], np.float32)
    bc_exprs = {"u0": lambda d: d["u"][0:1], "u__x": lambda d: jacobian(d["u"], d["x"])[1:2],
                "u__x__x": lambda d: hessian(d["u"], d["x"])[2:3],
                "u__x__x__x": lambda d: jacobian(hessian(d["u"], d["x"]), d["x"])[3:4], "all": lambda d: d["u"][0:4]}
    lab = {k: rng.standard_normal((4, 1)).astype(np.float32) * 0.3 for k in bc_exprs}
    wts = {"u0": rng.uniform(0.5, 2.0, (4, 1)).astype(np.float32), "u__x__x": rng.uniform(0.5, 2.0, (4, 1)).astype(np.float32)}
    bc = _sup_constraint({"x": Xb}, lab, bc_exprs, ppsci.loss.MSELoss(reduction), weights=wts, name="BC")
    solver = _solver(tmp_path, model, {"BC": bc})
    cc = solver._compiled["BC"]
    assert not cc.specialised_to and cc._row_slices == {"u0": 0, "u__x": 1, "u__x__x": 2, "u__x__x__x": 3}
    solver.engine.forward_backward([cc.fused])
    g = solver.engine.grad.cpu().numpy().astype(np.float64)
    omodel = R.MLP(("x",), ("u",), net.astype(np.float32).astype(np.float64))

    def ograd(y, x):
        return torch.autograd.grad(y, x, torch.ones_like(y), create_graph=True)[0]

    def d1(d): return ograd(d["u"], d["x"])  # noqa: E704
    def d2(d): return ograd(d1(d), d["x"])  # noqa: E704
    def d3(d): return ograd(d2(d), d["x"])  # noqa: E704
    oc = [dict(name="BC", input={"x": Xb.astype(np.float64)},
               exprs={"u0": lambda d: d["u"][0:1], "u__x": lambda d: d1(d)[1:2], "u__x__x": lambda d: d2(d)[2:3],
                      "u__x__x__x": lambda d: d3(d)[3:4], "all": lambda d: d["u"][0:4]},
               label={k: v.astype(np.float64) for k, v in lab.items()}, weight={k: v.astype(np.float64) for k, v in wts.items()},
               reduction=reduction)]
    total, losses, gref, _ = R.loss_and_grads(omodel, oc)
    mine = cc.fused.losses()
    for k in losses:
        assert mine[k] == pytest.approx(losses[k], rel=2e-4, abs=1e-8), k
    assert rel(g, gref) < 2e-4
    if cc.fused.resid is not None:  # a sliced output is [1, 1] (the reference's `expr[a:a+1]`), a full one [n, 1]
        vals = cc.values()
        assert tuple(vals["u__x"].shape) == (1, 1) and tuple(vals["all"].shape) == (4, 1)
    # re-binding the same arrays takes the cached columns (no recomputation, no device-to-host copies on the step path)
    inp, labd, wd = {"x": Xb}, dict(lab), dict(wts)
    cc.bind(inp, labd, wd)
    first = cc._row_slice_cache
    cc.bind(inp, labd, wd)
    assert cc._row_slice_cache is first
    solver.engine.forward_backward([cc.fused])
    assert rel(solver.engine.grad.cpu().numpy().astype(np.float64), gref) < 2e-4
    # ... but a loader that refills the SAME buffers in place does not: the key is identity and content
    off0 = dict(cc.fused.loss_offsets)
    labd["u0"][:] = labd["u0"] + 1.0
    cc.bind(inp, labd, wd)
    assert cc._row_slice_cache is not first and cc.fused.loss_offsets["u0"] != off0["u0"]
    lab["u0"][:] = lab["u0"] - 1.0  # (labd holds the same arrays)
