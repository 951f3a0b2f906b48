"""Pins the two CPU oracles against each other and against the reference's stored known answers.

* oracle/ref_torch.py  : reverse-over-reverse restatement of the reference algorithm
* oracle/taylor_np.py  : closed-form Taylor-mode streams (what the HIP kernels implement)

Known answers from the reference tree:
  MSELoss   /root/reference/ppsci/loss/mse.py:46-68
  NS print  /root/reference/ppsci/equation/pde/base.py:99-105
"""
import numpy as np
import pytest
import sympy as sp
import torch

from oracle import ref_torch as R
from oracle import taylor_np as T


def _pts(n, d, seed=42, lo=-1.0, hi=1.0):
    return np.random.default_rng(seed).uniform(lo, hi, size=(n, d))


def test_mse_known_answers():
    out = {"u": torch.tensor([[0.5, 0.9], [1.1, -1.3]]), "v": torch.tensor([[0.5, 0.9], [1.1, -1.3]])}
    lab = {"u": torch.tensor([[-1.8, 1.0], [-0.2, 2.5]]), "v": torch.tensor([[0.1, 0.1], [0.1, 0.1]])}
    w = {"u": 0.8, "v": 0.2}
    m = R.mse_loss(out, lab, None, "mean", w)
    assert float(m["u"]) == pytest.approx(4.28600025, rel=1e-6)
    assert float(m["v"]) == pytest.approx(0.18800001, rel=1e-6)
    s = R.mse_loss(out, lab, None, "sum", w)
    assert float(s["u"]) == pytest.approx(17.14400101, rel=1e-6)
    assert float(s["v"]) == pytest.approx(0.75200003, rel=1e-6)


def test_ns_expression_strings():
    e = R.navier_stokes_exprs(1.0, 1.0, 2, False)
    assert str(e["continuity"]) == "Derivative(u(x, y), x) + Derivative(v(x, y), y)"
    assert str(e["momentum_x"]) == (
        "u(x, y)*Derivative(u(x, y), x) + v(x, y)*Derivative(u(x, y), y) + 1.0*Derivative(p(x, y), x)"
        " - 1.0*Derivative(u(x, y), (x, 2)) - 1.0*Derivative(u(x, y), (y, 2))"
    )


@pytest.mark.parametrize("act", ["tanh", "silu", "sin", "sigmoid", "cos", "gelu", "siren"])
def test_laplace2d_streams_match_reverse_over_reverse(act):
    net = T.make_net(2, [20, 20, 20], 1, activation=act, bias_scale=0.1)
    X = _pts(37, 2)
    model = R.MLP(("x", "y"), ("u",), net)
    fn = R.lambdify(R.laplace_exprs(2)["laplace"], model)
    data = {k: torch.tensor(X[:, i : i + 1], requires_grad=True) for i, k in enumerate(("x", "y"))}
    res = fn(data).detach().numpy()[:, 0]
    R.clear()
    U = T.taylor_forward(net, X, np.eye(2), 2)
    np.testing.assert_allclose(U[0, 3] + U[0, 4], res, rtol=1e-9, atol=1e-10)
    x, y = sp.symbols("x y")
    fx = R.lambdify(sp.Function("u")(x, y).diff(y), model)
    data = {k: torch.tensor(X[:, i : i + 1], requires_grad=True) for i, k in enumerate(("x", "y"))}
    np.testing.assert_allclose(U[0, 2], fx(data).detach().numpy()[:, 0], rtol=1e-10, atol=1e-12)
    R.clear()


def test_allen_cahn_with_period_embedding():
    w = 2 * np.pi / 2.0
    net = T.make_net(2, [16, 16, 16], 1, periods={1: w}, bias_scale=0.1)
    X = _pts(29, 2)
    model = R.MLP(("t", "x"), ("u",), net)
    data = {k: torch.tensor(X[:, i : i + 1], requires_grad=True) for i, k in enumerate(("t", "x"))}
    data.update(model(data))
    res = R.allen_cahn_fn(0.01)(data).detach().numpy()[:, 0]
    R.clear()
    # streams: dirs = (x, t) so the single second-order stream is along x
    U = T.taylor_forward(net, X, np.array([[0.0, 1.0], [1.0, 0.0]]), 1)[0]
    u, u_x, u_t, u_xx = U
    mine = u_t - (0.01**2) * u_xx + 5 * u * u * u - 5 * u
    np.testing.assert_allclose(mine, res, rtol=1e-10, atol=1e-12)


def test_skip_connection_quirk():
    net = T.make_net(2, [8, 8, 8, 8, 8], 1, skip_connection=True, bias_scale=0.1)
    X = _pts(11, 2)
    model = R.MLP(("x", "y"), ("u",), net)
    fn = R.lambdify(R.laplace_exprs(2)["laplace"], model)
    data = {k: torch.tensor(X[:, i : i + 1], requires_grad=True) for i, k in enumerate(("x", "y"))}
    res = fn(data).detach().numpy()[:, 0]
    R.clear()
    U = T.taylor_forward(net, X, np.eye(2), 2)
    np.testing.assert_allclose(U[0, 3] + U[0, 4], res, rtol=1e-10, atol=1e-12)


def _ns_constraint(X, nu=0.01, rho=1.0):
    ex = R.navier_stokes_exprs(nu, rho, 2, False)
    return ex


def test_ns2d_loss_and_param_grads_match():
    """Taylor forward + epilogue adjoint + Taylor backward == autograd through the double-backward graph."""
    net = T.make_net(2, [12, 12, 12], 3, bias_scale=0.1)
    N = 23
    X = _pts(N, 2, lo=-0.05, hi=0.05)
    model = R.MLP(("x", "y"), ("u", "v", "p"), net)
    ex = R.navier_stokes_exprs(0.01, 1.0, 2, False)
    exprs = {k: R.lambdify(e, model) for k, e in ex.items()}
    cst = dict(
        name="EQ", input={"x": X[:, :1], "y": X[:, 1:]}, exprs=exprs,
        label={k: np.zeros((N, 1)) for k in ex}, weight={k: np.full((N, 1), 1e-4) for k in ex}, reduction="sum",
    )
    total, losses, gref, outs = R.loss_and_grads(model, [cst])

    U, cache = T.taylor_forward(net, X, np.eye(2), 2, keep=True)
    (u, ux, uy, uxx, uyy), (v, vx, vy, vxx, vyy), (p, px, py, _, _) = U
    nu = float(np.float32(0.01))  # ConstantNode stores fp32 (symbolic.py:448-454)
    cont = ux + vy
    mx = -nu * uxx + -nu * uyy + u * ux + v * uy + px
    my = -nu * vxx + -nu * vyy + u * vx + v * vy + py
    np.testing.assert_allclose(mx, outs[0]["momentum_x"].detach().numpy()[:, 0], rtol=1e-9, atol=1e-13)
    w = 1e-4
    loss = w * ((cont**2).sum() + (mx**2).sum() + (my**2).sum())
    assert loss == pytest.approx(total, rel=1e-10)
    # adjoints of the residuals, then of the streams
    rc, rx, ry = 2 * w * cont, 2 * w * mx, 2 * w * my
    Ub = np.zeros_like(U)
    Ub[0, 0] = rx * ux + ry * vx
    Ub[0, 1] = rc + rx * u
    Ub[0, 2] = rx * v
    Ub[0, 3] = -nu * rx
    Ub[0, 4] = -nu * rx
    Ub[1, 0] = rx * uy + ry * vy
    Ub[1, 1] = ry * u
    Ub[1, 2] = rc + ry * v
    Ub[1, 3] = -nu * ry
    Ub[1, 4] = -nu * ry
    Ub[2, 1] = rx
    Ub[2, 2] = ry
    gW, gb = T.taylor_backward(net, cache, Ub)
    np.testing.assert_allclose(T.flat_grads(gW, gb), gref, rtol=1e-8, atol=1e-14)


def test_laplace_last_bias_has_no_gradient():
    """pure second-derivative residual gives d loss / d b_last == 0."""
    net = T.make_net(2, [10, 10], 1, bias_scale=0.1)
    X = _pts(9, 2)
    model = R.MLP(("x", "y"), ("u",), net)
    fn = R.lambdify(R.laplace_exprs(2)["laplace"], model)
    cst = dict(input={"x": X[:, :1], "y": X[:, 1:]}, exprs={"laplace": fn}, label={"laplace": np.zeros((9, 1))}, reduction="sum")
    _, _, g, _ = R.loss_and_grads(model, [cst])
    assert g[-1] == 0.0


def test_adam_matches_torch_adam():
    rng = np.random.default_rng(0)
    p0 = rng.standard_normal(50)
    p = torch.tensor(p0.copy(), requires_grad=True)
    opt = torch.optim.Adam([p], lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
    mine = R.Adam(50, 1e-3)
    q = p0.copy()
    for _ in range(7):
        g = rng.standard_normal(50)
        p.grad = torch.tensor(g.copy())
        opt.step()
        q = mine.step(q, g)
    np.testing.assert_allclose(q, p.detach().numpy(), rtol=1e-9, atol=1e-12)


def test_fp32_restatement_close_to_fp64():
    """Sizes the tolerance: fp32 reference-like vs fp64 truth, Allen-Cahn 4x64."""
    net = T.make_net(2, [64] * 4, 1)
    X = np.random.default_rng(42).uniform([0, -1], [1, 1], (2000, 2))
    out = {}
    for dt, npdt in ((torch.float64, np.float64), (torch.float32, np.float32)):
        model = R.MLP(("t", "x"), ("u",), net, dtype=dt)
        data = {k: torch.tensor(X[:, i : i + 1].astype(npdt), requires_grad=True) for i, k in enumerate(("t", "x"))}
        data.update(model(data))
        out[dt] = R.allen_cahn_fn(0.01)(data).detach().numpy().astype(np.float64)[:, 0]
        R.clear()
    rel = np.linalg.norm(out[torch.float32] - out[torch.float64]) / np.linalg.norm(out[torch.float64])
    assert rel < 5e-6
