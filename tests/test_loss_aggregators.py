"""GradNorm / NTK (/root/reference/ppsci/loss/mtl/grad_norm.py, ntk.py) through the Solver: weights
after the update steps and parameters after three Adam steps against the oracle (per-term gradients by autograd
through the restated reference algorithm, then the documented weight formulas)."""
import numpy as np
import pytest
import torch

import ppsci
from oracle import ref_torch as R
from oracle import taylor_np as T
from tests.common import make_dev_fixture, set_model_weights

dev = make_dev_fixture()


def _setup(tmp_path, agg_factory):
    model = ppsci.arch.MLP(("x", "y"), ("u",), 2, 16, "tanh")
    net = T.make_net(2, [16, 16], 1, bias_scale=0.1)
    set_model_weights(model, net)
    N = 40
    rng = np.random.default_rng(5)
    X = rng.uniform(0, 1, (N, 2)).astype(np.float32)
    lab_u = (np.cos(X[:, :1]) * np.cosh(X[:, 1:])).astype(np.float32)
    eq = ppsci.equation.Laplace(dim=2)
    cfg = {"dataset": {"name": "IterableNamedArrayDataset", "input": {"x": X[:, :1], "y": X[:, 1:]},
                       "label": {"laplace": np.zeros((N, 1), np.float32), "u": lab_u}}}
    cst = ppsci.constraint.SupervisedConstraint(cfg, ppsci.loss.MSELoss("mean"),
                                                {**eq.equations, "u": lambda out: out["u"]}, name="EQ")
    opt = ppsci.optimizer.Adam(1e-3)(model)
    agg = agg_factory(model)
    solver = ppsci.solver.Solver(model, {"EQ": cst}, str(tmp_path), opt, epochs=3, iters_per_epoch=1, log_freq=1,
                                 loss_aggregator=agg)
    return solver, model, net, X, lab_u, agg


def _oracle(net, X, lab_u, kind, steps=3, momentum=0.9, update_freq=2):
    omodel = R.MLP(("x", "y"), ("u",), net.astype(np.float32).astype(np.float64))
    params = list(omodel.parameters())
    flat = np.concatenate([p.detach().numpy().ravel() for p in params])
    adam = R.Adam(flat.size, 1e-3)
    w = np.ones(2)
    x = {"x": torch.tensor(X[:, :1].astype(np.float64), requires_grad=True),
         "y": torch.tensor(X[:, 1:].astype(np.float64), requires_grad=True)}
    for step in range(steps):
        out = omodel(x)
        ux = torch.autograd.grad(out["u"].sum(), x["x"], create_graph=True)[0]
        uy = torch.autograd.grad(out["u"].sum(), x["y"], create_graph=True)[0]
        lap = torch.autograd.grad(ux.sum(), x["x"], create_graph=True)[0] + torch.autograd.grad(uy.sum(), x["y"], create_graph=True)[0]
        terms = [(lap ** 2).mean(), ((out["u"] - torch.tensor(lab_u.astype(np.float64))) ** 2).mean()]
        gs = [np.concatenate([(torch.zeros_like(p) if g is None else g).numpy().ravel() for g, p in
                              zip(torch.autograd.grad(t, params, retain_graph=True, allow_unused=True), params)]) for t in terms]
        total_g = w[0] * gs[0] + w[1] * gs[1]
        if step % update_freq == 0:
            norms = np.array([np.linalg.norm(g) for g in gs])
            if kind == "gradnorm":
                w = momentum * w + (1 - momentum) * norms.mean() / norms
            else:
                w = norms.sum() / norms
        flat = adam.step(flat, total_g)
        off = 0
        with torch.no_grad():
            for p in params:
                k = p.numel()
                p.copy_(torch.from_numpy(flat[off:off + k].reshape(p.shape)))
                off += k
    return flat, w


@pytest.mark.parametrize("kind", ["gradnorm", "ntk"])
def test_grad_weighted_aggregators_match_oracle(kind, dev, tmp_path):
    fac = (lambda m: ppsci.loss.mtl.GradNorm(m, 2, update_freq=2, momentum=0.9)) if kind == "gradnorm" else \
          (lambda m: ppsci.loss.mtl.NTK(m, 2, update_freq=2))
    solver, model, net, X, lab_u, agg = _setup(tmp_path, fac)
    solver.train()
    flat, w = _oracle(net, X, lab_u, kind)
    np.testing.assert_allclose(agg.weight, w, rtol=2e-4)
    np.testing.assert_allclose(model.flat_params.cpu().numpy(), flat, rtol=0, atol=3e-5)
    assert set(solver.last_losses) == {"loss", "EQ"} and np.isfinite(solver.last_losses["loss"])
