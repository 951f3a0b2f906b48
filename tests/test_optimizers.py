"""Row a26 beyond Adam: SGD / Momentum (+nesterov) / RMSProp (+centered) / AdamW fused updates against the oracle's
restatement of the documented paddle.optimizer rules (oracle/ref_torch.FirstOrder), L-BFGS through the Solver on a
small Laplace problem, and the closed forms of the remaining LR schedules (paddle.optimizer.lr semantics as wrapped
by /root/reference/ppsci/optimizer/lr_scheduler.py)."""
import math

import numpy as np
import pytest
import torch

import ppsci
from oracle import ref_torch as R
from tests.common import make_dev_fixture

dev = make_dev_fixture()

CASES = [
    ("sgd", lambda lr: ppsci.optimizer.SGD(lr, weight_decay=0.01), dict(weight_decay=0.01)),
    ("momentum", lambda lr: ppsci.optimizer.Momentum(lr, 0.9), dict(momentum=0.9)),
    ("momentum", lambda lr: ppsci.optimizer.Momentum(lr, 0.8, weight_decay=1e-3, use_nesterov=True),
     dict(momentum=0.8, weight_decay=1e-3, use_nesterov=True)),
    ("rmsprop", lambda lr: ppsci.optimizer.RMSProp(lr), dict()),
    ("rmsprop", lambda lr: ppsci.optimizer.RMSProp(lr, rho=0.9, epsilon=1e-5, momentum=0.5, centered=True),
     dict(rho=0.9, epsilon=1e-5, momentum=0.5, centered=True)),
    ("adamw", lambda lr: ppsci.optimizer.AdamW(lr, weight_decay=0.05), dict(weight_decay=0.05)),
]


@pytest.mark.parametrize("kind,factory,kw", CASES)
def test_fused_first_order_updates_match_oracle(kind, factory, kw, dev):
    model = ppsci.arch.MLP(("x", "y"), ("u",), 2, 16, "tanh")
    rng = np.random.default_rng(3)
    n = model.flat_params.numel()
    p0 = rng.standard_normal(n).astype(np.float32)
    model.flat_params.copy_(torch.from_numpy(p0).to(model.flat_params.device))
    opt = factory(3e-3)(model)
    ref = R.FirstOrder(kind, n, lr=3e-3, **kw)
    p = p0.astype(np.float64)
    for step in range(4):
        g = rng.standard_normal(n).astype(np.float32)
        opt.step(torch.from_numpy(g).to(model.flat_params.device))
        p = ref.step(p, g.astype(np.float64))
    np.testing.assert_allclose(model.flat_params.cpu().numpy(), p, rtol=2e-5, atol=2e-6)


def test_lbfgs_through_solver_reduces_laplace_loss(dev, tmp_path):
    np.random.seed(1)
    model = ppsci.arch.MLP(("x", "y"), ("u",), 2, 16, "tanh")
    eq = ppsci.equation.Laplace(dim=2)
    N = 64
    X = np.random.default_rng(0).uniform(0, 1, (N, 2)).astype(np.float32)
    lab = (np.cos(X[:, :1]) * np.cosh(X[:, 1:])).astype(np.float32)
    cfg = {"dataset": {"name": "IterableNamedArrayDataset", "input": {"x": X[:, :1], "y": X[:, 1:]},
                       "label": {"laplace": np.zeros((N, 1), np.float32), "u": lab}}}
    cst = ppsci.constraint.SupervisedConstraint(cfg, ppsci.loss.MSELoss("mean"),
                                                {**eq.equations, "u": lambda out: out["u"]}, name="EQ")
    opt = ppsci.optimizer.LBFGS(max_iter=8)(model)
    solver = ppsci.solver.Solver(model, {"EQ": cst}, str(tmp_path), opt, epochs=1, iters_per_epoch=1, log_freq=1)
    solver.engine.forward_backward([c.fused for c in solver._compiled.values()])
    solver._update_train_loss()
    l0 = solver.last_losses["loss"]
    solver.train()
    assert solver.last_losses["loss"] < 0.5 * l0
    assert torch.isfinite(model.flat_params).all()


def test_remaining_lr_schedules_closed_forms():
    S = ppsci.optimizer.lr_scheduler
    lin = S.Linear(10, 2, 0.1, end_lr=0.01, power=2.0)()
    vals = []
    for _ in range(22):
        vals.append(lin())
        lin.step()
    assert vals[0] == pytest.approx(0.1) and vals[10] == pytest.approx((0.1 - 0.01) * (1 - 10 / 20) ** 2 + 0.01)
    assert vals[21] == pytest.approx(0.01)
    ms = S.MultiStepDecay(10, 1, 1.0, milestones=(2, 5), gamma=0.1, by_epoch=True)()
    seq = []
    for _ in range(7):
        seq.append(ms())
        ms.step()
    assert seq == pytest.approx([1, 1, 0.1, 0.1, 0.1, 0.01, 0.01])
    cw = S.CosineWarmRestarts(20, 1, 1.0, T_0=2, T_mult=2, eta_min=0.0, by_epoch=True)()
    seq = []
    for _ in range(7):
        seq.append(cw())
        cw.step()
    exp = [1.0, 0.5, 1.0, (1 + math.cos(math.pi / 4)) / 2, 0.5, (1 + math.cos(3 * math.pi / 4)) / 2, 1.0]
    assert seq == pytest.approx(exp)
    oc = S.OneCycleLR(10, 1, 1.0, divide_factor=10.0, end_learning_rate=0.01, phase_pct=0.3, anneal_strategy="linear",
                      by_epoch=True)()
    seq = []
    for _ in range(10):
        seq.append(oc())
        oc.step()
    assert seq[0] == pytest.approx(0.1) and seq[2] == pytest.approx(1.0) and seq[9] == pytest.approx(0.01)
    assert all(a <= b + 1e-12 for a, b in zip(seq[:2], seq[1:3])) and all(a >= b - 1e-12 for a, b in zip(seq[2:9], seq[3:10]))
    lam = S.LambdaDecay(10, 1, 0.5, lambda t: 0.9 ** t, by_epoch=True)()
    lam.step()
    lam.step()
    assert lam() == pytest.approx(0.5 * 0.81)
    lst = S.SchedulerList((S.ConstLR(1, 1, 0.3)(), S.ConstLR(1, 1, 0.7)()))
    lst.step()
    assert lst.get_lr() == pytest.approx(0.3)


def test_adam_weight_decay_grad_clip_and_optimizer_list(dev):
    """Adam(weight_decay=c) is paddle's L2Decay (g += c p before the moments); grad_clip follows paddle.nn.ClipGradBy*;
    OptimizerList steps every member of a ModelList with its own optimizer on its slice of the flat gradient."""
    import ppsci
    from ppsci.optimizer import ClipGradByGlobalNorm, ClipGradByNorm, ClipGradByValue

    rng = np.random.default_rng(0)

    def adam_ref(p, g, lr, t=1, b1=0.9, b2=0.999, eps=1e-8):
        m, v = (1 - b1) * g, (1 - b2) * g * g
        return p - lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t) * m / (np.sqrt(v) + eps * np.sqrt(1 - b2 ** t))

    model = ppsci.arch.MLP(("x",), ("u",), 2, 16, "tanh")
    p0 = model.flat_params.cpu().numpy().astype(np.float64)
    g = rng.standard_normal(p0.size).astype(np.float32)
    gt = torch.tensor(g).to(model.flat_params.device)
    opt = ppsci.optimizer.Adam(1e-2, weight_decay=0.1)(model)
    opt.step(gt)
    np.testing.assert_allclose(model.flat_params.cpu().numpy(), adam_ref(p0, g + 0.1 * p0, 1e-2), rtol=2e-5, atol=1e-7)

    for clip, fn in [(ClipGradByValue(0.3), lambda a: np.clip(a, -0.3, 0.3)),
                     (ClipGradByGlobalNorm(0.5), lambda a: a * 0.5 / max(np.linalg.norm(a), 0.5))]:
        model = ppsci.arch.MLP(("x",), ("u",), 2, 16, "tanh")
        p0 = model.flat_params.cpu().numpy().astype(np.float64)
        ppsci.optimizer.Adam(1e-2, grad_clip=clip)(model).step(gt)
        np.testing.assert_allclose(model.flat_params.cpu().numpy(), adam_ref(p0, fn(g.astype(np.float64)), 1e-2),
                                   rtol=2e-5, atol=1e-7)
        assert np.array_equal(gt.cpu().numpy(), g)  # the caller's gradient buffer is left alone
    model = ppsci.arch.MLP(("x",), ("u",), 2, 16, "tanh")
    p0 = model.flat_params.cpu().numpy().astype(np.float64)
    ppsci.optimizer.SGD(0.5, grad_clip=ClipGradByNorm(0.1))(model).step(gt)
    want, off = p0.copy(), 0
    for q in model.parameters():
        n = q.numel()
        gq = g[off:off + n].astype(np.float64)
        want[off:off + n] -= 0.5 * gq * min(1.0, 0.1 / np.linalg.norm(gq))
        off += n
    np.testing.assert_allclose(model.flat_params.cpu().numpy(), want, rtol=2e-5, atol=1e-7)

    m1 = ppsci.arch.MLP(("x",), ("u",), 2, 16, "tanh")
    m2 = ppsci.arch.MLP(("x",), ("v",), 1, 8, "tanh")
    o1, o2 = ppsci.optimizer.SGD(0.1)(m1), ppsci.optimizer.SGD(1.0)(m2)
    both = ppsci.arch.ModelList((m1, m2))
    opts = ppsci.optimizer.OptimizerList((o1, o2))
    assert len(opts) == 2 and opts[1] is o2 and opts.get_lr() == 0.1
    gb = torch.ones_like(both.flat_params)
    a0, b0 = m1.flat_params.clone(), m2.flat_params.clone()
    opts.step(gb)
    np.testing.assert_allclose((a0 - m1.flat_params).cpu().numpy(), 0.1, rtol=1e-5)
    np.testing.assert_allclose((b0 - m2.flat_params).cpu().numpy(), 1.0, rtol=1e-5)
    with pytest.raises(ValueError):
        ppsci.optimizer.OptimizerList((o1, ppsci.optimizer.LBFGS()(m2)))
