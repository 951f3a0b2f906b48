"""ppsci.arch.PirateNet (layer-by-layer HIP path: csrc/pirate.hip + the MFMA 1x1-conv GEMMs) against tests/golden/piratenet.npz
-- produced by executing the REFERENCE's own PirateNet / PeriodEmbedding / FourierEmbedding / RandomWeightFactorization
(ppsci/arch/mlp.py:28-137, :530-820), autodiff/ad.py, utils/symbolic.py and loss/mse.py in float64 under the torch-backed
paddle shim (tests/golden/make_piratenet_golden.py): network outputs, per-point residuals (u_t, u_xx, products of two
outputs ...), loss terms and the gradient w.r.t. every named parameter incl. alpha, the Fourier kernel and the factorised
weights.  Tolerances: fp32 kernels against fp64 reference values -- residual rel-L2 <= 1e-5, gradient rel-L2 <= 1e-4
(BASELINE.md section 4).  Where the error comes from: the reference ALGORITHM itself evaluated in fp32 (torch CPU,
oracle/ref_torch.PirateNet(dtype=float32)) is 2.1e-6 / 7.5e-6 (residual / gradient) off the fp64 values on the `allen_cahn_rwf`
case -- the Fourier features make u_xx a sum of large cancelling terms -- and 1e-7 .. 1e-6 on the others; the HIP path is held
to within 3x of that fp32 floor on every case."""
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
            assert rel(g[off:off + k], ref) < 1e-4, n
        off += k
    assert rel(g, gref) < 1e-4
    res = solver.predict(inp, eqs, batch_size=None, return_numpy=True)
    for k in keys:
        assert rel(res[k][:, 0], GOLD[f"{name}/res/{k}"]) < 1e-5, k
    # ... and no worse than 3x the error of the reference algorithm itself in fp32 arithmetic (the conditioning of the case)
    from oracle import ref_torch as R

    state = {k.split("/", 2)[2]: GOLD[k] for k in GOLD.files if k.startswith(f"{name}/param/")}
    m32 = R.PirateNet(c["inputs"], c["outputs"], state, c["act"], c["periods"], dtype=torch.float32)
    c32 = dict(name="EQ", input={k: X[:, j:j + 1] for j, k in enumerate(c["inputs"])},
               exprs={k: R.lambdify(e, m32) for k, e in eqs.items()},
               label={k: GOLD[f"{name}/label/{k}"][:, None].astype(np.float32) for k in keys}, reduction=c["reduction"])
    _, _, g32, o32 = R.loss_and_grads(m32, [c32])
    if keys:
        e_res32 = max(rel(o32[0][k].detach().numpy()[:, 0], GOLD[f"{name}/res/{k}"]) for k in keys)
        e_res = max(rel(res[k][:, 0], GOLD[f"{name}/res/{k}"]) for k in keys)
        assert e_res < max(3 * e_res32, 1e-6), (e_res, e_res32)
    assert rel(g, gref) < max(3 * rel(g32, gref), 2e-6), (rel(g, gref), rel(g32, gref))


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_is_pinned_by_reference_run(name):
    """oracle/ref_torch.PirateNet (fp64 restatement) reproduces the reference-run values to ~1e-9."""
    from oracle import ref_torch as R

    c = CASES[name]
    state = {k.split("/", 2)[2]: GOLD[k] for k in GOLD.files if k.startswith(f"{name}/param/")}
    model = R.PirateNet(c["inputs"], c["outputs"], state, c["act"], c["periods"])
    X = GOLD[f"{name}/X"]
    keys = [k.split("/")[-1] for k in GOLD.files if k.startswith(f"{name}/res/")]
    cst = dict(name="EQ", input={k: X[:, j:j + 1] for j, k in enumerate(c["inputs"])},
               exprs={k: R.lambdify(e, model) for k, e in equations(c).items()},
               label={k: GOLD[f"{name}/label/{k}"][:, None] for k in keys}, reduction=c["reduction"])
    total, losses, g, outs = R.loss_and_grads(model, [cst])
    for k in keys:
        assert rel(outs[0][k].detach().numpy()[:, 0], GOLD[f"{name}/res/{k}"]) < 1e-9
        assert losses[k] == pytest.approx(float(GOLD[f"{name}/loss/{k}"]), rel=1e-9)
    gref = np.concatenate([GOLD[f"{name}/grad/{n}"].ravel() for n in state])
    assert rel(g, gref) < 1e-8


@pytest.mark.parametrize("hidden,blocks,n", [(256, 3, 300), (128, 2, 1000)])
def test_piratenet_full_width_matches_oracle(dev, tmp_path, hidden, blocks, n):
    """The reference configuration's width (allen_cahn_piratenet.yaml: 3 blocks x 256 -- 256 x 256 weights run through two
    output-channel slabs of the GEMM kernel) against the fp64 oracle pinned above: Allen-Cahn residual, loss, gradient."""
    import sympy as sp

    from oracle import ref_torch as R

    if dev == "emu":
        pytest.skip("full-width case: minutes on the CPU SIMT emulator; the GPU half is the parity evidence (the small cases "
                    "above cover the same code on the emulator)")
    np.random.seed(3)
    periods = {"x": (2.0, False)}
    model = ppsci.arch.PirateNet(("t", "x"), ("u",), blocks, hidden, "tanh", periods=periods,
                                 fourier={"dim": hidden, "scale": 2.0}, random_weight={"mean": 1.0, "std": 0.1})
    with torch.no_grad():
        for nm, v in model.named_parameters():
            if nm.endswith("alpha"):
                v.fill_(0.3)
            elif nm.endswith("bias"):
                v.copy_(torch.as_tensor(np.random.normal(0, 0.1, tuple(v.shape)).astype(np.float32)))
    state = {nm: v.detach().cpu().numpy().astype(np.float64) for nm, v in model.named_parameters()}
    X = np.random.default_rng(1).uniform([0, -1], [1, 1], (n, 2)).astype(np.float32)
    lab = np.random.default_rng(2).standard_normal((n, 1)).astype(np.float32) * 0.1
    t, x = sp.symbols("t x")
    u = sp.Function("u")(t, x)
    eqs = {"allen_cahn": u.diff(t) - 0.0001 * u.diff(x, 2) + 5 * u**3 - 5 * u}
    om = R.PirateNet(("t", "x"), ("u",), state, "tanh", periods)
    Xd = X.astype(np.float64)
    cst = dict(name="EQ", input={"t": Xd[:, :1], "x": Xd[:, 1:]}, exprs={k: R.lambdify(e, om) for k, e in eqs.items()},
               label={"allen_cahn": lab.astype(np.float64)}, reduction="mean")
    total, losses, gref, outs = R.loss_and_grads(om, [cst])
    inp = {"t": X[:, :1], "x": X[:, 1:]}
    cfg = {"dataset": {"name": "IterableNamedArrayDataset", "input": inp, "label": {"allen_cahn": lab}}}
    c = ppsci.constraint.SupervisedConstraint(cfg, ppsci.loss.MSELoss("mean"), eqs, name="EQ")
    solver = ppsci.solver.Solver(model, {"EQ": c}, str(tmp_path), ppsci.optimizer.Adam(1e-3)(model), epochs=1, iters_per_epoch=1)
    solver.engine.forward_backward([solver._compiled["EQ"].fused])
    assert solver._compiled["EQ"].fused.losses()["allen_cahn"] == pytest.approx(losses["allen_cahn"], rel=1e-4)
    assert rel(solver.engine.grad.cpu().numpy(), gref) < 1e-4
    res = solver.predict(inp, eqs, batch_size=None, return_numpy=True)
    assert rel(res["allen_cahn"][:, 0], outs[0]["allen_cahn"].detach().numpy()[:, 0]) < 1e-5


def test_model_list_of_an_mlp_and_a_piratenet(dev, tmp_path):
    """ppsci.arch.ModelList((MLP, PirateNet)): the members' outputs are coupled in one residual; both gradients against the fp64
    oracle (torch autograd over oracle MLP + oracle PirateNet)."""
    import sympy as sp

    from oracle import ref_torch as R
    from oracle import taylor_np as T
    from tests.common import set_model_weights

    c, pir = _model("two_out_gelu")  # (x, y) -> (u, v)
    net = T.make_net(2, [16, 16], 1, seed=9, bias_scale=0.05)
    mlp = ppsci.arch.MLP(("x", "y"), ("p",), 2, 16, "tanh")
    set_model_weights(mlp, net)
    model = ppsci.arch.ModelList((mlp, pir))
    assert model.output_keys == ("p", "u", "v")
    x, y = sp.symbols("x y")
    u, v, p = (sp.Function(k)(x, y) for k in ("u", "v", "p"))
    eqs = {"mx": u * u.diff(x) + v * u.diff(y) - 0.1 * (u.diff(x, 2) + u.diff(y, 2)) + p.diff(x), "div": u.diff(x) + v.diff(y) + p}
    N = 37
    X = np.random.default_rng(4).uniform(-1, 1, (N, 2)).astype(np.float32)
    inp = {"x": X[:, :1], "y": X[:, 1:]}
    lab = {k: np.zeros((N, 1), np.float32) for k in eqs}
    cfg = {"dataset": {"name": "IterableNamedArrayDataset", "input": inp, "label": lab}}
    cst = ppsci.constraint.SupervisedConstraint(cfg, ppsci.loss.MSELoss("mean"), eqs, name="EQ")
    solver = ppsci.solver.Solver(model, {"EQ": cst}, str(tmp_path), ppsci.optimizer.Adam(1e-3)(model), epochs=1, iters_per_epoch=1)
    solver.engine.forward_backward([solver._compiled["EQ"].fused])
    g = solver.engine.grad.cpu().numpy()
    state = {k.split("/", 2)[2]: GOLD[k] for k in GOLD.files if k.startswith("two_out_gelu/param/")}
    om = R.ModelList((R.MLP(("x", "y"), ("p",), net.astype(np.float32).astype(np.float64)),
                      R.PirateNet(c["inputs"], c["outputs"], state, c["act"], c["periods"])))
    oc = dict(name="EQ", input={k: v.astype(np.float64) for k, v in inp.items()}, exprs={k: R.lambdify(e, om) for k, e in eqs.items()},
              label={k: np.zeros((N, 1)) for k in eqs}, reduction="mean")
    total, losses, gref, _ = R.loss_and_grads(om, [oc])
    got = solver._compiled["EQ"].fused.losses()
    for k in eqs:
        assert got[k] == pytest.approx(losses[k], rel=1e-4)
    n_mlp = mlp.flat_params.numel()
    assert rel(g[mlp._param_offset:mlp._param_offset + n_mlp], gref[:n_mlp]) < 1e-4
    n_p = pir.flat_params.numel()
    assert rel(g[pir._param_offset:pir._param_offset + n_p], gref[n_mlp:]) < 1e-4
    solver.train()


def test_piratenet_trains(dev, tmp_path):
    c, model = _model("three_blocks_silu")
    rng = np.random.default_rng(0)
    X = rng.uniform(c["lo"], c["hi"], (64, 3)).astype(np.float32)
    inp = {k: X[:, j:j + 1] for j, k in enumerate(c["inputs"])}
    cfg = {"dataset": {"name": "IterableNamedArrayDataset", "input": inp, "label": {"heat": np.zeros((64, 1), np.float32)}}}
    cst = ppsci.constraint.SupervisedConstraint(cfg, ppsci.loss.MSELoss("mean"), equations(c), name="EQ")
    solver = ppsci.solver.Solver(model, {"EQ": cst}, str(tmp_path), ppsci.optimizer.Adam(2e-3)(model), epochs=1,
                                 iters_per_epoch=12, log_freq=12)
    solver.engine.forward_backward([solver._compiled["EQ"].fused])
    l0 = solver._compiled["EQ"].fused.losses()["heat"]
    solver.train()
    assert solver.last_losses["loss"] < 0.9 * l0
