"""ppsci.arch.MLP outside the fused kernels' envelope -> arch/layerwise_mlp.py (GEMM per layer on all Taylor streams + the
pointwise stream kernels): hidden width 300, a Fourier embedding with dim != hidden_size and a non-tanh activation, per-layer
widths with random weight factorisation -- each against the fp64 oracle restatement of the reference's MLP
(oracle/ref_torch.MLP, pinned by tests/golden/hotpath.npz / variants.npz), plus the dispatch rules."""
import numpy as np
import pytest
import sympy as sp
import torch

import ppsci
from oracle import ref_torch as R
from oracle import taylor_np as T
from tests.common import make_dev_fixture, rel

dev = make_dev_fixture()


def _oracle(model, cfg):
    sd = {n: v.detach().cpu().numpy().astype(np.float64) for n, v in model.named_parameters()}
    nl = len(model.widths)
    rwf = "linears.0.weight_v" in sd
    wkey = "weight_v" if rwf else "weight"
    names = [f"linears.{i}" for i in range(nl)] + ["last_fc"]
    periods = {model.input_keys.index(k): model.period_emb.freqs_dict[k] for k in (cfg.get("periods") or {})}
    net = T.NetSpec([sd[f"{n}.{wkey}"] for n in names], [sd[f"{n}.bias"] for n in names], activation=cfg["act"], periods=periods)
    return R.MLP(model.input_keys, model.output_keys, net, factor="random_weight" if rwf else None,
                 weight_g=[sd[f"{n}.weight_g"] for n in names] if rwf else None,
                 fourier_kernel=sd.get("fourier_emb.kernel"))


CASES = {
    "wide_300": dict(inputs=("x", "y"), outputs=("u",), layers=3, hidden=300, act="tanh", n=33),
    "fourier_dim_ne_hidden_silu": dict(inputs=("t", "x"), outputs=("u", "v"), layers=2, hidden=32, act="silu",
                                       fourier={"dim": 16, "scale": 1.0}, periods={"x": (2.0, False)}, n=40),
    "ragged_rwf": dict(inputs=("x", "y"), outputs=("u",), layers=None, hidden=(16, 24, 8), act="sin",
                       rwf={"mean": 0.5, "std": 0.1}, n=29),
}


@pytest.mark.parametrize("name", list(CASES))
def test_layerwise_mlp_matches_oracle(name, dev, tmp_path):
    c = CASES[name]
    if name == "wide_300" and dev == "emu":
        pytest.skip("width 300: minutes on the CPU SIMT emulator; the two narrow cases cover the same code there")
    ppsci.utils.misc.set_random_seed(7)
    model = ppsci.arch.MLP(c["inputs"], c["outputs"], c["layers"], c["hidden"], c["act"], periods=c.get("periods"),
                           fourier=c.get("fourier"), random_weight=c.get("rwf"))
    assert type(model).__name__ == "LayerwiseMLP"
    with torch.no_grad():
        for n_, v_ in model.named_parameters():
            if n_.endswith("bias"):
                v_.copy_(torch.as_tensor(np.random.normal(0, 0.1, tuple(v_.shape)).astype(np.float32)))
    om = _oracle(model, c)
    N = c["n"]
    X = np.random.default_rng(5).uniform(-1, 1, (N, len(c["inputs"]))).astype(np.float32)
    inp = {k: X[:, j:j + 1] for j, k in enumerate(c["inputs"])}
    syms = sp.symbols(" ".join(c["inputs"]))
    fs = [sp.Function(k)(*syms) for k in c["outputs"]]
    a, b = syms
    eqs = {"r": fs[0].diff(a) + 0.3 * fs[0].diff(a, 2) + fs[0].diff(b, 2) + fs[-1] * fs[0].diff(b)}
    lab = {"r": np.random.default_rng(6).standard_normal((N, 1)).astype(np.float32) * 0.1}
    cfg = {"dataset": {"name": "IterableNamedArrayDataset", "input": inp, "label": lab}}
    cst = ppsci.constraint.SupervisedConstraint(cfg, ppsci.loss.MSELoss("mean"), eqs, name="EQ")
    solver = ppsci.solver.Solver(model, {"EQ": cst}, str(tmp_path), ppsci.optimizer.Adam(1e-3)(model), epochs=2, iters_per_epoch=1)
    solver.engine.forward_backward([solver._compiled["EQ"].fused])
    oc = dict(name="EQ", input={k: v.astype(np.float64) for k, v in inp.items()}, exprs={k: R.lambdify(e, om) for k, e in eqs.items()},
              label={"r": lab["r"].astype(np.float64)}, reduction="mean")
    total, losses, gref, outs = R.loss_and_grads(om, [oc])
    assert solver._compiled["EQ"].fused.losses()["r"] == pytest.approx(losses["r"], rel=1e-4)
    assert rel(solver.engine.grad.cpu().numpy(), gref) < 1e-4
    res = solver.predict(inp, eqs, batch_size=None, return_numpy=True)
    assert rel(res["r"][:, 0], outs[0]["r"].detach().numpy()[:, 0]) < 1e-5
    out = model({k: torch.as_tensor(v) for k, v in inp.items()})
    ref = om({k: torch.tensor(v.astype(np.float64)) for k, v in inp.items()})
    for k in c["outputs"]:
        assert rel(out[k].cpu().numpy(), ref[k].detach().numpy()) < 5e-6
    before = model.flat_params.clone()
    solver.train()
    assert not torch.equal(before, model.flat_params)


def test_dispatch_rules(dev):
    mk = lambda *a, **k: type(ppsci.arch.MLP(("x", "y"), ("u",), *a, **k)).__name__  # noqa: E731
    assert mk(3, 64) == "MLP" and mk(4, 256) == "MLP" and mk(None, (20, 30, 20)) == "MLP"
    assert mk(2, 64, "tanh", fourier={"dim": 64, "scale": 1.0}) == "MLP"
    assert mk(2, 512) == "LayerwiseMLP"
    assert isinstance(ppsci.arch.MLP(("x", "y"), ("u",), 2, 512), ppsci.arch.MLP)  # ... and still an MLP to user code
    assert mk(2, 64, "tanh", fourier={"dim": 32, "scale": 1.0}) == "LayerwiseMLP"
    assert mk(2, 64, "gelu", fourier={"dim": 64, "scale": 1.0}) == "LayerwiseMLP"
    assert mk(None, (20, 30), random_weight={"mean": 0.5, "std": 0.1}) == "LayerwiseMLP"
    with pytest.raises(NotImplementedError):  # outside both paths: stays with the fused class, which says why
        ppsci.arch.MLP(("x", "y"), ("u",), 2, 64, "stan", fourier={"dim": 32, "scale": 1.0})
