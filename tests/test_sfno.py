"""ppsci.arch.SFNONet on the native executor (fno_engine.FnoNative with the spherical-harmonic transform pair of csrc/sht.hip) against
tests/golden/sfno.npz -- produced by the REFERENCE's own sfnonet.py / paddle_harmonics (tests/golden/make_sfno_golden.py, float64):

  * the transform pair itself: RealSHT / InverseRealSHT of random planes / coefficients, rel-L2 <= 2e-6; and the adjoint identities the
    backward pass rests on (<synthesis(Z), g> == <Z, analysis_B(g)>, <analysis(x), G> == <x, synthesis_A(G)>);
  * the tables of arch/sht_tables.py are what the reference's quadrature / Legendre code computes (through those transforms);
  * the network: output rel-L2 <= 2e-5, every parameter gradient <= 2e-4, loss <= 1e-5, bit-identical repeat."""
import os
import sys

import numpy as np
import pytest
import torch

from tests.common import make_dev_fixture, rel

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from sfno_cases import CASES  # noqa: E402

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "sfno.npz"))
dev = make_dev_fixture()
GRIDS = sorted({k.split("/")[1] for k in G.files if k.startswith("sht/")})


@pytest.mark.parametrize("grid", ["equiangular", "legendre-gauss", "lobatto"])
def test_quadrature_and_legendre_tables_are_the_references(grid):
    """arch/sht_tables.py against the reference's quadrature.py / legendre.py run as they are (numpy-only): nodes, weights (the rules are
    symmetric, so the reference's unflipped weights equal the flipped ones here) and the orthonormal associated Legendre functions."""
    from paddlescience_amd.arch import sht_tables

    for n in (9, 16, 33):
        theta, w = sht_tables.quadrature(grid, n)
        assert np.abs(theta - G[f"quad/{grid}/{n}/theta"]).max() < 1e-12
        assert np.abs(w - G[f"quad/{grid}/{n}/w"]).max() < 1e-13
        assert np.abs(sht_tables.legendre(6, n - 1, theta) - G[f"quad/{grid}/{n}/pct"]).max() < 1e-11


def _tables(H, W, L, M, d):
    from paddlescience_amd.arch import sht_tables

    tw, ta, tb = sht_tables.tables(H, W, L, M)
    f = dict(dtype=torch.float32, device=d)
    (a_an, a_sy), (b_an, b_sy) = (tuple(torch.tensor(t, **f) for t in sht_tables.kernel_layouts(T)) for T in (ta, tb))
    return torch.tensor(tw, **f), a_an, a_sy, b_an, b_sy


@pytest.mark.parametrize("grid", GRIDS)
def test_transform_pair_reproduces_the_reference_and_its_adjoints(grid, dev):
    from paddlescience_amd import _lib as L_
    from paddlescience_amd.device import get_device
    from paddlescience_amd.hotpath import _p, _stream_ptr

    hw, lm = grid.split("_")
    H, W = (int(v) for v in hw.split("x"))
    L, M = (int(v) for v in lm.split("x"))
    d = get_device()
    tw, a_an, a_sy, b_an, b_sy = _tables(H, W, L, M, d)
    x = torch.as_tensor(G[f"sht/{grid}/x"].astype(np.float32)).to(d).contiguous()
    n = x.shape[0]
    X = torch.empty((n, L, M, 2), dtype=torch.float32, device=d)
    L_.check(L_.lib().ppsci_sht_analysis(n, H, W, L, M, _p(tw), _p(a_an), _p(x), _p(X), _stream_ptr(X)))
    assert rel(X.cpu().numpy(), G[f"sht/{grid}/X"]) < 2e-6
    Z = torch.as_tensor(G[f"sht/{grid}/Z"].astype(np.float32)).to(d).contiguous()
    y = torch.empty((n, H, W), dtype=torch.float32, device=d)
    L_.check(L_.lib().ppsci_sht_synthesis(n, H, W, L, M, _p(tw), _p(b_sy), _p(Z), _p(y), _stream_ptr(y)))
    assert rel(y.cpu().numpy(), G[f"sht/{grid}/y"]) < 2e-6
    # adjoints: each kernel on the other transform's table
    rng = np.random.default_rng(H + W)
    g = torch.as_tensor(rng.standard_normal((n, H, W)).astype(np.float32)).to(d)
    gz = torch.empty_like(Z)
    L_.check(L_.lib().ppsci_sht_analysis(n, H, W, L, M, _p(tw), _p(b_an), _p(g), _p(gz), _stream_ptr(gz)))
    lhs, rhs = float((y.double() * g.double()).sum()), float((Z.double() * gz.double()).sum())
    assert abs(lhs - rhs) < 1e-5 * max(1.0, abs(lhs))
    Gc = torch.as_tensor(rng.standard_normal((n, L, M, 2)).astype(np.float32)).to(d)
    gx = torch.empty_like(x)
    L_.check(L_.lib().ppsci_sht_synthesis(n, H, W, L, M, _p(tw), _p(a_sy), _p(Gc), _p(gx), _stream_ptr(gx)))
    lhs, rhs = float((X.double() * Gc.double()).sum()), float((x.double() * gx.double()).sum())
    assert abs(lhs - rhs) < 1e-5 * max(1.0, abs(lhs))


@pytest.mark.parametrize("c", sorted(CASES))
def test_oracle_restatement_reproduces_the_reference_fixture(c):
    """oracle/ref_torch.sfno_forward (tables from scipy's Legendre functions and a moment solve, independent of arch/sht_tables.py)
    against the outputs and gradients of the REFERENCE's sfnonet.py: fp64 round-off."""
    from oracle import ref_torch as R

    k = CASES[c]
    P = {n[len(c) + 7:]: torch.tensor(G[n]).requires_grad_(True) for n in G.files if n.startswith(f"{c}/param/")}
    y = R.sfno_forward(torch.tensor(G[f"{c}/x"]), P, k["layers"], k["modes"], k["norm"])
    assert rel(y.detach().numpy(), G[f"{c}/y"]) < 1e-10
    loss = ((y - torch.tensor(G[f"{c}/target"])) ** 2).mean()
    names = sorted(P)
    for n, g in zip(names, torch.autograd.grad(loss, [P[n] for n in names])):
        assert rel(g.numpy(), G[f"{c}/grad/{n}"]) < 1e-8, n


def _model(c):
    import ppsci

    k = CASES[c]
    model = ppsci.arch.SFNONet(("x",), ("y",), k["modes"], k["hidden"], in_channels=3, out_channels=k.get("out", 1),
                               lifting_channels=k["lift"], projection_channels=k["proj"], n_layers=k["layers"], norm=k["norm"])
    P = {n[len(c) + 7:]: G[n].astype(np.float32) for n in G.files if n.startswith(f"{c}/param/")}
    assert set(P) == {n for n, _ in torch.nn.Module.named_parameters(model)}
    model.set_state_dict(P)
    return model


@pytest.mark.parametrize("c", sorted(CASES))
def test_native_path_reproduces_reference_sfno(c, dev):
    model = _model(c)
    d = model.flat_params.device
    x = torch.as_tensor(G[f"{c}/x"].astype(np.float32)).to(d)
    eng = model.native()
    y = eng.forward(x)
    assert eng.sht and not eng.kept
    assert rel(y.cpu().numpy(), G[f"{c}/y"]) < 2e-5
    tgt = torch.as_tensor(G[f"{c}/target"].astype(np.float32)).to(d)
    yl = y.detach().clone().requires_grad_(True)
    loss = ((yl - tgt) ** 2).mean()
    (gy,) = torch.autograd.grad(loss, yl)
    model.flat_grad.fill_(float("nan"))
    eng.backward(gy)
    assert abs(float(loss.detach()) - float(G[f"{c}/loss"])) < 1e-5 * float(G[f"{c}/loss"])
    assert torch.isfinite(model.flat_grad).all()
    for n, p in torch.nn.Module.named_parameters(model):
        assert rel(p.grad.cpu().numpy(), G[f"{c}/grad/{n}"]) < 2e-4, n
    g1 = model.flat_grad.clone()
    eng.forward(x)
    eng.backward(gy)
    assert torch.equal(g1, model.flat_grad)


def test_contraction_inside_the_synthesis_equals_the_two_launches(dev, monkeypatch):
    """ppsci_sht_synthesis_contract (forward and as the data gradient) against ppsci_sht_contract + ppsci_sht_synthesis: the same sums in
    the same order -- outputs and gradients bit-identical."""
    c = sorted(CASES)[0]
    res = []
    for fuse in ("1", "0"):
        monkeypatch.setenv("PPSCI_SHT_FUSE_CONTRACT", fuse)
        model = _model(c)
        d = model.flat_params.device
        x = torch.as_tensor(G[f"{c}/x"].astype(np.float32)).to(d)
        eng = model.native()
        y = eng.forward(x).clone()
        eng.backward(torch.ones_like(y) / y.numel())
        res.append((y.cpu().numpy(), model.flat_grad.cpu().numpy().copy()))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])


def test_sfnonet_trains_through_the_solver(dev, tmp_path):
    """SupervisedConstraint + Solver on an SFNONet at the reference example's grid (32 x 64), evaluated at 64 x 128 as its yaml does."""
    import ppsci

    torch.manual_seed(0)
    model = ppsci.arch.SFNONet(("x",), ("y",), (16, 16), 8, in_channels=1, out_channels=1, lifting_channels=12, projection_channels=12,
                               n_layers=2, norm="group_norm")
    rng = np.random.default_rng(1)
    x = rng.standard_normal((4, 1, 32, 64)).astype(np.float32)
    y = 0.3 * np.roll(x, 3, axis=-1)
    cst = ppsci.constraint.SupervisedConstraint(
        {"dataset": {"name": "NamedArrayDataset", "input": {"x": x}, "label": {"y": y}}, "batch_size": 4,
         "sampler": {"name": "BatchSampler", "drop_last": True, "shuffle": False}}, ppsci.loss.MSELoss("mean"), name="sup")
    opt = ppsci.optimizer.Adam(1e-2)(model)
    solver = ppsci.solver.Solver(model, {cst.name: cst}, str(tmp_path), opt, epochs=10, iters_per_epoch=1)

    def mse():
        out = solver.predict({"x": x}, batch_size=4)["y"]
        return float(((np.asarray(out.cpu() if hasattr(out, "cpu") else out) - y) ** 2).mean())

    before = mse()
    solver.train()
    assert mse() < 0.7 * before
    out = model({"x": torch.as_tensor(rng.standard_normal((2, 1, 64, 128)).astype(np.float32))})["y"]
    assert tuple(out.shape) == (2, 1, 64, 128) and bool(torch.isfinite(out).all())
