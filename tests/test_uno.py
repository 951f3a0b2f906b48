"""ppsci.arch.UNONet on the native executor (paddlescience_amd/uno_engine.py: the FNO block kernels + csrc/uno.hip) against

  * tests/golden/uno.npz -- produced by the REFERENCE's own unonet.py / fno_block.py (tests/golden/make_uno_golden.py, float64):
    output rel-L2 <= 2e-5, every parameter gradient rel-L2 <= 2e-4, loss rel <= 1e-5;
  * torch for the two kernels of csrc/uno.hip: bicubic resampling (F.interpolate) with its adjoint, and the spectrum crop / pad of
    irfftn(s=) with the Hermitian-weight correction of its way back (autograd through torch.fft)."""
import os
import sys

import numpy as np
import pytest
import torch

from tests.common import make_dev_fixture, rel

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from uno_cases import CASES  # noqa: E402

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "uno.npz"))
dev = make_dev_fixture()


def _model(c):
    import ppsci

    k = CASES[c]
    model = ppsci.arch.UNONet(("x",), ("y",), 3, 1, k["hidden"], lifting_channels=k["lift"], projection_channels=k["proj"],
                              n_layers=len(k["outs"]), uno_out_channels=k["outs"], uno_n_modes=k["modes"], uno_scalings=k["scal"],
                              horizontal_skips_map=k["skips"], norm=k["norm"], domain_padding=k["pad"],
                              domain_padding_mode=k["pad_mode"], fft_norm=k["fft_norm"])
    P = {n[len(c) + 7:]: G[n].astype(np.float32) for n in G.files if n.startswith(f"{c}/param/")}
    assert set(P) == {n for n, _ in torch.nn.Module.named_parameters(model)}
    model.set_state_dict(P)
    return model


@pytest.mark.parametrize("c", sorted(CASES))
def test_oracle_restatement_reproduces_the_reference_fixture(c):
    """oracle/ref_torch.uno_forward (the checker of bench.py's UNO entry and of the full-size test) against the outputs and gradients
    the REFERENCE's own unonet.py produced (tests/golden/uno.npz): fp64 round-off."""
    from oracle import ref_torch as R

    k = CASES[c]
    P = {n[len(c) + 7:]: torch.tensor(G[n]).requires_grad_(True) for n in G.files if n.startswith(f"{c}/param/")}
    y = R.uno_forward(torch.tensor(G[f"{c}/x"]), P, k["outs"], k["modes"], k["scal"], k["skips"], k["norm"], domain_padding=k["pad"],
                      domain_padding_mode=k["pad_mode"])
    assert rel(y.detach().numpy(), G[f"{c}/y"]) < 1e-12
    loss = ((y - torch.tensor(G[f"{c}/target"])) ** 2).mean()
    names = sorted(P)
    for n, g in zip(names, torch.autograd.grad(loss, [P[n] for n in names])):
        assert rel(g.numpy(), G[f"{c}/grad/{n}"]) < 1e-10, n


@pytest.mark.parametrize("full_fft", [False, True])
@pytest.mark.parametrize("c", sorted(CASES))
def test_native_path_reproduces_reference_uno(c, dev, full_fft, monkeypatch):
    """full_fft: library FFTs on whole spectra + ppsci_spectrum_resize (the path of planes too large for the kept-mode transforms);
    otherwise the kept-mode transforms between the two grids of a block (ppsci_dft2_kept_*_from)."""
    monkeypatch.setenv("PPSCI_FNO_FULL_FFT", "1" if full_fft else "0")
    model = _model(c)
    d = model.flat_params.device
    x = torch.as_tensor(G[f"{c}/x"].astype(np.float32)).to(d)
    eng = model.native()
    y = eng.forward(x)
    assert all(e["kept"] != full_fft for e in eng.blk)
    assert tuple(y.shape) == G[f"{c}/y"].shape
    assert rel(y.cpu().numpy(), G[f"{c}/y"]) < 2e-5
    tgt = torch.as_tensor(G[f"{c}/target"].astype(np.float32)).to(d)
    yl = y.detach().clone().requires_grad_(True)
    loss = ((yl - tgt) ** 2).mean()
    (gy,) = torch.autograd.grad(loss, yl)
    model.flat_grad.fill_(float("nan"))  # every entry must be written
    eng.backward(gy)
    assert abs(float(loss.detach()) - float(G[f"{c}/loss"])) < 1e-5 * float(G[f"{c}/loss"])
    assert torch.isfinite(model.flat_grad).all()
    for n, p in torch.nn.Module.named_parameters(model):
        assert rel(p.grad.cpu().numpy(), G[f"{c}/grad/{n}"]) < 2e-4, n
    g1 = model.flat_grad.clone()
    eng.forward(x)
    eng.backward(gy)
    assert torch.equal(g1, model.flat_grad)  # fixed-order sums: bit-identical


@pytest.mark.parametrize("shape", [(16, 16, 8, 8), (10, 10, 20, 20), (20, 20, 19, 19), (12, 20, 18, 10), (7, 5, 7, 9)])
def test_bicubic_resampling_and_its_adjoint(shape, dev):
    """ppsci_resample2d with uno_engine.bicubic_matrix == F.interpolate(mode="bicubic", align_corners=True) (fno_block.py:497-498);
    with the transposed matrices it is the exact adjoint: <A x, g> == <x, A^T g>."""
    from paddlescience_amd import uno_engine as U
    from paddlescience_amd.device import get_device

    H, W, H2, W2 = shape
    d = get_device()
    rng = np.random.default_rng(H * 100 + W2)
    x = torch.as_tensor(rng.standard_normal((3, 2, H, W)).astype(np.float32))
    want = torch.nn.functional.interpolate(x.double(), size=(H2, W2), mode="bicubic", align_corners=True)
    rs = U._Resampler(H, W, H2, W2, d)
    y = torch.empty((3, 2, H2, W2), dtype=torch.float32, device=d)
    rs.apply(6, x.to(d).contiguous(), y)
    assert rel(y.cpu().numpy(), want.numpy()) < 2e-6
    g = torch.as_tensor(rng.standard_normal((3, 2, H2, W2)).astype(np.float32)).to(d)
    gx = torch.full((3, 2, H, W), float("nan"), dtype=torch.float32, device=d)
    rs.adjoint(6, g, gx)
    lhs = float((y.double() * g.double()).sum())
    rhs = float((x.double().to(d) * gx.double()).sum())
    assert abs(lhs - rhs) < 1e-5 * max(1.0, abs(lhs))
    base = gx.clone()
    rs.adjoint(6, g, gx, accumulate=True)
    assert rel(gx.cpu().numpy(), 2 * base.cpu().numpy()) < 1e-6


@pytest.mark.parametrize("sizes", [(16, 16, 8, 8), (8, 8, 16, 16), (19, 19, 10, 10), (10, 10, 20, 20), (12, 20, 18, 10), (9, 7, 6, 12)])
def test_spectrum_resize_is_the_crop_of_irfftn_and_its_way_back(sizes, dev):
    """y = C2R_{H2 x W2}(resize(Z)) equals torch.fft.irfftn(Z, s=(H2, W2)) (unnormalised), and
    C2R_{H x W}-side weights: resize(R2C(g), w_num=W2, w_den=W) carries c_{W2}(j) / c_W(j), so that c_W(j) * that == what autograd
    through irfftn(s=) leaves on Z (its real and imaginary parts as independent variables)."""
    from paddlescience_amd import _lib as L
    from paddlescience_amd.device import get_device
    from paddlescience_amd.hotpath import _p, _stream_ptr

    H, W, H2, W2 = sizes
    d = get_device()
    rng = np.random.default_rng(sum(sizes))
    n, Wf, Wf2 = 3, W // 2 + 1, W2 // 2 + 1
    Z = torch.as_tensor(rng.standard_normal((n, H, Wf, 2)).astype(np.float32))
    Zc = torch.view_as_complex(Z.double().contiguous()).requires_grad_(True)
    y_ref = torch.fft.irfftn(Zc, s=(H2, W2), dim=(-2, -1), norm="forward")
    Zd = Z.to(d).contiguous()
    Z2 = torch.empty((n, H2, Wf2, 2), dtype=torch.float32, device=d)
    L.check(L.lib().ppsci_spectrum_resize(n, H, Wf, H2, Wf2, 0, 0, _p(Zd), _p(Z2), _stream_ptr(Z2)))
    y = torch.empty((n, H2, W2), dtype=torch.float32, device=d)
    L.check(L.lib().ppsci_fft2d_c2r(n, H2, W2, _p(Z2), _p(y), _stream_ptr(y)))
    assert rel(y.cpu().numpy(), y_ref.detach().numpy()) < 2e-6
    # the way back
    g = torch.as_tensor(rng.standard_normal((n, H2, W2)).astype(np.float32))
    (gz,) = torch.autograd.grad((y_ref * g.double()).sum(), Zc)  # dL/dRe + i dL/dIm
    gd = g.to(d).contiguous()
    ghat = torch.empty((n, H2, Wf2, 2), dtype=torch.float32, device=d)
    L.check(L.lib().ppsci_fft2d_r2c(n, H2, W2, _p(gd), _p(ghat), _stream_ptr(ghat)))
    Gb = torch.empty((n, H, Wf, 2), dtype=torch.float32, device=d)
    L.check(L.lib().ppsci_spectrum_resize(n, H2, Wf2, H, Wf, W2, W, _p(ghat), _p(Gb), _stream_ptr(Gb)))
    got = torch.view_as_complex(Gb.cpu().double().contiguous())
    cW = torch.tensor([1.0 if (j == 0 or 2 * j == W) else 2.0 for j in range(Wf)], dtype=torch.float64)
    # y = sum_j c(j) Re(Z_j e^{+i..}): dL/dRe Z_j + i dL/dIm Z_j = c(j) sum_x g_x e^{-i..} = c_{W2}(j) * rfft(g)_j on the entries that reach y
    want = gz
    have = got * cW
    # Im of the DC / Nyquist columns of the OUTPUT grid never reaches y: autograd has 0 there, rfft(g) is real there as well
    assert rel(have.real.numpy(), want.real.numpy()) < 2e-5
    assert rel(have.imag.numpy(), want.imag.numpy()) < 2e-5


@pytest.mark.parametrize("sizes", [(16, 16, 8, 8, 16, 16), (16, 16, 8, 8, 8, 8), (8, 8, 16, 16, 8, 8), (19, 19, 10, 10, 8, 8), (10, 10, 20, 20, 8, 8),
                                   (12, 20, 18, 10, 6, 8), (20, 20, 19, 19, 16, 16)])
def test_kept_mode_transforms_between_two_grids(sizes, dev):
    """ppsci_dft2_kept_inv_from == irfftn(S, s=(H2, W2)) of the half spectrum S (laid out for H x W) that holds Z at the rows / columns
    FactorizedSpectralConv writes to; ppsci_dft2_kept_fwd_from followed by the c_W weights == autograd through that irfftn."""
    from paddlescience_amd import _lib as L
    from paddlescience_amd.device import get_device
    from paddlescience_amd.hotpath import _p, _stream_ptr

    H, W, H2, W2, nmx, nmy = sizes
    mx, my = nmx, nmy // 2 + 1
    d = get_device()
    rng = np.random.default_rng(sum(sizes))
    n, Wf = 3, W // 2 + 1
    Z = torch.as_tensor(rng.standard_normal((n, mx, my, 2)).astype(np.float32))
    Zc = torch.view_as_complex(Z.double().contiguous()).requires_grad_(True)
    S = torch.zeros((n, H, Wf), dtype=torch.complex128)
    st = H - mx
    sl = slice(st // 2, -st // 2) if st else slice(None)  # fno_block.py:747-757 on the shifted spectrum
    S = torch.index_put(S, (torch.arange(n)[:, None, None], torch.arange(H)[sl][None, :, None], torch.arange(my)[None, None, :]), Zc)
    S = torch.fft.fftshift(S, dim=-2)  # fno_block.py:788-789
    y_ref = torch.fft.irfftn(S, s=(H2, W2), dim=(-2, -1), norm="forward")
    y = torch.empty((n, H2, W2), dtype=torch.float32, device=d)
    Zd = Z.to(d).contiguous()
    L.check(L.lib().ppsci_dft2_kept_inv_from(n, H2, W2, mx, my, H, W, _p(Zd), _p(y), _stream_ptr(y)))
    assert rel(y.cpu().numpy(), y_ref.detach().numpy()) < 2e-6
    g = torch.as_tensor(rng.standard_normal((n, H2, W2)).astype(np.float32))
    (gz,) = torch.autograd.grad((y_ref * g.double()).sum(), Zc)
    gd = g.to(d).contiguous()
    X = torch.empty((n, mx, my, 2), dtype=torch.float32, device=d)
    L.check(L.lib().ppsci_dft2_kept_fwd_from(n, H2, W2, mx, my, H, W, _p(gd), _p(X), _stream_ptr(X)))
    cW = torch.tensor([1.0 if (j == 0 or 2 * j == W) else 2.0 for j in range(my)], dtype=torch.float64)
    have = torch.view_as_complex(X.cpu().double().contiguous()) * cW
    assert rel(have.real.numpy(), gz.real.numpy()) < 2e-5
    assert rel(have.imag.numpy(), gz.imag.numpy()) < 2e-5


def test_unonet_trains_through_the_solver(dev, tmp_path):
    """SupervisedConstraint + Solver on a UNONet: the operator engine takes it like an FNONet (forward + backward on the kernels,
    Adam in the same launch as the gradient sums), the loss falls, eval at another resolution works."""
    import ppsci

    torch.manual_seed(0)
    np.random.seed(0)
    model = ppsci.arch.UNONet(("x",), ("y",), 1, 1, 8, lifting_channels=12, projection_channels=12, n_layers=3,
                              uno_out_channels=[8, 8, 8], uno_n_modes=[[8, 8], [4, 4], [4, 4]],
                              uno_scalings=[[0.5, 0.5], [1, 1], [2, 2]], norm="group_norm", domain_padding=None)
    rng = np.random.default_rng(1)
    x = rng.standard_normal((8, 1, 16, 16)).astype(np.float32)
    y = np.cumsum(x, axis=-1) * 0.1
    cst = ppsci.constraint.SupervisedConstraint(
        {"dataset": {"name": "NamedArrayDataset", "input": {"x": x}, "label": {"y": y}}, "batch_size": 4,
         "sampler": {"name": "BatchSampler", "drop_last": True, "shuffle": False}},
        ppsci.loss.MSELoss("mean"), name="sup")
    opt = ppsci.optimizer.Adam(1e-2)(model)
    solver = ppsci.solver.Solver(model, {cst.name: cst}, str(tmp_path), opt, epochs=12, iters_per_epoch=2)
    def mse():
        out = solver.predict({"x": x[:4]}, batch_size=4)["y"]
        assert tuple(out.shape) == (4, 1, 16, 16)
        return float(((np.asarray(out.cpu() if hasattr(out, "cpu") else out) - y[:4]) ** 2).mean())

    before = mse()
    solver.train()
    after = mse()
    assert after < 0.4 * before, (before, after)
    out32 = model({"x": torch.as_tensor(rng.standard_normal((2, 1, 32, 32)).astype(np.float32))})["y"]
    assert tuple(out32.shape) == (2, 1, 32, 32) and bool(torch.isfinite(out32).all())
