"""The native FNO path (paddlescience_amd/fno_engine.py + csrc/fno.hip: 1x1 convolutions, GroupNorm + skip + GELU block
tail, spectral contraction, all with hand-written backward; FFTs by hipFFT) against

  * tests/golden/fno.npz -- produced by the REFERENCE's own fno_block.py / tfnonet.py (tests/golden/make_fno_golden.py):
    output rel-L2 <= 2e-5, every parameter gradient rel-L2 <= 2e-4, loss rel <= 1e-5;
  * the torch-autograd path of the same model (the previous implementation) on a larger shape;
  * kernel-level known answers for the 1x1 convolution (forward, data gradient, weight gradient) and the block tail."""
import os

import numpy as np
import pytest
import torch

from tests.common import make_dev_fixture, rel

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "fno.npz"))
CASES = sorted({k.split("/")[0] for k in G.files})
dev = make_dev_fixture()


def _case(c):
    mx, my, hid, lift, proj, nl, gn = [int(v) for v in G[f"{c}/config"]]
    P = {k[len(c) + 7:]: G[k] for k in G.files if k.startswith(f"{c}/param/")}
    Gr = {k[len(c) + 6:]: G[k] for k in G.files if k.startswith(f"{c}/grad/")}
    return (mx, my, hid, lift, proj, nl, "group_norm" if gn else None), P, Gr


def _padding(c):
    """DomainPadding of the case (fno_block.py:19-140): (fractions or None, mode)."""
    fh, fw, sym = [float(v) for v in G[f"{c}/domain_padding"]]
    return ([fh, fw] if fh or fw else None), ("symmetric" if sym else "one-sided")


@pytest.mark.parametrize("c", CASES[:2])
def test_contraction_inside_the_inverse_transform_equals_the_two_launches(c, dev, monkeypatch):
    """ppsci_spectral_conv2d_inv_kept (the forward contraction inside the inverse transform's launch, with or without the tail's row
    sums) against ppsci_spectral_conv2d_fwd_kept + ppsci_dft2_kept_inv[_stats]: the same output up to the order of the channel sum."""
    import ppsci
    from paddlescience_amd.fno_engine import FnoNative

    (mx, my, hid, lift, proj, nl, norm), P, _ = _case(c)
    pad, pad_mode = _padding(c)
    outs = []
    for fuse in ("1", "0"):
        monkeypatch.setenv("PPSCI_FNO_FUSE_CONTRACT", fuse)
        model = ppsci.arch.TFNO2dNet(("x",), ("y",), mx, my, hid, 3, 1, lift, proj, nl, norm=norm, domain_padding=pad,
                                     domain_padding_mode=pad_mode)
        model.set_state_dict({k: v.astype(np.float32) for k, v in P.items()})
        eng = FnoNative(model)
        x = torch.as_tensor(G[f"{c}/x"].astype(np.float32)).to(model.flat_params.device)
        outs.append(eng.forward(x).detach().cpu().numpy().copy())
        assert eng.fuse_contract == (fuse == "1" and eng.kept)
    assert rel(outs[0], outs[1]) < 1e-6


@pytest.mark.parametrize("full_fft", [False, True])
@pytest.mark.parametrize("c", CASES)
def test_native_path_reproduces_reference_fno(c, dev, full_fft, monkeypatch):
    """full_fft: the library FFT on the whole spectrum (the path of planes too large for the kept-mode transforms)."""
    import ppsci

    monkeypatch.setenv("PPSCI_FNO_FULL_FFT", "1" if full_fft else "0")
    from paddlescience_amd.fno_engine import FnoNative

    (mx, my, hid, lift, proj, nl, norm), P, Gr = _case(c)
    pad, pad_mode = _padding(c)
    model = ppsci.arch.TFNO2dNet(("x",), ("y",), mx, my, hid, 3, 1, lift, proj, nl, norm=norm, domain_padding=pad,
                                 domain_padding_mode=pad_mode)
    model.set_state_dict({k: v.astype(np.float32) for k, v in P.items()})
    d = model.flat_params.device
    x = torch.as_tensor(G[f"{c}/x"].astype(np.float32)).to(d)
    eng = FnoNative(model)
    y = eng.forward(x)
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
        assert rel(p.grad.cpu().numpy(), Gr[n]) < 2e-4, n


@pytest.mark.parametrize("stabilizer", [None, "tanh"])
def test_native_path_matches_oracle_and_is_reproducible(dev, stabilizer):
    """Darcy-like shape (16x16, batch 4, 16 hidden channels, 2 blocks), with and without the tanh stabilizer
    (fno_block.py:1199): output and gradients against the fp64 restatement (oracle/ref_torch.fno_forward), loss value and
    adjoint from the field-loss kernels; a second run gives bit-identical gradients (fixed-order reductions)."""
    import ppsci
    from oracle import ref_torch as R

    torch.manual_seed(3)
    model = ppsci.arch.FNONet(("x",), ("y",), (8, 8), 16, 3, 1, 24, 20, 2, norm="group_norm", stabilizer=stabilizer)
    d = model.flat_params.device
    rng = np.random.default_rng(0)
    x = torch.as_tensor(rng.standard_normal((4, 3, 16, 16)).astype(np.float32)).to(d)
    tgt = torch.as_tensor(rng.standard_normal((4, 1, 16, 16)).astype(np.float32)).to(d)
    P = {n: p.detach().cpu().double().requires_grad_(True) for n, p in torch.nn.Module.named_parameters(model)}
    yo = R.fno_forward(x.cpu().double(), P, 2, (8, 8), "group_norm", stabilizer=stabilizer)
    lo = ((yo - tgt.cpu().double()) ** 2).mean()
    names = [n for n, _ in torch.nn.Module.named_parameters(model)]
    go = np.concatenate([g.numpy().ravel() for g in torch.autograd.grad(lo, [P[n] for n in names])])
    eng = model.native()
    mse = ppsci.loss.MSELoss("mean")
    runs = []
    for _ in range(2):
        y = eng.forward(x)
        losses, gy = mse.value_and_grad(y, tgt, "y")
        model.flat_grad.fill_(float("nan"))
        eng.backward(gy)
        runs.append(model.flat_grad.clone())
    assert rel(y.cpu().numpy(), yo.detach().numpy()) < 1e-5
    assert abs(float(losses["y"]) / float(lo) - 1.0) < 1e-5
    assert rel(runs[0].cpu().numpy(), go) < 2e-4
    assert torch.equal(runs[0], runs[1])


@pytest.mark.parametrize("hw,pad", [((10, 10), 0.1), ((9, 7), None), ((14, 14), [0.078125, 0.0])])
def test_native_path_on_plane_sizes_that_are_not_multiples_of_16(dev, hw, pad):
    """DomainPadding fractions like the reference yaml's 0.078125 (64 -> 69) or 0.1 (64 -> 70) give planes whose size is
    not a multiple of 16 or even odd (rows then do not start on 16-byte boundaries): 10 x 10 -> 11 x 11, 9 x 7 unpadded,
    14 x 14 -> 15 x 14.  Output, loss and every gradient against the fp64 restatement."""
    import ppsci
    from oracle import ref_torch as R

    torch.manual_seed(5)
    H, W = hw
    model = ppsci.arch.FNONet(("x",), ("y",), (4, 4), 12, 3, 1, 20, 16, 2, norm="group_norm", domain_padding=pad)
    d = model.flat_params.device
    rng = np.random.default_rng(2)
    x = torch.as_tensor(rng.standard_normal((3, 3, H, W)).astype(np.float32)).to(d)
    tgt = torch.as_tensor(rng.standard_normal((3, 1, H, W)).astype(np.float32)).to(d)
    P = {n: p.detach().cpu().double().requires_grad_(True) for n, p in torch.nn.Module.named_parameters(model)}
    yo = R.fno_forward(x.cpu().double(), P, 2, (4, 4), "group_norm", domain_padding=pad)
    lo = ((yo - tgt.cpu().double()) ** 2).mean()
    names = [n for n, _ in torch.nn.Module.named_parameters(model)]
    go = np.concatenate([g.numpy().ravel() for g in torch.autograd.grad(lo, [P[n] for n in names])])
    eng = model.native()
    y = eng.forward(x)
    losses, gy = ppsci.loss.MSELoss("mean").value_and_grad(y, tgt, "y")
    model.flat_grad.fill_(float("nan"))
    eng.backward(gy)
    assert rel(y.cpu().numpy(), yo.detach().numpy()) < 1e-5
    assert abs(float(losses["y"]) / float(lo) - 1.0) < 1e-5
    assert rel(model.flat_grad.cpu().numpy(), go) < 2e-4


@pytest.mark.parametrize("shape", [(2, 5, 37, 48), (3, 256, 256, 80), (2, 200, 176, 64), (1, 64, 1, 32), (2, 36, 260, 16),
                                   (2, 5, 37, 75), (1, 40, 24, 100), (1, 33, 17, 301)])
@pytest.mark.parametrize("npx", [0, 1, 2, 4])
def test_pw_conv_and_tail_kernels_known_answers(dev, shape, npx):
    """(B, Ci, Co, P): small ragged channels (scalar weight staging), 256 x 256 (two output-channel slabs, 16-byte staging,
    2 x 2 weight-gradient tiles), channel counts that are not multiples of 16 on the vector path, one output channel, more
    output channels than one slab with a ragged tail; plane sizes that are odd (75, 301: two weight-gradient chunks) or a
    multiple of 4 but not of 16 (100)."""
    import ctypes as C

    from paddlescience_amd import _lib as L
    from paddlescience_amd import device, hotpath as hp

    d = device.get_device()
    rng = np.random.default_rng(1)
    B, Ci, Co, P = shape
    if dev != "gpu" and npx in (1, 2) and Ci * Co > 20000:
        pytest.skip("covered at the other sizes (emulator time)")
    L.lib().ppsci_set_pw_pixels_per_lane(npx)  # work items of 16 npx pixels (0: chosen from the problem size)
    x = torch.as_tensor(rng.standard_normal((B, Ci, P)).astype(np.float32)).to(d)
    W = torch.as_tensor(rng.standard_normal((Co, Ci)).astype(np.float32)).to(d)
    b = torch.as_tensor(rng.standard_normal(Co).astype(np.float32)).to(d)
    z = torch.full((B, Co, P), float("nan"), device=d)
    a = torch.full((B, Co, P), float("nan"), device=d)
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())  # noqa: E731
    L.check(L.lib().ppsci_pw_conv(B, Ci, Co, P, p(x), p(W), 0, p(b), None, 0, p(z), p(a), None))
    zr = torch.einsum("oi,bip->bop", W.double(), x.double()) + b.double()[None, :, None]
    assert rel(z.cpu().numpy(), zr.cpu().numpy()) < 1e-6
    assert rel(a.cpu().numpy(), torch.nn.functional.gelu(zr).cpu().numpy()) < 1e-6
    gy = torch.as_tensor(rng.standard_normal((B, Co, P)).astype(np.float32)).to(d)
    zprev = torch.as_tensor(rng.standard_normal((B, Ci, P)).astype(np.float32)).to(d)
    gx = torch.full((B, Ci, P), float("nan"), device=d)
    L.check(L.lib().ppsci_pw_conv(B, Co, Ci, P, p(gy), p(W), 1, None, p(zprev), 0, p(gx), None, None))
    zp = zprev.double().requires_grad_(True)
    torch.nn.functional.gelu(zp).sum().backward()
    gxr = torch.einsum("oi,bop->bip", W.double(), gy.double()) * zp.grad
    assert rel(gx.cpu().numpy(), gxr.cpu().numpy()) < 1e-6
    chunks = int(L.lib().ppsci_pw_conv_wgrad_chunks(B, P))
    pw = torch.full((chunks * Co * Ci,), float("nan"), device=d)
    pb = torch.full((chunks * Co,), float("nan"), device=d)
    L.check(L.lib().ppsci_pw_conv_wgrad(B, Ci, Co, P, p(x), p(gy), p(pw), p(pb), None))
    gW = torch.zeros(Co * Ci, device=d)
    gb = torch.zeros(Co, device=d)
    hp.reduce_rows(pw, chunks, Co * Ci, gW, False)
    hp.reduce_rows(pb, chunks, Co, gb, False)
    assert rel(gW.cpu().numpy().reshape(Co, Ci), torch.einsum("bop,bip->oi", gy.double(), x.double()).cpu().numpy()) < 1e-6
    assert rel(gb.cpu().numpy(), gy.double().sum((0, 2)).cpu().numpy()) < 1e-6
    L.lib().ppsci_set_pw_pixels_per_lane(0)


@pytest.mark.parametrize("shape", [(2, 3, 40, 24, 100), (1, 4, 256, 32, 77), (2, 2, 33, 17, 64)])
def test_pw_conv_operands_evaluated_on_load(dev, shape):
    """ppsci_pw_virtual (B, K0, hidden C, Co, P): a 1x1 convolution / weight gradient whose input is GELU(z) of a stored
    pre-activation z (mode 1) or GELU(W0 x0 + b0) of the K0 <= 4 channel tensor x0, never stored (mode 2), and a data
    gradient multiplied by GELU'(W0 x0 + b0) -- against the same calls on the materialised tensors."""
    import ctypes as C

    from paddlescience_amd import _lib as L
    from paddlescience_amd import device, hotpath as hp
    from paddlescience_amd.fno_engine import _virtual

    d = device.get_device()
    rng = np.random.default_rng(4)
    B, K0, Ch, Co, P = shape
    t = lambda *sh: torch.as_tensor(rng.standard_normal(sh).astype(np.float32)).to(d)  # noqa: E731
    x0, W0, b0, W, bias, gy = t(B, K0, P), t(Ch, K0), t(Ch), t(Co, Ch), t(Co), t(B, Co, P)
    p = lambda v: None if v is None else C.c_void_p(v.data_ptr())  # noqa: E731
    lib = L.lib()
    # materialised: z = W0 x0 + b0, a = GELU(z)
    z = torch.empty((B, Ch, P), device=d)
    a = torch.empty((B, Ch, P), device=d)
    L.check(lib.ppsci_pw_conv(B, K0, Ch, P, p(x0), p(W0), 0, p(b0), None, 0, p(z), p(a), None))
    ref = torch.empty((B, Co, P), device=d)
    L.check(lib.ppsci_pw_conv(B, Ch, Co, P, p(a), p(W), 0, p(bias), None, 0, p(ref), None, None))
    v1, v2 = _virtual(1), _virtual(2, x0, W0, b0)
    for x, xv in ((z, v1), (None, v2)):
        out = torch.full((B, Co, P), float("nan"), device=d)
        L.check(lib.ppsci_pw_conv_v(B, Ch, Co, P, p(x), C.byref(xv), p(W), 0, p(bias), None, None, 0, p(out), None, None))
        assert rel(out.cpu().numpy(), ref.cpu().numpy()) < 1e-6
    # data gradient x GELU'(z): zmul = z against the virtual zmul
    gref = torch.empty((B, Ch, P), device=d)
    L.check(lib.ppsci_pw_conv(B, Co, Ch, P, p(gy), p(W), 1, None, p(z), 0, p(gref), None, None))
    gout = torch.full((B, Ch, P), float("nan"), device=d)
    L.check(lib.ppsci_pw_conv_v(B, Co, Ch, P, p(gy), None, p(W), 1, None, None, C.byref(v2), 0, p(gout), None, None))
    assert rel(gout.cpu().numpy(), gref.cpu().numpy()) < 1e-6
    # weight gradient with x = a
    chunks = int(lib.ppsci_pw_conv_wgrad_chunks(B, P))
    res = []
    for x, xv in ((a, None), (z, v1), (None, v2)):
        pw = torch.full((chunks * Co * Ch,), float("nan"), device=d)
        pb = torch.full((chunks * Co,), float("nan"), device=d)
        L.check(lib.ppsci_pw_conv_wgrad_v(B, Ch, Co, P, p(x), C.byref(xv) if xv is not None else None, p(gy), p(pw), p(pb), 0, None))
        gW, gb = torch.zeros(Co * Ch, device=d), torch.zeros(Co, device=d)
        hp.reduce_rows(pw, chunks, Co * Ch, gW, False)
        hp.reduce_rows(pb, chunks, Co, gb, False)
        res.append((gW.cpu().numpy(), gb.cpu().numpy()))
    for gW, gb in res[1:]:
        assert rel(gW, res[0][0]) < 1e-6 and rel(gb, res[0][1]) < 1e-6

