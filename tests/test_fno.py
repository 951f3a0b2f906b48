"""FNO spectral convolution (BASELINE config 4): rfftn -> per-mode complex contraction -> irfftn (+ bias) through the
C ABI (ppsci_fft2d_r2c / ppsci_spectral_conv2d_fwd_scaled / ppsci_fft2d_c2r) and its hand-written adjoint
(ppsci_spectral_conv2d_bwd_real_scaled) against a plain-torch restatement of FactorizedSpectralConv.forward
(/root/reference/ppsci/arch/fno_block.py:707-796, _contract_dense_trick :346-372) in fp64."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import ref_torch as R
from paddlescience_amd import _lib as L
from paddlescience_amd.arch import fno
from paddlescience_amd.hotpath import _p, _stream_ptr
from tests.common import make_dev_fixture, rel

dev = make_dev_fixture()


def _layer(layer, x, g):
    """y = irfftn(contract(rfftn(x))) + bias and the adjoints (gx, gw_re, gw_im) of <y, g>, as fno_engine.FnoNative runs them."""
    B, ci, H, W = x.shape
    co, Wf = layer.out_channels, W // 2 + 1
    lib = L.lib()
    st = _stream_ptr(x)
    f = dict(dtype=torch.float32, device=x.device)
    d = L.SpectralDesc()
    d.batch, d.c_in, d.c_out, d.h, d.wf, d.modes_x, d.modes_y = B, ci, co, H, Wf, *layer.n_modes
    xft, oft = torch.empty((B, ci, H, Wf, 2), **f), torch.empty((B, co, H, Wf, 2), **f)
    y = torch.empty((B, co, H, W), **f)
    inv_n = 1.0 / (H * W)
    L.check(lib.ppsci_fft2d_r2c(B * ci, H, W, _p(x), _p(xft), st))
    L.check(lib.ppsci_spectral_conv2d_fwd_scaled(C.byref(d), _p(xft), _p(layer.weight_real), _p(layer.weight_imag), _p(oft),
                                                 inv_n, 1, st))
    L.check(lib.ppsci_fft2d_c2r(B * co, H, W, _p(oft), _p(y), st))
    y = y + layer.bias.detach().view(1, -1, 1, 1)
    ghat, gxft = torch.empty((B, co, H, Wf, 2), **f), torch.empty((B, ci, H, Wf, 2), **f)
    gx = torch.empty_like(x)
    gwr, gwi = torch.empty_like(layer.weight_real), torch.empty_like(layer.weight_imag)
    L.check(lib.ppsci_fft2d_r2c(B * co, H, W, _p(g.contiguous()), _p(ghat), st))
    L.check(lib.ppsci_spectral_conv2d_bwd_real_scaled(C.byref(d), _p(xft), _p(layer.weight_real), _p(layer.weight_imag), _p(ghat),
                                                      _p(gxft), _p(gwr), _p(gwi), inv_n, W, inv_n, 1, st))
    L.check(lib.ppsci_fft2d_c2r(B * ci, H, W, _p(gxft), _p(gx), st))
    return y, gx, gwr, gwi


@pytest.mark.parametrize("B,ci,co,H,W,modes", [(3, 5, 7, 16, 16, (8, 8)), (16, 32, 32, 64, 64, (12, 12)), (2, 4, 4, 8, 12, (8, 6)),
                                               (2, 3, 4, 9, 8, (4, 4)), (2, 3, 3, 11, 11, (6, 5)), (1, 4, 2, 10, 7, (5, 4)),
                                               (2, 8, 8, 69, 69, (12, 12)), (18, 96, 100, 8, 8, (4, 4))])
def test_spectral_conv_forward_and_gradients(B, ci, co, H, W, modes, dev):
    """Even sizes and odd ones (H = 9, 11, 69: the reference's second fftshift then lands one row off the first, see
    spec_row_in / spec_row_out; odd H - modes_x: the crop starts at (H - modes_x) // 2; odd W: no Nyquist column); 96 x 100
    channels: the weights of a mode do not fit LDS and are read from global memory; 18 samples: two batch tiles."""
    if dev == "emu" and (B * ci * co > 200000 or H * W > 2000 or (B * ci * co > 2000 and H * W > 100)):
        pytest.skip("too slow under the CPU emulator; runs on the GPU")
    from paddlescience_amd.device import get_device

    d = get_device()
    torch.manual_seed(0)
    layer = fno.SpectralConv2d(ci, co, modes, bias=True, fft_norm="forward").to(d)
    x = torch.randn(B, ci, H, W, device=d)
    g = torch.randn(B, co, H, W, device=d)
    y, gx, gwr, gwi = _layer(layer, x, g)
    # fp64 reference
    x64 = x.detach().double().cpu().requires_grad_(True)
    wr = layer.weight_real.detach().double().cpu().requires_grad_(True)
    wi = layer.weight_imag.detach().double().cpu().requires_grad_(True)
    yref = R.reference_spectral_conv2d(x64, wr, wi, modes[0], "forward", layer.bias.detach().double().cpu())
    gxr, gwrr, gwir = torch.autograd.grad(yref, [x64, wr, wi], g.double().cpu())
    assert rel(y.detach().cpu().numpy(), yref.detach().numpy()) < 2e-6
    assert rel(gx.cpu().numpy(), gxr.numpy()) < 2e-6
    assert rel(gwr.cpu().numpy(), gwrr.numpy()) < 2e-6
    assert rel(gwi.cpu().numpy(), gwir.numpy()) < 2e-6


@pytest.mark.parametrize("n,H,W,modes", [(3, 16, 16, (8, 5)), (2, 9, 8, (4, 3)), (2, 11, 13, (6, 4)), (4, 64, 64, (12, 7)),
                                         (2, 10, 12, (10, 7)), (1, 69, 69, (12, 7))])
def test_transforms_on_the_kept_modes(n, H, W, modes, dev):
    """ppsci_dft2_kept_fwd / _inv: rfftn at the rows / columns the spectral convolution keeps, and irfftn of a spectrum that
    is zero everywhere else -- against torch.fft in fp64 with the reference's fftshift / slice bookkeeping (rows 0: the rows
    taken from the shifted input spectrum; rows 1: where the second fftshift puts them; odd H: one apart).  A 10 x 12 case
    keeps ALL columns (Nyquist included), 64 x 64 is the BASELINE shape."""
    if dev == "emu" and H * W > 2000:
        pytest.skip("too slow under the CPU emulator; runs on the GPU")
    from paddlescience_amd.device import get_device

    d = get_device()
    mx, my = modes
    lib = L.lib()
    assert lib.ppsci_dft2_kept_supported(H, W, mx, my) == 1
    rng = np.random.default_rng(3)
    x = torch.as_tensor(rng.standard_normal((n, H, W)).astype(np.float32)).to(d)
    start = H - mx
    rows = slice(start // 2, -start // 2) if start else slice(None)
    xf = torch.fft.rfftn(x.double().cpu(), dim=(-2, -1))
    for which in (0, 1):
        # the frequency rows behind shifted rows `rows`.  rows 0: shifted = fftshift(spectrum), i.e. shifted[s] = spectrum[
        # fftshift(arange)[s]]; rows 1: spectrum = fftshift(shifted), i.e. shifted[s] lands at spectrum[ifftshift(arange)[s]]
        idx = torch.arange(H)
        src_rows = (torch.fft.fftshift(idx) if which == 0 else torch.fft.ifftshift(idx))[rows]
        X = torch.full((n, mx, my, 2), float("nan"), device=d)
        L.check(lib.ppsci_dft2_kept_fwd(n, H, W, mx, my, which, _p(x), _p(X), _stream_ptr(x)))
        ref = xf[:, src_rows][:, :, :my]
        got = torch.view_as_complex(X.double().cpu().contiguous())
        assert rel(torch.view_as_real(got).numpy(), torch.view_as_real(ref).numpy()) < 2e-6, which
        # inverse: the kept modes placed at those rows of a zero spectrum
        Z = torch.as_tensor(rng.standard_normal((n, mx, my, 2)).astype(np.float32)).to(d)
        full = torch.zeros((n, H, W // 2 + 1), dtype=torch.complex128)
        full[:, src_rows, :my] = torch.view_as_complex(Z.double().cpu().contiguous())
        yref = torch.fft.irfftn(full, s=(H, W), dim=(-2, -1), norm="forward")  # unscaled, as hipFFT C2R
        y = torch.full((n, H, W), float("nan"), device=d)
        L.check(lib.ppsci_dft2_kept_inv(n, H, W, mx, my, which, _p(Z), _p(y), _stream_ptr(y)))
        assert rel(y.cpu().numpy(), yref.numpy()) < 2e-6, which

