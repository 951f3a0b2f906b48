"""FNO spectral convolution (BASELINE config 4): the HIP per-mode complex contraction (+ its gradients)
against a plain-torch restatement of FactorizedSpectralConv.forward
(/root/reference/ppsci/arch/fno_block.py:707-796, _contract_dense_trick :346-372) in fp64."""
import numpy as np
import pytest
import torch

from oracle import ref_torch as R
from paddlescience_amd.arch import fno
from tests.common import make_dev_fixture, rel

dev = make_dev_fixture()


@pytest.mark.parametrize("B,ci,co,H,W,modes", [(3, 5, 7, 16, 16, (8, 8)), (16, 32, 32, 64, 64, (12, 12)), (2, 4, 4, 8, 12, (8, 6))])
def test_spectral_conv_forward_and_gradients(B, ci, co, H, W, modes, dev):
    if dev == "emu" and B * ci * co > 2000:
        pytest.skip("too slow under the CPU emulator; runs on the GPU")
    from paddlescience_amd.device import get_device

    d = get_device()
    torch.manual_seed(0)
    layer = fno.SpectralConv2d(ci, co, modes, bias=True, fft_norm="forward").to(d)
    x = torch.randn(B, ci, H, W, device=d, requires_grad=True)
    y = layer(x)
    g = torch.randn_like(y)
    gx, gwr, gwi = torch.autograd.grad(y, [x, layer.weight_real, layer.weight_imag], g)
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


def test_odd_sizes_are_rejected(dev):
    from paddlescience_amd.device import get_device

    layer = fno.SpectralConv2d(2, 2, (4, 4)).to(get_device())
    with pytest.raises(RuntimeError):
        layer(torch.randn(1, 2, 9, 8, device=get_device()))
