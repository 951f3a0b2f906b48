"""FNO spectral convolution on the HIP MFMA kernel (BASELINE config 4).

`SpectralConv2d` follows FactorizedSpectralConv (dense, non-factorized weights, order 2) of
/root/reference/ppsci/arch/fno_block.py:545-796: weights real/imag `[Ci, Co, n_modes[0], n_modes[1]//2+1]`
drawn N(0, sqrt(2/(Ci+Co))) (:621-622, :522-532), optional bias `[Co, 1, 1]`, `fft_norm` forwarded to the
FFTs.  The FFTs run in hipFFT through torch.fft (glue, as SURVEY.md allows); the per-mode complex channel
contraction -- the part the reference does with four real einsums -- is `ppsci_spectral_conv2d_fwd/bwd`,
wired into torch autograd so the layer can sit inside a larger torch module (lifting / projection 1x1
convolutions, GroupNorm, GELU of FNOBlocks are plain library ops)."""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional, Tuple

import numpy as np
import torch

from .. import _lib as L
from ..hotpath import _p, _require_device, _stream_ptr


def _desc(B, ci, co, H, Wf, mx, my) -> L.SpectralDesc:
    d = L.SpectralDesc()
    d.batch, d.c_in, d.c_out, d.h, d.wf, d.modes_x, d.modes_y = B, ci, co, H, Wf, mx, my
    return d


class _SpectralContract(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_ft: torch.Tensor, w_re: torch.Tensor, w_im: torch.Tensor):
        _require_device(w_re)
        B, ci, H, Wf = x_ft.shape
        _, co, mx, my = w_re.shape
        xr = torch.view_as_real(x_ft.contiguous()).contiguous()
        out = torch.zeros((B, co, H, Wf, 2), dtype=torch.float32, device=x_ft.device)
        d = _desc(B, ci, co, H, Wf, mx, my)
        L.check(L.lib().ppsci_spectral_conv2d_fwd(C.byref(d), _p(xr), _p(w_re.contiguous()), _p(w_im.contiguous()),
                                                  _p(out), _stream_ptr(out)))
        ctx.save_for_backward(xr, w_re, w_im)
        ctx.desc = d
        return torch.view_as_complex(out)

    @staticmethod
    def backward(ctx, g_ft: torch.Tensor):
        xr, w_re, w_im = ctx.saved_tensors
        d = ctx.desc
        g = torch.view_as_real(g_ft.contiguous()).contiguous()
        gx = torch.zeros_like(xr)
        gwr = torch.empty_like(w_re)
        gwi = torch.empty_like(w_im)
        L.check(L.lib().ppsci_spectral_conv2d_bwd(C.byref(d), _p(xr), _p(w_re.contiguous()), _p(w_im.contiguous()), _p(g),
                                                  _p(gx), _p(gwr), _p(gwi), _stream_ptr(gx)))
        return torch.view_as_complex(gx), gwr, gwi


def spectral_contract(x_ft: torch.Tensor, w_re: torch.Tensor, w_im: torch.Tensor) -> torch.Tensor:
    """out_ft[b,o,r,c] = sum_i x_ft[b,i,r,c] * (w_re + i w_im)[i,o,m(r),c] on the kept modes, 0 elsewhere."""
    return _SpectralContract.apply(x_ft, w_re, w_im)


class SpectralConv2d(torch.nn.Module):
    def __init__(self, in_channels: int, out_channels: int, n_modes: Tuple[int, int], bias: bool = True,
                 fft_norm: str = "backward", init_std: Optional[float] = None):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.n_modes = (int(n_modes[0]), int(n_modes[1]) // 2 + 1)  # fno_block.py:673-684
        self.fft_norm = fft_norm
        std = (2 / (in_channels + out_channels)) ** 0.5 if init_std is None else init_std
        shape = (in_channels, out_channels, *self.n_modes)
        self.weight_real = torch.nn.Parameter(torch.randn(shape) * std)
        self.weight_imag = torch.nn.Parameter(torch.randn(shape) * std)
        self.bias = torch.nn.Parameter(std * torch.randn(out_channels, 1, 1)) if bias else None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        H, W = x.shape[-2:]
        x_ft = torch.fft.rfftn(x, norm=self.fft_norm, dim=(-2, -1))
        out_ft = spectral_contract(x_ft, self.weight_real, self.weight_imag)
        y = torch.fft.irfftn(out_ft, s=(H, W), dim=(-2, -1), norm=self.fft_norm)
        if self.bias is not None:
            y = y + self.bias
        return y


def reference_spectral_conv2d(x, w_re, w_im, n_modes_x, fft_norm="backward", bias=None):
    """Plain torch restatement of FactorizedSpectralConv.forward (fno_block.py:707-796) with the explicit
    fftshift / slicing / four-einsum sequence; used by tests only."""
    B, ci, H, W = x.shape
    co, mx, my = w_re.shape[1], w_re.shape[2], w_re.shape[3]
    xf = torch.fft.rfftn(x, norm=fft_norm, dim=(-2, -1))
    xf = torch.fft.fftshift(xf, dim=(-2,))
    out = torch.zeros((B, co, H, W // 2 + 1), dtype=xf.dtype, device=x.device)
    start = H - mx
    rows = slice(start // 2, -start // 2) if start else slice(None)
    cols = slice(None, my)
    xs = xf[:, :, rows, cols]
    eq = "abcd,becd->aecd"
    o_r = torch.einsum(eq, xs.real, w_re) - torch.einsum(eq, xs.imag, w_im)
    o_i = torch.einsum(eq, xs.imag, w_re) + torch.einsum(eq, xs.real, w_im)
    out[:, :, rows, cols] = torch.complex(o_r, o_i)
    out = torch.fft.fftshift(out, dim=(-2,))
    y = torch.fft.irfftn(out, s=(H, W), dim=(-2, -1), norm=fft_norm)
    return y if bias is None else y + bias
