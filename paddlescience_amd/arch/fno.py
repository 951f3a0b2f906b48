"""FNO spectral convolution on the HIP MFMA kernel (BASELINE config 4).

`SpectralConv2d` follows FactorizedSpectralConv (dense, non-factorized weights, order 2) of
/root/reference/ppsci/arch/fno_block.py:545-796: weights real/imag `[Ci, Co, n_modes[0], n_modes[1]//2+1]`
drawn N(0, sqrt(2/(Ci+Co))) (:621-622, :522-532), optional bias `[Co, 1, 1]`, `fft_norm` forwarded to the
FFTs.  The FFTs run in hipFFT through torch.fft (glue, as SURVEY.md allows); the per-mode complex channel
contraction -- the part the reference does with four real einsums -- is `ppsci_spectral_conv2d_fwd/bwd`,
wired into torch autograd so the layer can sit inside a larger torch module.

`FNOBlocks`, `FNONet`, `TFNO2dNet` follow fno_block.py:1047-1255 and tfnonet.py:13-406 for the configuration of
BASELINE config 4 (post-activation blocks, linear skips, optional GroupNorm, dense weights); the lifting /
projection / skip 1x1 convolutions, GroupNorm and GELU are plain library ops (MIOpen / rocBLAS through torch).
Options of the reference signature that are not built raise NotImplementedError."""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional, Tuple

import numpy as np
import torch

from .. import _lib as L
from ..hotpath import _p, _require_device, _stream_ptr
from . import base


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




class ChannelMLP(torch.nn.Module):
    """fno_block.MLP (fno_block.py:263-320): n_layers 1x1 convolutions with the non-linearity in between."""

    def __init__(self, in_channels: int, out_channels: Optional[int] = None, hidden_channels: Optional[int] = None,
                 n_layers: int = 2, non_linearity=torch.nn.functional.gelu):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        hidden_channels = in_channels if hidden_channels is None else hidden_channels
        self.n_layers = n_layers
        self.non_linearity = non_linearity
        dims = [in_channels] + [hidden_channels] * (n_layers - 1) + [out_channels]
        self.fcs = torch.nn.ModuleList([torch.nn.Conv2d(dims[i], dims[i + 1], 1) for i in range(n_layers)])

    def forward(self, x):
        for i, fc in enumerate(self.fcs):
            x = fc(x)
            if i < self.n_layers - 1:
                x = self.non_linearity(x)
        return x


class FNOBlocks(torch.nn.Module):
    """fno_block.FNOBlocks, post-activation form (fno_block.py:1191-1220): per layer
    x <- act( norm(SpectralConv_i(x)) + skip_i(x) ), no activation after the last layer."""

    def __init__(self, in_channels: int, out_channels: int, n_modes: Tuple[int, int], n_layers: int = 1,
                 non_linearity=torch.nn.functional.gelu, stabilizer: Optional[str] = None, norm: Optional[str] = None,
                 fno_skip: str = "linear", fft_norm: str = "forward"):
        super().__init__()
        if in_channels != out_channels:
            raise NotImplementedError("FNOBlocks with in_channels != out_channels")
        if fno_skip not in ("linear", "identity"):
            raise NotImplementedError(f"fno_skip={fno_skip!r} (built: 'linear', 'identity')")
        if norm not in (None, "group_norm", "instance_norm"):
            raise NotImplementedError(f"norm={norm!r} (built: None, 'group_norm', 'instance_norm')")
        if stabilizer not in (None, "tanh"):
            raise ValueError(f"stabilizer={stabilizer!r}")
        self.n_layers, self.non_linearity, self.stabilizer = n_layers, non_linearity, stabilizer
        # FactorizedSpectralConv holds the weights of all layers; bias per layer (fno_block.py:652-663)
        self.convs = torch.nn.ModuleList([SpectralConv2d(in_channels, out_channels, n_modes, bias=True, fft_norm=fft_norm)
                                          for _ in range(n_layers)])
        self.fno_skips = torch.nn.ModuleList([
            torch.nn.Conv2d(in_channels, out_channels, 1, bias=False) if fno_skip == "linear" else torch.nn.Identity()
            for _ in range(n_layers)])
        if norm == "group_norm":
            self.norm = torch.nn.ModuleList([torch.nn.GroupNorm(1, out_channels) for _ in range(n_layers)])
        elif norm == "instance_norm":
            self.norm = torch.nn.ModuleList([torch.nn.InstanceNorm2d(out_channels) for _ in range(n_layers)])
        else:
            self.norm = None

    def forward(self, x, index: int = 0):
        x_skip = self.fno_skips[index](x)
        if self.stabilizer == "tanh":
            x = torch.tanh(x)
        x_fno = self.convs[index](x)
        if self.norm is not None:
            x_fno = self.norm[index](x_fno)
        x = x_fno + x_skip
        if index < self.n_layers - 1:
            x = self.non_linearity(x)
        return x


class FNONet(base.Arch, torch.nn.Module):
    """ppsci.arch.FNONet for 2-D problems (tfnonet.py:13-193).  Input: dict with one `[B, C, H, W]` tensor per
    input key (concatenated along the channel axis, like `concat_to_tensor` with the reference's layout);
    output: `{output_keys[0]: [B, out_channels, H, W]}`.

    The parameters live in ONE flat fp32 buffer (`flat_params`, module parameters are views into it) so that
    the data-parallel all-reduce and the fused Adam kernel act on a single tensor, as for the PINN path."""

    is_operator = True  # Solver: train through torch autograd around the HIP spectral kernel

    def __init__(self, input_keys: Tuple[str, ...], output_keys: Tuple[str, ...], n_modes: Tuple[int, ...],
                 hidden_channels: int, in_channels: int = 3, out_channels: int = 1, lifting_channels: int = 256,
                 projection_channels: int = 256, n_layers: int = 4, use_mlp: bool = False, mlp=None,
                 max_n_modes=None, non_linearity=torch.nn.functional.gelu, stabilizer: Optional[str] = None,
                 norm: Optional[str] = None, ada_in_features=None, preactivation: bool = False,
                 fno_skip: str = "linear", mlp_skip: str = "soft-gating", separable: bool = False,
                 factorization: Optional[str] = None, rank: float = 1.0, joint_factorization: bool = False,
                 implementation: str = "factorized", domain_padding=None, domain_padding_mode: str = "one-sided",
                 fft_norm: str = "forward", patching_levels: int = 0, **kwargs):
        torch.nn.Module.__init__(self)
        base.Arch.__init__(self)
        if len(n_modes) != 2:
            raise NotImplementedError("only the 2-D spectral convolution has a HIP kernel (TFNO1dNet / TFNO3dNet)")
        for name, val, ok in (("use_mlp", use_mlp, False), ("preactivation", preactivation, False),
                              ("separable", separable, False), ("joint_factorization", joint_factorization, False),
                              ("patching_levels", patching_levels, 0), ("ada_in_features", ada_in_features, None),
                              ("max_n_modes", max_n_modes, None)):
            if val != ok:
                raise NotImplementedError(f"FNONet({name}={val!r}) is not built")
        # DomainPadding (fno_block.py:19-140, tfnonet.py:118-127): per-axis fractions of the resolution, or None
        if domain_padding is not None and (sum(domain_padding) if isinstance(domain_padding, (list, tuple)) else domain_padding) > 0:
            fr = list(domain_padding) if isinstance(domain_padding, (list, tuple)) else [float(domain_padding)] * 2
            if len(fr) != 2:
                raise ValueError("domain_padding length must match the number of spatial dimensions (2)")
            mode = domain_padding_mode.lower()
            if mode not in ("one-sided", "symmetric"):
                raise ValueError(f"Got self.padding_mode = {mode}")
            self.domain_padding = ([float(v) for v in fr], mode)
        else:
            self.domain_padding = None
        # `factorization`/`rank`: the reference's FactorizedTensor (fno_block.py:522-539) stores a DENSE complex
        # weight whatever the name says, so "Tucker" with rank 1.0 and None are the same parametrisation.
        self.input_keys, self.output_keys = tuple(input_keys), tuple(output_keys)
        self.n_modes, self.n_layers = tuple(n_modes), n_layers
        self.hidden_channels, self.in_channels, self.out_channels = hidden_channels, in_channels, out_channels
        self.fno_blocks = FNOBlocks(hidden_channels, hidden_channels, self.n_modes, n_layers, non_linearity, stabilizer,
                                    norm, fno_skip, fft_norm)
        if lifting_channels:
            self.lifting = ChannelMLP(in_channels, hidden_channels, lifting_channels, 2)
        else:
            self.lifting = ChannelMLP(in_channels, hidden_channels, hidden_channels, 1)
        self.projection = ChannelMLP(hidden_channels, out_channels, projection_channels, 2, non_linearity)
        self.flat_params: Optional[torch.Tensor] = None
        self.flat_grad: Optional[torch.Tensor] = None
        from ..device import get_device

        self.to_device(get_device())

    # ---- flat parameter buffer ---------------------------------------------------------------
    def to_device(self, device):
        """Moves the model and (re)packs every parameter as a view into `flat_params` / `flat_grad`."""
        torch.nn.Module.to(self, device)
        ps = [p for p in torch.nn.Module.parameters(self)]
        n = sum(p.numel() for p in ps)
        flat = torch.empty(n, dtype=torch.float32, device=device)
        grad = torch.zeros(n, dtype=torch.float32, device=device)
        off = 0
        for p in ps:
            k = p.numel()
            flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = flat[off:off + k].view_as(p.data)
            p.grad = grad[off:off + k].view_as(p.data)
            off += k
        self.flat_params, self.flat_grad = flat, grad
        return self

    def parameters(self, recurse: bool = True):
        return list(torch.nn.Module.parameters(self, recurse))

    def state_dict(self, *args, **kwargs):
        return {k: v.detach().clone() for k, v in torch.nn.Module.state_dict(self, *args, **kwargs).items()}

    def set_state_dict(self, state):
        own = torch.nn.Module.state_dict(self)
        with torch.no_grad():
            for k, v in state.items():
                own[k].copy_(torch.as_tensor(v).to(own[k].device))

    def train(self, mode: bool = True):
        torch.nn.Module.train(self, mode)
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    # ---- forward (tfnonet.py:179-193) ---------------------------------------------------------
    def padding_of(self, H: int, W: int):
        """(rows, columns, row offset, column offset) DomainPadding adds to an H x W plane: round(fraction * resolution)
        behind each axis (one-sided) or on both sides (symmetric), fno_block.py:72-115."""
        if self.domain_padding is None:
            return 0, 0, 0, 0
        (fh, fw), mode = self.domain_padding
        ph, pw = round(fh * H), round(fw * W)
        return (2 * ph, 2 * pw, ph, pw) if mode == "symmetric" else (ph, pw, 0, 0)

    def forward_tensor(self, x: torch.Tensor) -> torch.Tensor:
        x = self.lifting(x)
        H, W = x.shape[-2:]
        ah, aw, oh, ow = self.padding_of(H, W)
        if ah or aw:
            x = torch.nn.functional.pad(x, [ow, aw - ow, oh, ah - oh])
        for index in range(self.n_layers):
            x = self.fno_blocks(x, index)
        if ah or aw:
            x = x[..., oh:oh + H, ow:ow + W]
        return self.projection(x)

    def forward(self, x):
        if self._input_transform is not None:
            x = self._input_transform(x)
        dev = self.flat_params.device
        xs = [torch.as_tensor(x[k], dtype=torch.float32).to(dev) for k in self.input_keys]
        xt = xs[0] if len(xs) == 1 else torch.cat(xs, dim=1)
        out = {self.output_keys[0]: self.forward_tensor(xt)}
        if self._output_transform is not None:
            out = self._output_transform(x, out)
        return out

    __call__ = torch.nn.Module.__call__


class TFNO2dNet(FNONet):
    """ppsci.arch.TFNO2dNet (tfnonet.py:301-406)."""

    def __init__(self, input_keys: Tuple[str, ...], output_keys: Tuple[str, ...], n_modes_height: int,
                 n_modes_width: int, hidden_channels: int, in_channels: int = 3, out_channels: int = 1,
                 lifting_channels: int = 256, projection_channels: int = 256, n_layers: int = 4,
                 non_linearity=torch.nn.functional.gelu, use_mlp: bool = False, mlp=None, norm: Optional[str] = None,
                 skip: str = "soft-gating", separable: bool = False, preactivation: bool = False,
                 factorization: str = "Tucker", rank: float = 1.0, joint_factorization: bool = False,
                 implementation: str = "factorized", domain_padding=None, domain_padding_mode: str = "one-sided",
                 fft_norm: str = "forward", patching_levels: int = 0, **kwargs):
        # the reference forwards `skip` into FNONet's **kwargs, where it is ignored (fno_skip stays "linear")
        super().__init__(input_keys, output_keys, (n_modes_height, n_modes_width), hidden_channels, in_channels,
                         out_channels, lifting_channels, projection_channels, n_layers, use_mlp, mlp, None,
                         non_linearity, None, norm, None, preactivation, "linear", "soft-gating", separable,
                         factorization, rank, joint_factorization, implementation, domain_padding,
                         domain_padding_mode, fft_norm, patching_levels)
        self.n_modes_height, self.n_modes_width = n_modes_height, n_modes_width


class TFNO1dNet(FNONet):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("TFNO1dNet: only the 2-D spectral convolution has a HIP kernel")


class TFNO3dNet(FNONet):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("TFNO3dNet: only the 2-D spectral convolution has a HIP kernel")
