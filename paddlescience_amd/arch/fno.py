"""ppsci.arch.FNONet / TFNO2dNet (BASELINE config 4) on this framework's own kernels.

The classes here hold PARAMETERS only -- spectral weights real / imag `[Ci, Co, n_modes[0], n_modes[1]//2+1]` drawn
N(0, sqrt(2/(Ci+Co))) with a bias `[Co, 1, 1]` (/root/reference/ppsci/arch/fno_block.py:545-796, :621-622, :522-532), the
1x1 convolutions of lifting / projection / skips (fno_block.MLP :263-320, skip :190-226), GroupNorm scale / shift -- with
the reference's module tree and state-dict names.  Every forward (training, eval, predict, validators) and the whole
backward run in `fno_engine.FnoNative`: MFMA 1x1 convolutions, raw hipFFT executions, the per-mode complex contraction,
the fused GroupNorm + bias + skip + GELU block tail (csrc/fno.hip, csrc/fft.hip, csrc/spectral_conv.hip).  There is no
second (library-op) implementation of the network; options of the reference signature that the kernels do not cover raise
NotImplementedError with the reason."""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch

from . import base


def gelu(x):
    """The default `non_linearity` marker (paddle.nn.functional.gelu in the reference): the kernels fuse GELU into the
    1x1-convolution epilogue and the block tail; the object itself is only compared by identity."""
    raise NotImplementedError("fno.gelu names the activation fused into the kernels; it is not called")


_GELUS = (gelu, torch.nn.functional.gelu)


class Conv1x1(torch.nn.Module):
    """Parameters of a 1x1 convolution (nn.Conv2D(..., 1) in fno_block.py:286-291 / :203): weight [Co, Ci, 1, 1], bias
    [Co]; initialised like the framework default U(+-1/sqrt(Ci)).  Applied by ppsci_pw_conv (fno_engine)."""

    def __init__(self, in_channels: int, out_channels: int, bias: bool = True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        k = 1.0 / math.sqrt(in_channels)
        self.weight = torch.nn.Parameter((torch.rand(out_channels, in_channels, 1, 1) * 2 - 1) * k)
        self.bias = torch.nn.Parameter((torch.rand(out_channels) * 2 - 1) * k) if bias else None


class GroupNorm1(torch.nn.Module):
    """Parameters of nn.GroupNorm(num_groups=1, C) (fno_block.py:1157-1165): scale ones, shift zeros, eps 1e-5.  Applied
    inside ppsci_fno_tail_fwd / _bwd."""

    num_groups = 1

    def __init__(self, num_channels: int, eps: float = 1e-5):
        super().__init__()
        self.eps = eps
        self.weight = torch.nn.Parameter(torch.ones(num_channels))
        self.bias = torch.nn.Parameter(torch.zeros(num_channels))


class SpectralConv2d(torch.nn.Module):
    """Parameters of one layer of FactorizedSpectralConv (dense weights, order 2)."""

    def __init__(self, in_channels: int, out_channels: int, n_modes: Tuple[int, int], bias: bool = True,
                 fft_norm: str = "backward", init_std: Optional[float] = None):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.n_modes = (int(n_modes[0]), int(n_modes[1]) // 2 + 1)  # fno_block.py:673-684
        self.fft_norm = fft_norm  # (the rfftn / irfftn pair scales by 1/(H W) for every norm: folded into the contraction)
        std = (2 / (in_channels + out_channels)) ** 0.5 if init_std is None else init_std
        shape = (in_channels, out_channels, *self.n_modes)
        self.weight_real = torch.nn.Parameter(torch.randn(shape) * std)
        self.weight_imag = torch.nn.Parameter(torch.randn(shape) * std)
        self.bias = torch.nn.Parameter(std * torch.randn(out_channels, 1, 1)) if bias else None


class SphericalConv2d(torch.nn.Module):
    """Parameters of one layer of SphericalConv (/root/reference/ppsci/arch/sfnonet.py:183-360): complex weights PER DEGREE
    `[Ci, Co, n_modes[0]]` (`weight_shape = (in, out, *n_modes[:-1])`, :268-276; the contraction runs with dhconv=True), bias
    `[Co, 1, 1]`.  `n_modes` here = (degrees L, orders M) = (n_modes[0], n_modes[1] // 2), the `s=` of its sht call (:333-338)."""

    def __init__(self, in_channels: int, out_channels: int, n_modes: Tuple[int, int], bias: bool = True,
                 fft_norm: str = "backward", init_std: Optional[float] = None):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.n_modes = (int(n_modes[0]), int(n_modes[1]) // 2)
        std = (2 / (in_channels + out_channels)) ** 0.5 if init_std is None else init_std
        shape = (in_channels, out_channels, self.n_modes[0])
        self.weight_real = torch.nn.Parameter(torch.randn(shape) * std)
        self.weight_imag = torch.nn.Parameter(torch.randn(shape) * std)
        self.bias = torch.nn.Parameter(std * torch.randn(out_channels, 1, 1)) if bias else None


class ChannelMLP(torch.nn.Module):
    """fno_block.MLP (fno_block.py:263-320): n_layers 1x1 convolutions with the non-linearity in between."""

    def __init__(self, in_channels: int, out_channels: Optional[int] = None, hidden_channels: Optional[int] = None,
                 n_layers: int = 2, non_linearity=gelu):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        hidden_channels = in_channels if hidden_channels is None else hidden_channels
        self.n_layers = n_layers
        self.non_linearity = non_linearity
        dims = [in_channels] + [hidden_channels] * (n_layers - 1) + [out_channels]
        self.fcs = torch.nn.ModuleList([Conv1x1(dims[i], dims[i + 1]) for i in range(n_layers)])


class FNOBlocks(torch.nn.Module):
    """fno_block.FNOBlocks, post-activation form (fno_block.py:1191-1220): per layer
    x <- act( norm(SpectralConv_i(stab(x))) + skip_i(x) ), stab = tanh with `stabilizer="tanh"` (:1199), no activation
    after the last layer."""

    def __init__(self, in_channels: int, out_channels: int, n_modes: Tuple[int, int], n_layers: int = 1,
                 non_linearity=gelu, stabilizer: Optional[str] = None, norm: Optional[str] = None,
                 fno_skip: str = "linear", fft_norm: str = "forward", conv_cls=None):
        super().__init__()
        conv_cls = conv_cls or SpectralConv2d
        if in_channels != out_channels:
            raise NotImplementedError("FNOBlocks with in_channels != out_channels")
        if fno_skip not in ("linear", "identity"):
            raise NotImplementedError(f"fno_skip={fno_skip!r} (built: 'linear', 'identity')")
        if norm not in (None, "group_norm"):
            raise NotImplementedError(f"norm={norm!r}: the block-tail kernel normalises over one group (built: None, 'group_norm')")
        if stabilizer not in (None, "tanh"):
            raise ValueError(f"stabilizer={stabilizer!r}")
        if non_linearity not in _GELUS:
            raise NotImplementedError("non_linearity: GELU is the activation fused into the kernels")
        self.n_layers, self.non_linearity, self.stabilizer = n_layers, non_linearity, stabilizer
        # FactorizedSpectralConv holds the weights of all layers; bias per layer (fno_block.py:652-663)
        self.convs = torch.nn.ModuleList([conv_cls(in_channels, out_channels, n_modes, bias=True, fft_norm=fft_norm)
                                          for _ in range(n_layers)])
        self.fno_skips = torch.nn.ModuleList([
            Conv1x1(in_channels, out_channels, bias=False) if fno_skip == "linear" else torch.nn.Identity()
            for _ in range(n_layers)])
        self.norm = torch.nn.ModuleList([GroupNorm1(out_channels) for _ in range(n_layers)]) if norm == "group_norm" else None


class FNONet(base.Arch, torch.nn.Module):
    """ppsci.arch.FNONet for 2-D problems (tfnonet.py:13-193).  Input: dict with one `[B, C, H, W]` tensor per
    input key (concatenated along the channel axis, like `concat_to_tensor` with the reference's layout);
    output: `{output_keys[0]: [B, out_channels, H, W]}`.

    The parameters live in ONE flat fp32 buffer (`flat_params`, module parameters are views into it) so that
    the data-parallel all-reduce and the fused Adam kernel act on a single tensor, as for the PINN path."""

    is_operator = True  # Solver: the operator engine (hand-written forward + backward, fno_engine.FnoNative)
    _conv_cls = SpectralConv2d
    spectral = "fft"  # the transform pair of the spectral branch: "fft" (rfftn / irfftn) or "sht" (SFNONet)

    def __init__(self, input_keys: Tuple[str, ...], output_keys: Tuple[str, ...], n_modes: Tuple[int, ...],
                 hidden_channels: int, in_channels: int = 3, out_channels: int = 1, lifting_channels: int = 256,
                 projection_channels: int = 256, n_layers: int = 4, use_mlp: bool = False, mlp=None,
                 max_n_modes=None, non_linearity=gelu, stabilizer: Optional[str] = None,
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
                                    norm, fno_skip, fft_norm, self._conv_cls)
        if lifting_channels:
            self.lifting = ChannelMLP(in_channels, hidden_channels, lifting_channels, 2)
        else:
            self.lifting = ChannelMLP(in_channels, hidden_channels, hidden_channels, 1)
        if non_linearity not in _GELUS:
            raise NotImplementedError("non_linearity: GELU is the activation fused into the kernels")
        self.projection = ChannelMLP(hidden_channels, out_channels, projection_channels, 2, non_linearity)
        self.flat_params: Optional[torch.Tensor] = None
        self.flat_grad: Optional[torch.Tensor] = None
        from ..device import get_device

        self.to_device(get_device())

    # ---- flat parameter buffer ---------------------------------------------------------------
    def to_device(self, device):
        """Moves the model and (re)packs every parameter as a view into `flat_params` / `flat_grad`."""
        torch.nn.Module.to(self, device)
        self._native = None  # (its buffers live on the old device)
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

    def native(self):
        """The kernels' executor for this model (buffers per batch shape); shared by training, eval and predict."""
        nat = getattr(self, "_native", None)
        if nat is None:
            from ..fno_engine import FnoNative

            nat = self._native = FnoNative(self)
        return nat

    def forward_tensor(self, x: torch.Tensor) -> torch.Tensor:
        """[B, C_in, H, W] -> [B, C_out, H, W] (a fresh tensor; the executor owns its buffers)."""
        return self.native().forward(x.to(dtype=torch.float32).contiguous()).clone()

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
                 non_linearity=gelu, use_mlp: bool = False, mlp=None, norm: Optional[str] = None,
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


class SFNONet(FNONet):
    """ppsci.arch.SFNONet (/root/reference/ppsci/arch/sfnonet.py:390-568): an FNONet -- same constructor, same lifting / blocks /
    projection -- whose spectral convolution transforms with the spherical-harmonic pair of arch/paddle_harmonics (equiangular
    colatitude grid, orthonormal harmonics: the defaults SphericalConv is built with, sfnonet.py:227-228) and holds its complex
    weights per degree.  Input planes are [latitude (north to south), longitude]."""

    _conv_cls = SphericalConv2d
    spectral = "sht"
    sht_grid, sht_norm = "equiangular", "ortho"


class TFNO1dNet(FNONet):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("TFNO1dNet: only the 2-D spectral convolution has a HIP kernel")


class TFNO3dNet(FNONet):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("TFNO3dNet: only the 2-D spectral convolution has a HIP kernel")
