"""ppsci.arch.UNONet: the U-shaped neural operator of /root/reference/ppsci/arch/unonet.py on this
framework's kernels.

Like arch/fno.py the class holds PARAMETERS only, in the reference's module tree: `lifting` (fno_block.MLP), one single-layer
`fno_blocks[i]` per Fourier layer (FNOBlocks(in_i, uno_out_channels[i], uno_n_modes[i], output_scaling_factor=uno_scalings[i]),
unonet.py:205-230), `horizontal_skips[str(a)]` for every source layer a of `horizontal_skips_map` (a bias-free 1x1 convolution,
unonet.py:232-238), `projection`.  Forward and backward run in `uno_engine.UnoNative`: the FNO block kernels plus the two
resolution-changing operations of csrc/uno.hip.

What the reference's forward does (unonet.py:246-289), for reading the executor against:

    x = lifting(x); x = pad(x)
    for i: if i in skips_map: x = concat(x, bicubic(skip_out[skips_map[i]] -> x's grid))
           x = block_i(x)          # norm(SpectralConv(x) on the scaled grid) + bicubic(skip_conv(x)); NO activation: each
                                   # FNOBlocks has n_layers = 1, and `index < n_layers - 1` never holds (fno_block.py:1207)
           if i in skips_map.values(): skip_out[i] = horizontal_skips[i](x)
    x = unpad(x); y = projection(x)
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from . import base
from . import fno as F


class UNOBlock(torch.nn.Module):
    """One `fno_block.FNOBlocks(n_layers=1)` of the UNO: parameters under the same attribute names as arch/fno.FNOBlocks."""

    def __init__(self, in_channels: int, out_channels: int, n_modes, scaling, norm: Optional[str], fno_skip: str, fft_norm: str):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.scaling = tuple(float(s) for s in scaling)
        self.n_layers, self.stabilizer = 1, None
        self.convs = torch.nn.ModuleList([F.SpectralConv2d(in_channels, out_channels, n_modes, bias=True, fft_norm=fft_norm)])
        self.fno_skips = torch.nn.ModuleList([F.Conv1x1(in_channels, out_channels, bias=False) if fno_skip == "linear"
                                              else torch.nn.Identity()])
        self.norm = torch.nn.ModuleList([F.GroupNorm1(out_channels)]) if norm == "group_norm" else None

    def out_shape(self, H: int, W: int) -> Tuple[int, int]:
        """round(size * factor), Python's round (fno_block.py:684-690, :779-786)."""
        return round(H * self.scaling[0]), round(W * self.scaling[1])


class UNONet(F.FNONet):
    """ppsci.arch.UNONet for 2-D problems (unonet.py:14-289); constructor arguments in the reference's order.  The flat
    parameter buffer, state dict, `forward` on dicts etc. are FNONet's."""

    def __init__(self, input_keys: Tuple[str, ...], output_keys: Tuple[str, ...], in_channels: int, out_channels: int,
                 hidden_channels: int, lifting_channels: int = 256, projection_channels: int = 256, n_layers: int = 4,
                 uno_out_channels=None, uno_n_modes=None, uno_scalings=None, horizontal_skips_map: Optional[Dict] = None,
                 incremental_n_modes=None, use_mlp: bool = False, mlp=None, non_linearity=F.gelu, norm: Optional[str] = None,
                 ada_in_features=None, preactivation: bool = False, fno_skip: str = "linear", horizontal_skip: str = "linear",
                 mlp_skip: str = "soft-gating", separable: bool = False, factorization: Optional[str] = None,
                 rank: float = 1.0, joint_factorization: bool = False, implementation: str = "factorized",
                 domain_padding=None, domain_padding_mode: str = "one-sided", fft_norm: str = "forward",
                 patching_levels: int = 0, **kwargs):
        torch.nn.Module.__init__(self)
        base.Arch.__init__(self)
        if uno_out_channels is None:
            raise ValueError("uno_out_channels can not be None")
        if uno_n_modes is None:
            raise ValueError("uno_n_modes can not be None")
        if uno_scalings is None:
            raise ValueError("uno_scalings can not be None")
        if len(uno_out_channels) != n_layers:
            raise ValueError("Output channels for all layers are not given")
        if len(uno_n_modes) != n_layers:
            raise ValueError("Number of modes for all layers are not given")
        if len(uno_scalings) != n_layers:
            raise ValueError("Scaling factor for all layers are not given")
        if len(uno_n_modes[0]) != 2:
            raise NotImplementedError("only the 2-D spectral convolution has a HIP kernel")
        for name, val, ok in (("use_mlp", use_mlp, False), ("preactivation", preactivation, False),
                              ("separable", separable, False), ("joint_factorization", joint_factorization, False),
                              ("patching_levels", patching_levels, 0), ("ada_in_features", ada_in_features, None),
                              ("incremental_n_modes", incremental_n_modes, None)):
            if val != ok:
                raise NotImplementedError(f"UNONet({name}={val!r}) is not built")
        if non_linearity not in F._GELUS:
            raise NotImplementedError("non_linearity: GELU is the activation fused into the kernels")
        if norm not in (None, "group_norm"):
            raise NotImplementedError(f"norm={norm!r}: the block-tail kernel normalises over one group (built: None, 'group_norm')")
        if fno_skip not in ("linear", "identity"):
            raise NotImplementedError(f"fno_skip={fno_skip!r} (built: 'linear', 'identity')")
        if horizontal_skip != "linear":
            raise NotImplementedError(f"horizontal_skip={horizontal_skip!r} (built: 'linear')")
        if fft_norm not in ("forward", "backward", "ortho"):
            raise ValueError(f"fft_norm={fft_norm!r}")
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
        self.input_keys, self.output_keys = tuple(input_keys), tuple(output_keys)
        self.in_channels, self.out_channels, self.hidden_channels = in_channels, out_channels, hidden_channels
        self.n_layers, self.fft_norm = n_layers, fft_norm
        self.uno_out_channels = [int(c) for c in uno_out_channels]
        self.uno_n_modes = [tuple(int(v) for v in m) for m in uno_n_modes]
        self.uno_scalings = [tuple(float(v) for v in s) for s in uno_scalings]
        if horizontal_skips_map is None:  # unonet.py:159-165
            horizontal_skips_map = {n_layers - i - 1: i for i in range(n_layers // 2)}
        self.horizontal_skips_map = {int(k): int(v) for k, v in dict(horizontal_skips_map).items()}
        for dst, src in self.horizontal_skips_map.items():
            if not 0 <= src < dst < n_layers:
                raise ValueError(f"horizontal_skips_map: {dst}: {src} does not point from an earlier layer to a later one")
        e2e = [1.0, 1.0]
        for s in self.uno_scalings:  # unonet.py:167-171
            e2e = [a * b for a, b in zip(e2e, s)]
        self.end_to_end_scaling_factor = e2e
        self.lifting = F.ChannelMLP(in_channels, hidden_channels, lifting_channels, 2)
        blocks = []
        self.horizontal_skips = torch.nn.ModuleDict()
        prev = hidden_channels
        for i in range(n_layers):
            if i in self.horizontal_skips_map:
                prev += self.uno_out_channels[self.horizontal_skips_map[i]]
            blocks.append(UNOBlock(prev, self.uno_out_channels[i], self.uno_n_modes[i], self.uno_scalings[i], norm, fno_skip,
                                   fft_norm))
            if i in self.horizontal_skips_map.values():
                self.horizontal_skips[str(i)] = F.Conv1x1(self.uno_out_channels[i], self.uno_out_channels[i], bias=False)
            prev = self.uno_out_channels[i]
        self.fno_blocks = torch.nn.ModuleList(blocks)
        self.projection = F.ChannelMLP(prev, out_channels, projection_channels, 2, non_linearity)
        self.flat_params = self.flat_grad = None
        from ..device import get_device

        self.to_device(get_device())

    def native(self):
        nat = getattr(self, "_native", None)
        if nat is None:
            from ..uno_engine import UnoNative

            nat = self._native = UnoNative(self)
        return nat
