"""ppsci.arch.SPINN (/root/reference/ppsci/arch/spinn.py:29-180) with ModifiedMLP branch nets
(/root/reference/ppsci/arch/mlp.py:318-527) on the HIP kernels of csrc/spinn.hip.

One branch net per input axis maps a coordinate [N_a,1] to r*m features; the output on the tensor-product
grid is u[i,j,k] = sum_r fx[i,r] fy[j,r] fz[k,r] (spinn.py:140-167).  All parameters live in one flat fp32
buffer (branch 0, branch 1, branch 2; inside a branch in `parameters()` order).  Linear layers are
re-initialised glorot-normal with zero bias like SPINN._init_weights (spinn.py:107-111).

Traced use (expression compilation) returns `GridLinear` proxies: linear combinations of
{u, u_xx, u_yy, u_zz}, which is what Helmholtz / Poisson-type residuals on a separable net need."""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional, Tuple, Union

import numpy as np
import torch

from .. import _lib as L
from ..device import get_device
from ..utils import initializer
from ..graph import Sym
from ..hotpath import _p, _stream_ptr
from . import activation as act_mod
from .base import Arch


class GridLinear:
    """cu*u + cxx*u_xx + cyy*u_yy + czz*u_zz on the tensor-product grid of a SPINN."""

    def __init__(self, model, cu=0.0, cxx=0.0, cyy=0.0, czz=0.0):
        self.model, self.c = model, np.array([cu, cxx, cyy, czz], dtype=np.float64)

    def _new(self, c):
        g = GridLinear(self.model)
        g.c = c
        return g

    def __add__(self, o):
        if isinstance(o, GridLinear):
            return self._new(self.c + o.c)
        if isinstance(o, (int, float)) and o == 0:
            return self
        raise NotImplementedError("only linear combinations of u, u_xx, u_yy, u_zz are fused for SPINN")

    __radd__ = __add__

    def __sub__(self, o):
        return self + (-1.0) * o

    def __mul__(self, k):
        if not isinstance(k, (int, float, np.floating)):
            raise NotImplementedError("SPINN residuals must be linear in u and its second derivatives")
        return self._new(self.c * float(k))

    __rmul__ = __mul__

    def __neg__(self):
        return self * -1.0


class ModifiedMLPSpec:
    """Shape of a one-input ModifiedMLP branch as the kernels see it."""

    def __init__(self, num_layers: int, hidden_size: int, d_out: int, activation: str):
        self.desc = L.ModMlpDesc()
        self.desc.n_hidden, self.desc.width, self.desc.d_out = num_layers, hidden_size, d_out
        self.desc.activation = L.ACT[activation]
        self.L, self.H, self.R = num_layers, hidden_size, d_out

    def param_shapes(self) -> List[Tuple[str, Tuple[int, ...]]]:
        H, out = self.H, []
        out += [("embed_u.0.weight", (1, H)), ("embed_u.0.bias", (H,)), ("embed_v.0.weight", (1, H)), ("embed_v.0.bias", (H,))]
        fin = 1
        for l in range(self.L):
            out += [(f"linears.{l}.weight", (fin, H)), (f"linears.{l}.bias", (H,))]
            fin = H
        out += [("last_fc.weight", (H, self.R)), ("last_fc.bias", (self.R,))]
        return out

    @property
    def n_params(self) -> int:
        return int(sum(int(np.prod(s)) for _, s in self.param_shapes()))


class SPINN(Arch):
    def __init__(self, input_keys: Tuple[str, ...], output_keys: Tuple[str, ...], r: int, num_layers: int,
                 hidden_size: Union[int, Tuple[int, ...]], activation: str = "tanh", skip_connection: bool = False,
                 weight_norm: bool = False, periods=None, fourier=None, random_weight=None):
        super().__init__()
        if len(input_keys) != 3 or len(output_keys) != 1:
            raise NotImplementedError("the HIP SPINN path covers 3 input axes and one output (Helmholtz3D)")
        if skip_connection or weight_norm or periods or fourier or random_weight or not isinstance(hidden_size, int):
            raise NotImplementedError("SPINN options beyond plain ModifiedMLP branches have no HIP kernel yet")
        self.input_keys, self.output_keys, self.r = tuple(input_keys), tuple(output_keys), r
        self.activation = act_mod.get_activation(activation)
        if self.activation not in ("tanh", "silu", "sin"):
            raise NotImplementedError(f"SPINN branch nets: activation {activation!r} has no HIP kernel (tanh, silu, sin)")
        self.spec = ModifiedMLPSpec(num_layers, hidden_size, r * len(output_keys), self.activation)
        self.branch_params = self.spec.n_params
        self.flat_params = torch.zeros(3 * self.branch_params, dtype=torch.float32, device=get_device())
        self._names, self._views = [], []
        for b in range(3):
            off = b * self.branch_params
            for name, shp in self.spec.param_shapes():
                n = int(np.prod(shp))
                self._names.append(f"branch_nets.{b}.{name}")
                v = self.flat_params[off:off + n].view(*shp)
                self._views.append(v)
                if len(shp) == 2:  # SPINN._init_weights (spinn.py:107-111): glorot_normal_ weights, zero biases
                    initializer.glorot_normal_(v)
                off += n

    def branch(self, b: int) -> torch.Tensor:
        return self.flat_params[b * self.branch_params:(b + 1) * self.branch_params]

    def parameters(self):
        return list(self._views)

    def state_dict(self):
        return dict(zip(self._names, self._views))

    def set_state_dict(self, state):
        for n, v in zip(self._names, self._views):
            if n in state:
                v.copy_(torch.as_tensor(np.asarray(state[n]), dtype=torch.float32))
        return [n for n in self._names if n not in state], [n for n in state if n not in self._names]

    # ---- kernels
    def branch_forward(self, b: int, x: torch.Tensor, stash: Optional[torch.Tensor] = None) -> torch.Tensor:
        n = x.numel()
        F = torch.empty((3, n, self.spec.R), dtype=torch.float32, device=x.device)
        L.check(L.lib().ppsci_modmlp_fwd(C.byref(self.spec.desc), _p(self.branch(b)), n, _p(x), _p(F), _p(stash),
                                         _stream_ptr(x)))
        return F

    def _coords(self, x: Dict[str, object]) -> List[torch.Tensor]:
        dev = self.flat_params.device
        out = []
        for k in self.input_keys:
            v = x[k]
            v = v if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v, dtype=np.float32))
            out.append(v.to(device=dev, dtype=torch.float32).contiguous().view(-1))
        return out

    def forward(self, x: Dict[str, object]):
        if any(isinstance(v, Sym) for v in x.values()):
            return {self.output_keys[0]: GridLinear(self, cu=1.0)}
        xs = self._coords(x)
        Fs = [self.branch_forward(b, xs[b]) for b in range(3)]
        d = L.SpinnGridDesc()
        d.n[0], d.n[1], d.n[2] = (t.numel() for t in xs)
        d.rank, d.cu, d.cxx, d.cyy, d.czz, d.scale = self.spec.R, 1.0, 0.0, 0.0, 0.0, 0.0
        total = xs[0].numel() * xs[1].numel() * xs[2].numel()
        res = torch.empty(total, dtype=torch.float32, device=xs[0].device)
        part = torch.empty(int(L.lib().ppsci_spinn_grid_partial_rows(C.byref(d))), dtype=torch.float32, device=res.device)
        L.check(L.lib().ppsci_spinn_grid_fwd(C.byref(d), _p(Fs[0]), _p(Fs[1]), _p(Fs[2]), None, _p(res), None, _p(part),
                                             _stream_ptr(res)))
        u = res.view(xs[0].numel(), xs[1].numel(), xs[2].numel(), 1)
        out = {self.output_keys[0]: u}
        if self._output_transform is not None:
            out = self._output_transform(x, out)
        return out

    def forward_tensor(self, x, y, z):
        return [self.forward({self.input_keys[0]: x, self.input_keys[1]: y, self.input_keys[2]: z})[self.output_keys[0]]]

    def second_derivative(self, axis_key: str) -> GridLinear:
        i = self.input_keys.index(axis_key)
        c = [0.0, 0.0, 0.0, 0.0]
        c[1 + i] = 1.0
        return GridLinear(self, *c)
