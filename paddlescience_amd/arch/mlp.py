"""ppsci.arch.MLP (/root/reference/ppsci/arch/mlp.py:139-315) on the fused HIP kernels.

Parameters live in ONE flat fp32 device buffer in `model.parameters()` order
(linears.0.weight [in,out], linears.0.bias, ..., last_fc.weight, last_fc.bias -- mlp.py:264-277);
`linears[i].weight` etc. are views into it, so the optimizer and the kernels see the same memory.

Calling the model
  * with traced inputs (graph.Sym, during expression compilation) returns traced network outputs;
  * with tensors / numpy arrays runs the forward kernel (no derivative streams) and returns
    [N, 1] tensors per output key -- the reference's eager `model(input_dict)`.
Options without a HIP kernel yet (weight_norm, fourier, random_weight, stan/siren/...,
per-layer widths, input/output transforms on the fused path) raise NotImplementedError."""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple, Union

import numpy as np
import torch

from .. import _lib as L
from .. import hotpath as hp
from ..device import get_device
from ..graph import Sym
from . import activation as act_mod
from .base import Arch


class _Linear:
    """View of one nn.Linear inside the flat parameter buffer (weight is [in, out])."""

    def __init__(self, weight: torch.Tensor, bias: torch.Tensor):
        self.weight, self.bias = weight, bias

    def parameters(self):
        return [self.weight, self.bias]


class PeriodEmbedding:
    """mlp.py:95-114: x_k -> [cos(w x_k), sin(w x_k)], w = 2*pi/period (non-trainable on the HIP path)."""

    def __init__(self, periods: Dict[str, Tuple[float, bool]]):
        for k, (p, trainable) in periods.items():
            if trainable:
                raise NotImplementedError("trainable period embedding has no fused HIP kernel yet")
        self.freqs_dict = {k: float(np.float32(2 * np.pi / float(p))) for k, (p, _) in periods.items()}


class MLP(Arch):
    def __init__(
        self,
        input_keys: Tuple[str, ...],
        output_keys: Tuple[str, ...],
        num_layers: Optional[int],
        hidden_size: Union[int, Tuple[int, ...]],
        activation: str = "tanh",
        skip_connection: bool = False,
        weight_norm: bool = False,
        input_dim: Optional[int] = None,
        output_dim: Optional[int] = None,
        periods: Optional[Dict[str, Tuple[float, bool]]] = None,
        fourier: Optional[Dict[str, Union[float, int]]] = None,
        random_weight: Optional[Dict[str, float]] = None,
    ):
        super().__init__()
        self.input_keys = tuple(input_keys)
        self.output_keys = tuple(output_keys)
        if isinstance(hidden_size, (tuple, list)):
            if num_layers is not None:
                raise ValueError("num_layers should be None when hidden_size is specified")
            hidden = list(hidden_size)
        elif isinstance(hidden_size, int):
            if not isinstance(num_layers, int):
                raise ValueError("num_layers should be an int when hidden_size is an int")
            hidden = [hidden_size] * num_layers
        else:
            raise ValueError(f"hidden_size should be list of int or int, but got {type(hidden_size)}")
        if weight_norm or fourier or random_weight:
            raise NotImplementedError("weight_norm / fourier / random_weight have no fused HIP kernel yet")
        if input_dim is not None or output_dim is not None:
            raise NotImplementedError("input_dim / output_dim overrides are not supported on the HIP path")
        if len(set(hidden)) != 1:
            raise NotImplementedError("the fused HIP kernels need one width for all hidden layers")
        self.activation = act_mod.get_activation(activation)
        self.skip_connection = bool(skip_connection)
        self.periods = periods
        self.fourier = None
        embed = [L.EMBED_NONE] * len(self.input_keys)
        omega = [0.0] * len(self.input_keys)
        if periods:
            self.period_emb = PeriodEmbedding(periods)
            for k, w in self.period_emb.freqs_dict.items():
                j = self.input_keys.index(k)
                embed[j], omega[j] = L.EMBED_PERIOD, w
        self.layout = hp.NetLayout(len(self.input_keys), len(hidden), hidden[0], len(self.output_keys),
                                   self.activation, self.skip_connection, embed, omega)
        self.flat_params = torch.zeros(self.layout.n_params, dtype=torch.float32, device=get_device())
        self._names: List[str] = []
        self._views: List[torch.Tensor] = []
        off = 0
        for name, shp in self.layout.param_shapes():
            n = int(np.prod(shp))
            self._names.append(name)
            self._views.append(self.flat_params[off:off + n].view(*shp))
            off += n
        self.linears = [_Linear(self._views[2 * i], self._views[2 * i + 1]) for i in range(len(hidden))]
        self.last_fc = _Linear(self._views[-2], self._views[-1])
        self.acts = [self.activation] * len(hidden)
        self._frozen = False
        self._init_parameters()

    # ---- parameters
    def _init_parameters(self):
        """paddle nn.Linear default: Xavier-uniform weight, zero bias (Paddle behaviour);
        drawn from numpy's global RNG so that ppsci.utils.misc.set_random_seed controls it."""
        layers = [(l.weight, l.bias) for l in self.linears] + [(self.last_fc.weight, self.last_fc.bias)]
        for i, (w, b) in enumerate(layers):
            fin, fout = w.shape
            lim = math.sqrt(6.0 / (fin + fout))
            if self.activation == "siren" and i < len(self.linears):
                # mlp.py:256-260 / activation.py:106-136: first layer U(-1/in, 1/in), hidden U(+-sqrt(6/in)/w0)
                lim = 1.0 / fin if i == 0 else math.sqrt(6.0 / fin) / L.SIREN_W0
            w.copy_(torch.from_numpy(np.random.uniform(-lim, lim, size=(fin, fout)).astype(np.float32)))
            b.zero_()

    def parameters(self) -> List[torch.Tensor]:
        return list(self._views)

    def named_parameters(self):
        return list(zip(self._names, self._views))

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return {n: v for n, v in zip(self._names, self._views)}

    def set_state_dict(self, state: Dict[str, "np.ndarray"]):
        missing = [n for n in self._names if n not in state]
        unexpected = [n for n in state if n not in self._names]
        for n, v in zip(self._names, self._views):
            if n in state:
                src = state[n]
                src = torch.as_tensor(np.asarray(src.detach().cpu() if isinstance(src, torch.Tensor) else src),
                                      dtype=torch.float32)
                if tuple(src.shape) != tuple(v.shape):
                    raise ValueError(f"shape mismatch for {n}: {tuple(src.shape)} vs {tuple(v.shape)}")
                v.copy_(src)
        return missing, unexpected

    # ---- forward
    def forward_tensor(self, x: torch.Tensor) -> torch.Tensor:  # mlp.py:281-296 (on already-embedded input)
        if self.periods:
            raise NotImplementedError("forward_tensor on an embedded tensor: call the model with a dict instead")
        cols = {k: x[:, i:i + 1] for i, k in enumerate(self.input_keys)}
        out = self._forward_numeric(cols)
        return torch.cat([out[k] for k in self.output_keys], dim=-1)

    def _forward_numeric(self, x: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        dev = self.flat_params.device
        ins = []
        for k in self.input_keys:
            v = x[k]
            if not isinstance(v, torch.Tensor):
                v = torch.as_tensor(np.asarray(v), dtype=torch.float32)
            ins.append(v.to(device=dev, dtype=torch.float32).contiguous().view(-1))
        n = ins[0].numel()
        desc = self.layout.desc(hp.StreamSpec([], 0))
        U = torch.empty((len(self.output_keys), n), dtype=torch.float32, device=dev)
        hp.taylor_fwd(desc, self.flat_params, ins, U, None)
        return {k: U[i].view(n, 1) for i, k in enumerate(self.output_keys)}

    def forward(self, x: Dict[str, object]) -> Dict[str, object]:  # mlp.py:298-315
        traced = any(isinstance(v, Sym) for v in x.values())
        if self._input_transform is not None:
            raise NotImplementedError("input transforms are not lowered to the fused HIP path yet")
        if traced:
            for k in self.input_keys:
                v = x[k]
                if not (isinstance(v, Sym) and v.kind == "in" and v.name == k):
                    raise NotImplementedError(f"network input {k!r} must be the raw variable of the data dict")
            y = {k: Sym.net(self, i) for i, k in enumerate(self.output_keys)}
        else:
            y = self._forward_numeric(x)
        if self._output_transform is not None:
            y = self._output_transform(x, y)
        return y
