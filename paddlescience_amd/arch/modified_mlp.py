"""ppsci.arch.ModifiedMLP (/root/reference/ppsci/arch/mlp.py:318-528; Wang, Teng & Perdikaris 2020) on HIP kernels, layer by
layer with the machinery of arch/piratenet.py:

    x0 = [period-embedded inputs]  (or  [cos(B e) ; sin(B e)]  with `fourier`)
    U, V = act(W_u x0 + b_u), act(W_v x0 + b_v)
    y <- act(W_l y + b_l);  y <- y * U + (1 - y) * V        for every hidden layer (the first one reads x0)
    out = W_L y + b_L

Dense layers are `ppsci_pw_conv` GEMMs over all Taylor streams; bias + activation + gate and their reverse are
`ppsci_pirate_act_*` (csrc/pirate.hip).  Trainable tensors in the reference's `parameters()` order and names:
[fourier_emb.kernel], embed_u.0.*, embed_v.0.*, linears.i.*, last_fc.* (`weight`/`bias`, or `weight_v`/`weight_g`/`bias` with
`random_weight`).  Not available (raise): weight_norm, skip_connection, learnable activations, derivative order > 2, input
transforms.  (SPINN's one-input branch nets use their own fused kernels, csrc/spinn.hip.)"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional, Tuple, Union

import numpy as np
import torch

from .. import _lib as L
from .. import hotpath as hp
from ..device import get_device
from .piratenet import _ACTS, PirateExec, PirateLayout, PirateNet, _p, _sp


class ModifiedMLP(PirateNet):
    num_blocks = 0

    def __init__(
        self,
        input_keys: Tuple[str, ...],
        output_keys: Tuple[str, ...],
        num_layers: int,
        hidden_size: int,
        activation: str = "tanh",
        skip_connection: bool = False,
        weight_norm: bool = False,
        input_dim: Optional[int] = None,
        output_dim: Optional[int] = None,
        periods: Optional[Dict[str, Tuple[float, bool]]] = None,
        fourier: Optional[Dict[str, Union[float, int]]] = None,
        random_weight: Optional[Dict[str, float]] = None,
    ):
        from . import activation as act_mod
        from .base import Arch

        Arch.__init__(self)
        self.input_keys, self.output_keys = tuple(input_keys), tuple(output_keys)
        if not isinstance(hidden_size, int):
            raise ValueError(f"hidden_size should be int, but got {type(hidden_size)}")  # mlp.py:374-375
        if not isinstance(num_layers, int):
            raise ValueError("num_layers should be an int")  # mlp.py:371-372
        if weight_norm or skip_connection:
            raise NotImplementedError("ModifiedMLP(weight_norm / skip_connection) has no HIP kernel path")
        if input_dim is not None and int(input_dim) != len(self.input_keys):
            raise NotImplementedError("multi-column inputs (input_dim != number of input keys)")
        if output_dim is not None and int(output_dim) != len(self.output_keys):
            raise NotImplementedError("multi-column outputs (output_dim != number of output keys)")
        self.activation = act_mod.get_activation(activation)
        if self.activation not in _ACTS:
            raise NotImplementedError(f"ModifiedMLP activation {activation!r}: the stream kernels carry {_ACTS}")
        if len(self.input_keys) > L.MAX_IN or len(self.output_keys) > L.MAX_OUT:
            raise NotImplementedError(f"at most {L.MAX_IN} inputs / {L.MAX_OUT} outputs")
        self.hidden, self.num_gated_layers = int(hidden_size), int(num_layers)
        self.periods, self.fourier = periods, fourier
        self._rwf = dict(random_weight) if random_weight else None
        self._embed = [L.EMBED_NONE] * len(self.input_keys)
        self._omega = [0.0] * len(self.input_keys)
        if periods:
            from .mlp import PeriodEmbedding

            self.period_emb = PeriodEmbedding(periods)
            for k, w in self.period_emb.freqs_dict.items():
                j = self.input_keys.index(k)
                self._embed[j], self._omega[j] = L.EMBED_PERIOD, w
        self.d0 = len(self.input_keys) + sum(1 for e in self._embed if e == L.EMBED_PERIOD)
        self.half = 0
        if fourier:
            if int(fourier["dim"]) % 2 != 0:
                raise ValueError(f"out_features must be even, but got {fourier['dim']}.")  # mlp.py:120-121
            self.half = int(fourier["dim"]) // 2
        self.c0 = 2 * self.half if self.half else self.d0  # width of x0
        H, m = self.hidden, len(self.output_keys)

        def lin(name, fin, fout):
            if self._rwf:
                return [(f"{name}.weight_v", (fin, fout)), (f"{name}.weight_g", (fout,)), (f"{name}.bias", (fout,))]
            return [(f"{name}.weight", (fin, fout)), (f"{name}.bias", (fout,))]

        shapes: List[Tuple[str, Tuple[int, ...]]] = [("fourier_emb.kernel", (self.d0, self.half))] if self.half else []
        shapes += lin("embed_u.0", self.c0, H) + lin("embed_v.0", self.c0, H)
        for i in range(self.num_gated_layers):
            shapes += lin(f"linears.{i}", self.c0 if i == 0 else H, H)
        shapes += lin("last_fc", H, m)
        self._shapes = shapes
        self.reparam = False
        self._bind_views(torch.zeros(sum(int(np.prod(s_)) for _, s_ in shapes), dtype=torch.float32, device=get_device()))
        self.layout = ModifiedLayout(self)
        self._frozen = False
        self._init_parameters()
        self._predict_exec: Dict[int, PirateExec] = {}

    def linear_names(self) -> List[str]:
        return ["embed_u.0", "embed_v.0"] + [f"linears.{i}" for i in range(self.num_gated_layers)] + ["last_fc"]

    def _init_parameters(self):
        """FourierEmbedding Normal(std=scale); nn.Linear Xavier-uniform / zero bias; RandomWeightFactorization as in PirateNet."""
        t = self._byname
        if self.half:
            k = t["fourier_emb.kernel"]
            k.copy_(torch.from_numpy(np.random.normal(0.0, float(self.fourier["scale"]), size=tuple(k.shape)).astype(np.float32)))
        for name in self.linear_names():
            w = t[name + (".weight_v" if self._rwf else ".weight")]
            fin, fout = w.shape
            if self._rwf:
                vv = np.random.normal(0.0, math.sqrt(2.0 / (fin + fout)), size=(fin, fout)).astype(np.float32)
                gg = np.exp(np.random.normal(self._rwf["mean"], self._rwf["std"], size=(fout,)).astype(np.float32))
                w.copy_(torch.from_numpy(vv / gg))
                t[name + ".weight_g"].copy_(torch.from_numpy(gg))
            else:
                lim = math.sqrt(6.0 / (fin + fout))
                w.copy_(torch.from_numpy(np.random.uniform(-lim, lim, size=(fin, fout)).astype(np.float32)))
            t[name + ".bias"].zero_()

    def _make_exec(self, spec, n, inputs, train=True):
        return ModifiedExec(self, spec, n, inputs, train)

    def _forward_numeric(self, x):
        dev = self.flat_params.device
        ins = []
        for k in self.input_keys:
            v = x[k]
            if not isinstance(v, torch.Tensor):
                v = torch.as_tensor(np.asarray(v), dtype=torch.float32)
            ins.append(v.to(device=dev, dtype=torch.float32).contiguous().view(-1))
        n = ins[0].numel()
        ex = self._predict_exec.get(n)
        if ex is None:
            if len(self._predict_exec) > 4:
                self._predict_exec.clear()
            ex = self._predict_exec[n] = ModifiedExec(self, hp.StreamSpec([], 0), n, [torch.empty_like(t) for t in ins], train=False)
        for dst, src in zip(ex.inputs, ins):
            dst.copy_(src)
        U = torch.empty((len(self.output_keys), n), dtype=torch.float32, device=dev)
        ex.forward(self.flat_params, U, False)
        return {k: U[i].view(n, 1) for i, k in enumerate(self.output_keys)}


class ModifiedLayout(PirateLayout):
    def __init__(self, model: ModifiedMLP):
        self.model = model
        self.d_raw, self.d_out = len(model.input_keys), len(model.output_keys)
        self.n_hidden, self.width = model.num_gated_layers, model.hidden
        self.embed, self.omega = model._embed, model._omega

    def make_exec(self, spec, n, inputs):
        return ModifiedExec(self.model, spec, n, inputs)


class ModifiedExec(PirateExec):
    """forward / backward launch sequences of ModifiedMLP over PirateExec's buffers and helpers."""

    def forward(self, params: torch.Tensor, Urows: torch.Tensor, train: bool) -> None:
        m, H, c0 = self.model, self.H, self.c0
        lib = L.lib()
        self._materialize(params)
        kern = self._t(params, "fourier_emb.kernel") if m.half else None
        L.check(lib.ppsci_pirate_embed_fwd(C.byref(self.desc), self._in_ptrs, _p(kern), _p(self.X0), _sp(self.X0)))
        self._dense(self.X0, self._w(params, "embed_u.0"), c0, H, self.ZU)
        self._act_fwd(L.PIRATE_ACT, self.ZU, self._t(params, "embed_u.0.bias"), self.U)
        self._dense(self.X0, self._w(params, "embed_v.0"), c0, H, self.ZV)
        self._act_fwd(L.PIRATE_ACT, self.ZV, self._t(params, "embed_v.0.bias"), self.V)
        y, fin = self.X0, c0
        for i, lay in enumerate(self.layers):
            self._dense(y, self._w(params, f"linears.{i}"), fin, H, lay["Z"])
            self._act_fwd(L.PIRATE_GATE, lay["Z"], self._t(params, f"linears.{i}.bias"), lay["O"], U=self.U, V=self.V)
            y, fin = lay["O"], H
        self._dense(y, self._w(params, "last_fc"), fin, self.m, self.Y)
        L.check(lib.ppsci_pirate_out_fwd(self.S, self.m, self.n, self.NP, _p(self.Y), _p(self._t(params, "last_fc.bias")),
                                         _p(Urows), _sp(Urows)))

    def backward(self, params: torch.Tensor, Ubar_rows: torch.Tensor, grad: torch.Tensor) -> None:
        if not self._train_ready:
            self._alloc_train()
        m, H, c0, lib = self.model, self.H, self.c0, L.lib()
        grad = grad.view(-1)
        nl = len(self.layers)
        self._pcall, self._psegs, self._pullbacks = 0, [], []  # (reductions are summed at the end: PirateExec._flush_sums)
        L.check(lib.ppsci_pirate_out_bwd(self.S, self.m, self.n, self.NP, _p(Ubar_rows), _p(self.Ybar), _sp(self.Ybar)))
        ob, _ = m._offsets["last_fc.bias"]
        for o in range(self.m):
            self._sum_later(Ubar_rows[o * self.S], self.n, 1, grad[ob + o:ob + o + 1])
        ylast, flast = (self.layers[-1]["O"], H) if nl else (self.X0, c0)
        self._wgrad(ylast, self.Ybar, flast, self.m, "last_fc", params, grad)
        xb0 = self.XB0 if self.XB0 is not None else self.XB[1]  # adjoint of x0
        ybar = self.OB if nl else xb0
        self._dense_t(self.Ybar, self._w(params, "last_fc"), flast, self.m, ybar)
        self.UB.zero_()
        self.VB.zero_()
        wrote_x0 = nl == 0
        for i in range(nl - 1, -1, -1):
            lay = self.layers[i]
            yin, fin = (self.layers[i - 1]["O"], H) if i > 0 else (self.X0, c0)
            self._act_bwd(L.PIRATE_GATE, lay["Z"], f"linears.{i}.bias", self.OB, params, grad, U=self.U, V=self.V)
            self._wgrad(yin, self.ZB, fin, H, f"linears.{i}", params, grad)
            if i > 0:
                self._dense_t(self.ZB, self._w(params, f"linears.{i}"), fin, H, self.OB)
            elif m.half:  # the adjoint of x0 is only needed for the Fourier kernel's gradient
                self._dense_t(self.ZB, self._w(params, f"linears.{i}"), fin, H, xb0)
                wrote_x0 = True
        for name, Z, B_ in (("embed_u.0", self.ZU, self.UB), ("embed_v.0", self.ZV, self.VB)):
            self._act_bwd(L.PIRATE_ACT, Z, name + ".bias", B_, params, grad)
            self._wgrad(self.X0, self.ZB, c0, H, name, params, grad)
            if m.half:
                self._dense_t(self.ZB, self._w(params, name), c0, H, xb0, accumulate=wrote_x0)
                wrote_x0 = True
        if m.half:
            L.check(lib.ppsci_pirate_embed_bwd(C.byref(self.desc), self._in_ptrs, _p(self._t(params, "fourier_emb.kernel")),
                                               _p(xb0), _p(self.pB), _sp(self.pB)))
            ok, nk = m._offsets["fourier_emb.kernel"]
            self._sum_later(self.pB, self.echunks, nk, grad[ok:ok + nk])
        self._flush_sums()
