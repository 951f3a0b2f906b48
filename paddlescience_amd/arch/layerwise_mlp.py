"""ppsci.arch.MLP configurations OUTSIDE the envelope of the fused single-kernel sweep (taylor_fwd / taylor_bwd: hidden width
<= 256, `fourier.dim == hidden_size`, per-layer widths only with plain layers), run layer by layer on Taylor streams with
the machinery of arch/piratenet.py: every dense layer is one MFMA GEMM over all streams (`ppsci_pw_conv`), bias + activation
and their reverse are `ppsci_pirate_act_*` (mode ACT), the period / Fourier embedding is `ppsci_pirate_embed_*`.

`ppsci.arch.MLP(...)` returns an instance of this class for

    * a hidden width above 256 (any width whose 16-row weight block fits LDS: up to 2 560),
    * a Fourier embedding whose dim differs from hidden_size, or with an activation other than tanh,
    * per-layer widths together with `random_weight` or `fourier`,

with the reference's parameter names and order (mlp.py:196-277): [fourier_emb.kernel], linears.i.{weight,bias} or
{weight_v,weight_g,bias}, last_fc.*.  Derivative orders 0-2; not available here either: weight_norm, skip_connection, learnable
activations (stan / swish), siren, input transforms."""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional, Tuple, Union

import numpy as np
import torch

from .. import _lib as L
from .. import hotpath as hp
from ..device import get_device
from .piratenet import _ACTS, PirateLayout, PirateNet, _p, _sp


def wants_layerwise(num_layers, hidden_size, activation="tanh", skip_connection=False, weight_norm=False, input_dim=None,
                    output_dim=None, periods=None, fourier=None, random_weight=None) -> bool:
    """True when the fused kernels cannot run this MLP but the layer-by-layer path can."""
    if isinstance(hidden_size, (tuple, list)):
        hidden = [int(h) for h in hidden_size]
    elif isinstance(hidden_size, int) and isinstance(num_layers, int):
        hidden = [hidden_size] * num_layers
    else:
        return False
    if not hidden or weight_norm or skip_connection or str(activation).lower() not in _ACTS:
        return False
    ragged = len(set(hidden)) != 1
    if max(hidden) > 256:
        return max(hidden) <= 2560
    if fourier and (int(fourier["dim"]) != hidden[0] or ragged or str(activation).lower() != "tanh"):
        return True
    return bool(ragged and random_weight)


class LayerwiseMLP(PirateNet):
    num_blocks = 0

    def __init__(self, input_keys, output_keys, num_layers, hidden_size, activation: str = "tanh", skip_connection: bool = False,
                 weight_norm: bool = False, input_dim: Optional[int] = None, output_dim: Optional[int] = None,
                 periods: Optional[Dict[str, Tuple[float, bool]]] = None, fourier: Optional[Dict[str, Union[float, int]]] = None,
                 random_weight: Optional[Dict[str, float]] = None):
        from . import activation as act_mod
        from .base import Arch

        Arch.__init__(self)
        self.input_keys, self.output_keys = tuple(input_keys), tuple(output_keys)
        if isinstance(hidden_size, (tuple, list)):
            if num_layers is not None:
                raise ValueError("num_layers should be None when hidden_size is specified")
            self.widths = [int(h) for h in hidden_size]
        else:
            self.widths = [int(hidden_size)] * int(num_layers)
        if weight_norm or skip_connection:
            raise NotImplementedError("layer-by-layer MLP: weight_norm / skip_connection are not available")
        if input_dim is not None and int(input_dim) != len(self.input_keys):
            raise NotImplementedError("multi-column inputs (input_dim != number of input keys)")
        if output_dim is not None and int(output_dim) != len(self.output_keys):
            raise NotImplementedError("multi-column outputs (output_dim != number of output keys)")
        self.activation = act_mod.get_activation(activation)
        if self.activation not in _ACTS:
            raise NotImplementedError(f"layer-by-layer MLP activation {activation!r}: the stream kernels carry {_ACTS}")
        self.hidden = max(self.widths)
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
                raise ValueError(f"out_features must be even, but got {fourier['dim']}.")
            self.half = int(fourier["dim"]) // 2
        self.c0 = 2 * self.half if self.half else self.d0
        m = len(self.output_keys)

        def lin(name, fin, fout):
            if self._rwf:
                return [(f"{name}.weight_v", (fin, fout)), (f"{name}.weight_g", (fout,)), (f"{name}.bias", (fout,))]
            return [(f"{name}.weight", (fin, fout)), (f"{name}.bias", (fout,))]

        shapes: List[Tuple[str, Tuple[int, ...]]] = [("fourier_emb.kernel", (self.d0, self.half))] if self.half else []
        fin = self.c0
        for i, w in enumerate(self.widths):
            shapes += lin(f"linears.{i}", fin, w)
            fin = w
        shapes += lin("last_fc", fin, m)
        self._shapes = shapes
        self.reparam = False
        self._bind_views(torch.zeros(sum(int(np.prod(s_)) for _, s_ in shapes), dtype=torch.float32, device=get_device()))
        self.layout = LayerwiseLayout(self)
        self._frozen = False
        self._init_parameters()
        self._predict_exec: Dict[int, "PlainExec"] = {}

    def linear_names(self) -> List[str]:
        return [f"linears.{i}" for i in range(len(self.widths))] + ["last_fc"]

    def _init_parameters(self):
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
            ex = self._predict_exec[n] = PlainExec(self, hp.StreamSpec([], 0), n, [torch.empty_like(t) for t in ins], train=False)
        for dst, src in zip(ex.inputs, ins):
            dst.copy_(src)
        U = torch.empty((len(self.output_keys), n), dtype=torch.float32, device=dev)
        ex.forward(self.flat_params, U, False)
        return {k: U[i].view(n, 1) for i, k in enumerate(self.output_keys)}


class LayerwiseLayout(PirateLayout):
    def __init__(self, model: LayerwiseMLP):
        self.model = model
        self.d_raw, self.d_out = len(model.input_keys), len(model.output_keys)
        self.n_hidden, self.width = len(model.widths), model.hidden
        self.embed, self.omega = model._embed, model._omega

    def make_exec(self, spec, n, inputs):
        return PlainExec(self.model, spec, n, inputs)


class PlainExec:
    """x0 -> (GEMM, bias + activation) per hidden layer -> GEMM: buffers and launch sequences of one (network, stream set,
    batch size); same contract as piratenet.PirateExec."""

    def __init__(self, model: LayerwiseMLP, spec: hp.StreamSpec, n: int, inputs, train: bool = True):
        if getattr(spec, "n3", 0) or getattr(spec, "n4", 0):
            raise NotImplementedError("layer-by-layer MLP: derivative order > 2")
        self.model, self.n, self.inputs = model, int(n), list(inputs)
        self.n1, self.n2 = len(spec.dirs), spec.n2
        self.S = 1 + self.n1 + self.n2
        self.NP = (self.n + 15) // 16 * 16
        self.m, self.c0, self.widths = len(model.output_keys), model.c0, list(model.widths)
        self.act = L.ACT[model.activation]
        dev = model.flat_params.device
        self.f32 = dict(dtype=torch.float32, device=dev)
        d = self.desc = L.PirateEmbedDesc()
        d.d_raw, d.d0, d.half, d.n1, d.n2 = len(model.input_keys), model.d0, model.half, self.n1, self.n2
        for j in range(d.d_raw):
            d.embed[j], d.omega[j] = model._embed[j], model._omega[j]
        for q, v in enumerate(spec.dirs):
            for j in range(d.d_raw):
                d.dirs[q][j] = float(v[j])
        d.N, d.NP = self.n, self.NP
        self._in_ptrs = (C.c_void_p * d.d_raw)(*[t.data_ptr() for t in self.inputs])
        blk = lambda c: torch.zeros((self.S, c, self.NP), **self.f32)  # noqa: E731
        self.X0 = blk(self.c0)
        self.Z = [blk(w) for w in self.widths]
        self.A = [blk(w) for w in self.widths]
        self.Y = blk(self.m)
        self.weff = {}
        if model._rwf:
            for name in model.linear_names():
                fin, fout = model._byname[name + ".weight_v"].shape
                self.weff[name] = torch.zeros(fin * fout, **self.f32)
        self._train = False
        if train:
            self._alloc_train()

    def _alloc_train(self):
        lib = L.lib()
        cmax = max(self.widths + [self.c0])
        flat = lambda: torch.zeros(self.S * cmax * self.NP, **self.f32)  # noqa: E731
        self.Ybar = torch.zeros((self.S, self.m, self.NP), **self.f32)
        self.OB, self.ZB = flat(), flat()
        self.achunks = int(lib.ppsci_pirate_act_chunks(self.NP))
        self.wchunks = int(lib.ppsci_pw_conv_wgrad_chunks(self.S, self.NP))
        self.echunks = int(lib.ppsci_pirate_embed_chunks(self.n))
        dims = [self.c0] + self.widths
        wmax = max(a * b for a, b in zip(dims, self.widths + [self.m]))
        self.pb = torch.zeros(self.achunks * max(self.widths), **self.f32)
        self.pw = torch.zeros(self.wchunks * wmax, **self.f32)
        self.pB = torch.zeros((self.echunks, max(1, self.model.d0 * self.model.half)), **self.f32)
        self.gw_eff = torch.zeros(wmax, **self.f32) if self.model._rwf else None
        self._train = True

    # ---- helpers
    def _t(self, params, name):
        off, n = self.model._offsets[name]
        return params[off:off + n]

    def _w(self, params, name):
        return self.weff[name] if self.model._rwf else self._t(params, name + ".weight")

    def _materialize(self, params):
        m = self.model
        if m._rwf:
            for name in m.linear_names():
                fin, fout = m._byname[name + ".weight_v"].shape
                hp.linear_materialize(L.LINEAR_RWF, fin, fout, self._t(params, name + ".weight_v"), self._t(params, name + ".weight_g"),
                                      None, self.weff[name], None)

    def _dense(self, x, W, fin, fout, out):
        L.check(L.lib().ppsci_pw_conv(self.S, fin, fout, self.NP, _p(x), _p(W), 1, None, None, 0, _p(out), None, _sp(out)))

    def _dense_t(self, gy, W, fin, fout, out):
        L.check(L.lib().ppsci_pw_conv(self.S, fout, fin, self.NP, _p(gy), _p(W), 0, None, None, 0, _p(out), None, _sp(out)))

    def _wgrad(self, x, zbar, fin, fout, name, params, grad):
        m = self.model
        cols = fin * fout
        part = self.pw[: self.wchunks * cols]
        L.check(L.lib().ppsci_pw_conv_wgrad(self.S, fout, fin, self.NP, _p(zbar), _p(x), _p(part), None, _sp(part)))
        if m._rwf:
            gw = self.gw_eff[:cols]
            hp.reduce_rows(part.view(self.wchunks, cols), self.wchunks, cols, gw, False)
            ov, nv = m._offsets[name + ".weight_v"]
            og, ng = m._offsets[name + ".weight_g"]
            hp.linear_pullback(L.LINEAR_RWF, fin, fout, params[ov:ov + nv], params[og:og + ng], gw, None, grad[ov:ov + nv],
                               grad[og:og + ng], None)
        else:
            ow, nw = m._offsets[name + ".weight"]
            hp.reduce_rows(part.view(self.wchunks, cols), self.wchunks, cols, grad[ow:ow + nw], False)

    # ---- forward / reverse
    def forward(self, params: torch.Tensor, Urows: torch.Tensor, train: bool) -> None:
        m, lib = self.model, L.lib()
        self._materialize(params)
        kern = self._t(params, "fourier_emb.kernel") if m.half else None
        L.check(lib.ppsci_pirate_embed_fwd(C.byref(self.desc), self._in_ptrs, _p(kern), _p(self.X0), _sp(self.X0)))
        y, fin = self.X0, self.c0
        for i, w in enumerate(self.widths):
            self._dense(y, self._w(params, f"linears.{i}"), fin, w, self.Z[i])
            L.check(lib.ppsci_pirate_act_fwd(L.PIRATE_ACT, self.act, w, self.n, self.NP, self.n1, self.n2, _p(self.Z[i]),
                                             _p(self._t(params, f"linears.{i}.bias")), None, None, None, None, _p(self.A[i]),
                                             _sp(self.A[i])))
            y, fin = self.A[i], w
        self._dense(y, self._w(params, "last_fc"), fin, self.m, self.Y)
        L.check(lib.ppsci_pirate_out_fwd(self.S, self.m, self.n, self.NP, _p(self.Y), _p(self._t(params, "last_fc.bias")),
                                         _p(Urows), _sp(Urows)))

    def backward(self, params: torch.Tensor, Ubar_rows: torch.Tensor, grad: torch.Tensor) -> None:
        if not self._train:
            self._alloc_train()
        m, lib = self.model, L.lib()
        grad = grad.view(-1)
        nl = len(self.widths)
        L.check(lib.ppsci_pirate_out_bwd(self.S, self.m, self.n, self.NP, _p(Ubar_rows), _p(self.Ybar), _sp(self.Ybar)))
        ob, _ = m._offsets["last_fc.bias"]
        for o in range(self.m):
            hp.reduce_rows(Ubar_rows[o * self.S].view(self.n, 1), self.n, 1, grad[ob + o:ob + o + 1], False)
        self._wgrad(self.A[-1], self.Ybar, self.widths[-1], self.m, "last_fc", params, grad)
        ob_buf = self.OB[: self.S * self.widths[-1] * self.NP]
        self._dense_t(self.Ybar, self._w(params, "last_fc"), self.widths[-1], self.m, ob_buf)
        for i in range(nl - 1, -1, -1):
            w = self.widths[i]
            yin, fin = (self.A[i - 1], self.widths[i - 1]) if i > 0 else (self.X0, self.c0)
            zb = self.ZB[: self.S * w * self.NP]
            pb = self.pb[: self.achunks * w]
            bname = f"linears.{i}.bias"
            L.check(lib.ppsci_pirate_act_bwd(L.PIRATE_ACT, self.act, w, self.n, self.NP, self.n1, self.n2, _p(self.Z[i]),
                                             _p(self._t(params, bname)), None, None, None, None, _p(ob_buf), _p(zb), None, None,
                                             None, _p(pb), None, _sp(zb)))
            o_, n_ = m._offsets[bname]
            hp.reduce_rows(pb.view(self.achunks, w), self.achunks, w, grad[o_:o_ + n_], False)
            self._wgrad(yin, zb, fin, w, f"linears.{i}", params, grad)
            if i > 0 or m.half:
                ob_buf = self.OB[: self.S * fin * self.NP]
                self._dense_t(zb, self._w(params, f"linears.{i}"), fin, w, ob_buf)
        if m.half:
            L.check(lib.ppsci_pirate_embed_bwd(C.byref(self.desc), self._in_ptrs, _p(self._t(params, "fourier_emb.kernel")),
                                               _p(ob_buf), _p(self.pB), _sp(self.pB)))
            ok, nk = m._offsets["fourier_emb.kernel"]
            hp.reduce_rows(self.pB, self.echunks, nk, grad[ok:ok + nk], False)
