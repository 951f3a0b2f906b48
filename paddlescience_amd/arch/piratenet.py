"""ppsci.arch.PirateNet (/root/reference/ppsci/arch/mlp.py:530-820) on HIP kernels, layer by layer.

PirateNet's gates multiply the streams of three tensors (f * U + (1 - f) * V), which the register-resident single-kernel
MLP sweep (taylor_fwd / taylor_bwd) cannot hold; here every dense layer is one MFMA GEMM over all Taylor streams at once
(`ppsci_pw_conv`, [S, C, NP] = its [B, C, P]) and the stages in between -- period / Fourier embedding, bias + activation
fused with the gate or with the residual connection, and their hand-written reverse -- are the kernels of csrc/pirate.hip.
The residual expressions, losses and the optimizer are the same fused path as for ppsci.arch.MLP: `PirateExec` is what
`engine.FusedConstraint` runs in place of `taylor_fwd` / `taylor_bwd` for this architecture.

Trainable tensors, in the reference's `parameters()` order and with its names (one flat fp32 device buffer):
    fourier_emb.kernel                                       [d0, dim/2]
    embed_u.0.{weight,bias} | {weight_v,weight_g,bias}       nn.Linear | RandomWeightFactorization
    embed_v.0. ...
    blocks.i.alpha [1], blocks.i.linear{1,2,3}. ...
    last_fc. ...
Not available (raise): weight_norm, learnable activations (stan / swish), derivative order > 2, a `fourier` embedding
whose dim differs from hidden_size (the reference's own block arithmetic needs them equal), input transforms."""
from __future__ import annotations

import ctypes as C
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

_p = hp._p
_ACTS = ("tanh", "silu", "sigmoid", "sin", "cos", "gelu")


def _sp(t: torch.Tensor):
    return hp._stream_ptr(t)


class PirateLayout:
    """What compile / engine need to know about the network (the role hotpath.NetLayout plays for MLP)."""

    is_pirate = True

    def __init__(self, model: "PirateNet"):
        self.model = model
        self.d_raw, self.d_out = len(model.input_keys), len(model.output_keys)
        self.n_hidden, self.width = 3 * model.num_blocks + 1, model.hidden
        self.embed, self.omega = model._embed, model._omega

    @property
    def n_params(self) -> int:
        return int(self.model.flat_params.numel())

    def desc(self, streams):
        return None

    def make_exec(self, spec: hp.StreamSpec, n: int, inputs) -> "PirateExec":
        return PirateExec(self.model, spec, n, inputs)


class PirateNet(Arch):
    def __init__(
        self,
        input_keys: Tuple[str, ...],
        output_keys: Tuple[str, ...],
        num_blocks: int,
        hidden_size: int,
        activation: str = "tanh",
        weight_norm: bool = False,
        input_dim: Optional[int] = None,
        output_dim: Optional[int] = None,
        periods: Optional[Dict[str, Tuple[float, bool]]] = None,
        fourier: Optional[Dict[str, Union[float, int]]] = None,
        random_weight: Optional[Dict[str, float]] = None,
    ):
        super().__init__()
        self.input_keys, self.output_keys = tuple(input_keys), tuple(output_keys)
        if not isinstance(hidden_size, int):
            raise ValueError(f"hidden_size should be int, but got {type(hidden_size)}")  # mlp.py:693-694
        if not isinstance(num_blocks, int):
            raise ValueError("num_blocks should be an int")  # mlp.py:690-691
        if weight_norm:
            raise NotImplementedError("PirateNet(weight_norm=True) has no HIP kernel path")
        if input_dim is not None and int(input_dim) != len(self.input_keys):
            raise NotImplementedError("multi-column inputs (input_dim != number of input keys)")
        if output_dim is not None and int(output_dim) != len(self.output_keys):
            raise NotImplementedError("multi-column outputs (output_dim != number of output keys)")
        self.activation = act_mod.get_activation(activation)
        if self.activation not in _ACTS:
            raise NotImplementedError(f"PirateNet activation {activation!r}: the stream kernels carry {_ACTS}")
        if len(self.input_keys) > L.MAX_IN or len(self.output_keys) > L.MAX_OUT:
            raise NotImplementedError(f"at most {L.MAX_IN} inputs / {L.MAX_OUT} outputs")
        self.num_blocks, self.hidden = int(num_blocks), int(hidden_size)
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
        if not fourier:
            raise NotImplementedError("PirateNet without a `fourier` embedding: the blocks add tensors of width "
                                      "hidden_size to the embedded input (mlp.py:614-621), so the reference itself only "
                                      "runs when the two agree; give fourier={'dim': hidden_size, 'scale': ...}")
        if int(fourier["dim"]) % 2 != 0:
            raise ValueError(f"out_features must be even, but got {fourier['dim']}.")  # mlp.py:120-121
        if int(fourier["dim"]) != self.hidden:
            raise NotImplementedError("fourier['dim'] must equal hidden_size (PirateNetBlock adds them, mlp.py:614-621)")
        self.half = self.hidden // 2
        H, m = self.hidden, len(self.output_keys)

        def lin(name, fin, fout):
            if self._rwf:
                return [(f"{name}.weight_v", (fin, fout)), (f"{name}.weight_g", (fout,)), (f"{name}.bias", (fout,))]
            return [(f"{name}.weight", (fin, fout)), (f"{name}.bias", (fout,))]

        shapes: List[Tuple[str, Tuple[int, ...]]] = [("fourier_emb.kernel", (self.d0, self.half))]
        shapes += lin("embed_u.0", H, H) + lin("embed_v.0", H, H)
        for i in range(self.num_blocks):
            shapes.append((f"blocks.{i}.alpha", (1,)))
            for j in (1, 2, 3):
                shapes += lin(f"blocks.{i}.linear{j}", H, H)
        shapes += lin("last_fc", H, m)
        self._shapes = shapes
        self.reparam = False  # factorised layers are materialised inside PirateExec; gradients come out trainable-layout
        self._bind_views(torch.zeros(sum(int(np.prod(s)) for _, s in shapes), dtype=torch.float32, device=get_device()))
        self.layout = PirateLayout(self)
        self._frozen = False
        self._init_parameters()
        self._predict_exec: Dict[int, "PirateExec"] = {}

    # ---- parameters
    def _bind_views(self, flat: torch.Tensor) -> None:
        self.flat_params = self.kernel_params = flat
        self._names, self._views, self._offsets = [], [], {}
        off = 0
        for name, shp in self._shapes:
            n = int(np.prod(shp))
            self._names.append(name)
            self._views.append(flat[off:off + n].view(tuple(shp)))
            self._offsets[name] = (off, n)
            off += n
        self._byname = dict(zip(self._names, self._views))

    def rehome(self, flat: torch.Tensor, flat_grad=None, kernel=None) -> None:
        """ModelList: move the trainable parameters into `flat` (a slice of the list's buffer), keeping their values."""
        assert flat.numel() == self.flat_params.numel()
        flat.copy_(self.flat_params)
        self._bind_views(flat)

    def linear_names(self) -> List[str]:
        out = ["embed_u.0", "embed_v.0"]
        for i in range(self.num_blocks):
            out += [f"blocks.{i}.linear{j}" for j in (1, 2, 3)]
        return out + ["last_fc"]

    def _init_parameters(self):
        """Same draws as the reference's constructors, from numpy's global RNG (ppsci.utils.misc.set_random_seed):
        FourierEmbedding Normal(std=scale) (mlp.py:123-126); nn.Linear Xavier-uniform / zero bias; RandomWeightFactorization
        glorot normal v, g = exp(N(mean, std)), v <- v / g (mlp.py:78-85); alpha = 0 (mlp.py:590-595)."""
        t = self._byname
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
        for i in range(self.num_blocks):
            t[f"blocks.{i}.alpha"].zero_()

    def parameters(self) -> List[torch.Tensor]:
        return list(self._views)

    def named_parameters(self):
        return list(zip(self._names, self._views))

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return dict(zip(self._names, self._views))

    def set_state_dict(self, state):
        missing = [n for n in self._names if n not in state]
        unexpected = [n for n in state if n not in self._names]
        for n, v in zip(self._names, self._views):
            if n in state:
                src = state[n]
                src = torch.as_tensor(np.asarray(src.detach().cpu() if isinstance(src, torch.Tensor) else src), dtype=torch.float32)
                if tuple(src.shape) != tuple(v.shape):
                    raise ValueError(f"shape mismatch for {n}: {tuple(src.shape)} vs {tuple(v.shape)}")
                v.copy_(src)
        return missing, unexpected

    def materialize(self) -> torch.Tensor:
        return self.flat_params

    def pull_back(self, grad: torch.Tensor) -> torch.Tensor:
        return grad

    # ---- forward
    def _forward_numeric(self, x: Dict[str, object]) -> Dict[str, torch.Tensor]:
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
            ex = self._predict_exec[n] = PirateExec(self, hp.StreamSpec([], 0), n, [torch.empty_like(t) for t in ins], train=False)
        for dst, src in zip(ex.inputs, ins):
            dst.copy_(src)
        U = torch.empty((len(self.output_keys), n), dtype=torch.float32, device=dev)
        ex.forward(self.flat_params, U, False)
        return {k: U[i].view(n, 1) for i, k in enumerate(self.output_keys)}

    def forward(self, x: Dict[str, object]) -> Dict[str, object]:  # mlp.py:802-820
        if self._input_transform is not None:
            raise NotImplementedError("PirateNet with a registered input transform")
        traced = any(isinstance(v, Sym) for v in x.values())
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


class PirateExec:
    """Buffers and launch sequence of one (network, stream set, batch size): forward(params, U, train) fills the U row
    block [d_out * S, N]; backward(params, Ubar, grad_row) writes d loss / d (trainable parameters) into grad_row."""

    def __init__(self, model: PirateNet, spec: hp.StreamSpec, n: int, inputs, train: bool = True):
        if getattr(spec, "n3", 0) or getattr(spec, "n4", 0):
            raise NotImplementedError("PirateNet: derivative order > 2")
        self.model, self.spec, self.n = model, spec, int(n)
        self.inputs = list(inputs)
        self.n1, self.n2 = len(spec.dirs), spec.n2
        self.S = 1 + self.n1 + self.n2
        self.NP = (self.n + 15) // 16 * 16
        self.H, self.m, self.nb = model.hidden, len(model.output_keys), model.num_blocks
        self.act = L.ACT[model.activation]
        dev = model.flat_params.device
        f32 = dict(dtype=torch.float32, device=dev)
        d = self.desc = L.PirateEmbedDesc()
        d.d_raw, d.d0, d.half, d.n1, d.n2 = len(model.input_keys), model.d0, model.half, self.n1, self.n2
        for j in range(d.d_raw):
            d.embed[j], d.omega[j] = model._embed[j], model._omega[j]
        for q, v in enumerate(spec.dirs):
            for j in range(d.d_raw):
                d.dirs[q][j] = float(v[j])
        d.N, d.NP = self.n, self.NP
        self._in_ptrs = (C.c_void_p * d.d_raw)(*[t.data_ptr() for t in self.inputs])
        blk = lambda c=self.H: torch.zeros((self.S, c, self.NP), **f32)  # noqa: E731
        self.c0 = int(getattr(model, "c0", self.H))  # channels of x0 (PirateNet: = hidden; ModifiedMLP: d0 or fourier dim)
        self.X0, self.ZU, self.ZV, self.U, self.V = blk(self.c0), blk(), blk(), blk(), blk()
        self.blocks = [dict(Z1=blk(), O1=blk(), Z2=blk(), O2=blk(), Z3=blk(), X=blk()) for _ in range(self.nb)]
        self.layers = [dict(Z=blk(), O=blk()) for _ in range(int(getattr(model, "num_gated_layers", 0)))]  # ModifiedMLP
        self.Y = blk(self.m)
        self._train_ready = False
        # effective weights of the factorised layers (v * g): [fin, fout] row-major, what the GEMMs read
        self.weff = {}
        if model._rwf:
            for name in model.linear_names():
                fin, fout = model._byname[name + ".weight_v"].shape
                self.weff[name] = torch.zeros(fin * fout, **f32)
        if train:
            self._alloc_train()

    # ---- helpers
    def _w(self, params: torch.Tensor, name: str):
        m = self.model
        if m._rwf:
            return self.weff[name]
        off, n = m._offsets[name + ".weight"]
        return params[off:off + n]

    def _t(self, params: torch.Tensor, name: str):
        off, n = self.model._offsets[name]
        return params[off:off + n]

    def _materialize(self, params: torch.Tensor):
        m = self.model
        if not m._rwf:
            return
        jobs = []
        for name in m.linear_names():
            fin, fout = m._byname[name + ".weight_v"].shape
            jobs.append((L.LINEAR_RWF, fin, fout, self._t(params, name + ".weight_v"), self._t(params, name + ".weight_g"),
                         None, self.weff[name], None))
        hp.linear_multi(jobs, False, params)  # every factored layer in one launch (16 per launch)

    def _dense(self, x, W, fin, fout, out, accumulate=False):
        # out[s, o, p] = sum_i W[i, o] x[s, i, p]: nn.Linear weight [in, out] used as the transposed conv weight
        L.check(L.lib().ppsci_pw_conv(self.S, fin, fout, self.NP, _p(x), _p(W), 1, None, None, 1 if accumulate else 0,
                                      _p(out), None, _sp(out)))

    def _dense_t(self, gy, W, fin, fout, out, accumulate=False):
        # data gradient: out[s, i, p] (+)= sum_o W[i, o] gy[s, o, p]  (W [fin, fout] read as a conv weight [Co = fin, Ci = fout])
        L.check(L.lib().ppsci_pw_conv(self.S, fout, fin, self.NP, _p(gy), _p(W), 0, None, None, 1 if accumulate else 0,
                                      _p(out), None, _sp(out)))

    def _act_fwd(self, mode, z, bias, out, U=None, V=None, x=None, alpha=None):
        L.check(L.lib().ppsci_pirate_act_fwd(mode, self.act, self.H, self.n, self.NP, self.n1, self.n2, _p(z), _p(bias), _p(U),
                                             _p(V), _p(x), _p(alpha), _p(out), _sp(out)))

    # ---- forward
    def forward(self, params: torch.Tensor, Urows: torch.Tensor, train: bool) -> None:
        m, H = self.model, self.H
        lib = L.lib()
        self._materialize(params)
        L.check(lib.ppsci_pirate_embed_fwd(C.byref(self.desc), self._in_ptrs, _p(self._t(params, "fourier_emb.kernel")),
                                           _p(self.X0), _sp(self.X0)))
        self._dense(self.X0, self._w(params, "embed_u.0"), H, H, self.ZU)
        self._act_fwd(L.PIRATE_ACT, self.ZU, self._t(params, "embed_u.0.bias"), self.U)
        self._dense(self.X0, self._w(params, "embed_v.0"), H, H, self.ZV)
        self._act_fwd(L.PIRATE_ACT, self.ZV, self._t(params, "embed_v.0.bias"), self.V)
        x = self.X0
        for i, b in enumerate(self.blocks):
            pre = f"blocks.{i}."
            self._dense(x, self._w(params, pre + "linear1"), H, H, b["Z1"])
            self._act_fwd(L.PIRATE_GATE, b["Z1"], self._t(params, pre + "linear1.bias"), b["O1"], U=self.U, V=self.V)
            self._dense(b["O1"], self._w(params, pre + "linear2"), H, H, b["Z2"])
            self._act_fwd(L.PIRATE_GATE, b["Z2"], self._t(params, pre + "linear2.bias"), b["O2"], U=self.U, V=self.V)
            self._dense(b["O2"], self._w(params, pre + "linear3"), H, H, b["Z3"])
            self._act_fwd(L.PIRATE_RES, b["Z3"], self._t(params, pre + "linear3.bias"), b["X"], x=x,
                          alpha=self._t(params, pre + "alpha"))
            x = b["X"]
        self._dense(x, self._w(params, "last_fc"), H, self.m, self.Y)
        L.check(lib.ppsci_pirate_out_fwd(self.S, self.m, self.n, self.NP, _p(self.Y), _p(self._t(params, "last_fc.bias")),
                                         _p(Urows), _sp(Urows)))

    # ---- reverse
    def _alloc_train(self):
        dev = self.model.flat_params.device
        f32 = dict(dtype=torch.float32, device=dev)
        blk = lambda c=self.H: torch.zeros((self.S, c, self.NP), **f32)  # noqa: E731
        self.Ybar = blk(self.m)
        self.XB = [blk(), blk()]  # adjoint of the running block input / output (ping-pong)
        self.OB, self.ZB, self.UB, self.VB = blk(), blk(), blk(), blk()
        lib = L.lib()
        self.achunks = int(lib.ppsci_pirate_act_chunks(self.NP))
        self.wchunks = int(lib.ppsci_pw_conv_wgrad_chunks(self.S, self.NP))
        self.echunks = int(lib.ppsci_pirate_embed_chunks(self.n))
        # Per-chunk partial sums: every producer of a backward pass keeps its OWN buffer until the end of the pass, where
        # ppsci_reduce_rows_multi sums them all in two launches (27 reductions of ~6 us each were launch latency, not work)
        self._pbufs: List[torch.Tensor] = []
        self._pcall, self._psegs, self._pullbacks = 0, [], []
        self.pB = torch.zeros((self.echunks, max(1, self.model.d0 * self.model.half)), **f32)
        self.XB0 = blk(self.c0) if self.c0 != self.H else None  # adjoint of x0 when its width differs from the hidden one
        self._train_ready = True

    def _pbuf(self, n: int) -> torch.Tensor:
        i = self._pcall
        self._pcall += 1
        if i == len(self._pbufs):
            self._pbufs.append(torch.zeros(n, dtype=torch.float32, device=self.ZB.device))
        assert self._pbufs[i].numel() >= n
        return self._pbufs[i]

    def _sum_later(self, part: torch.Tensor, rows: int, cols: int, dst: torch.Tensor) -> None:
        self._psegs.append((part.data_ptr(), dst.data_ptr(), rows, cols))

    def _flush_sums(self) -> None:
        st = _sp(self.ZB)
        for i0 in range(0, len(self._psegs), 16):
            batch = self._psegs[i0:i0 + 16]
            arr = (L.ReduceSeg * len(batch))()
            for k, (src, dst, rows, cols) in enumerate(batch):
                arr[k].partials, arr[k].out, arr[k].rows, arr[k].cols, arr[k].accumulate = src, dst, rows, cols, 0
            L.check(L.lib().ppsci_reduce_rows_multi(len(batch), arr, st))
        self._psegs = []
        if self._pullbacks:  # the trainable tensors behind the summed kernel-layout gradients, 16 layers per launch
            hp.linear_multi(self._pullbacks, True, self.ZB)
        self._pullbacks = []

    def _wgrad(self, x, zbar, fin, fout, name, params, grad):
        """d loss / d W[i, o] = sum_{s,p} x[s,i,p] zbar[s,o,p] -> the layer's trainable tensors in `grad` (summed at the
        end of the pass: _flush_sums)."""
        m = self.model
        cols = fin * fout
        pw = self._pbuf(self.wchunks * cols)
        # (conv roles swapped: "x" = zbar with Ci = fout, "gy" = x with Co = fin, so the partial blocks are [fin, fout])
        L.check(L.lib().ppsci_pw_conv_wgrad(self.S, fout, fin, self.NP, _p(zbar), _p(x), _p(pw), None, _sp(pw)))
        if m._rwf:
            gw = self._pbuf(cols)
            self._sum_later(pw, self.wchunks, cols, gw)
            ov, nv = m._offsets[name + ".weight_v"]
            og, ng = m._offsets[name + ".weight_g"]
            self._pullbacks.append((L.LINEAR_RWF, fin, fout, params[ov:ov + nv], params[og:og + ng], gw, None,
                                    grad[ov:ov + nv], grad[og:og + ng], None))
        else:
            ow, nw = m._offsets[name + ".weight"]
            self._sum_later(pw, self.wchunks, cols, grad[ow:ow + nw])

    def _act_bwd(self, mode, z, bias_name, obar, params, grad, U=None, V=None, x=None, alpha_name=None, xbar=None):
        m = self.model
        alpha = self._t(params, alpha_name) if alpha_name else None
        pb = self._pbuf(self.achunks * self.H)
        palpha = self._pbuf(self.H * self.achunks) if mode == L.PIRATE_RES else None
        L.check(L.lib().ppsci_pirate_act_bwd(mode, self.act, self.H, self.n, self.NP, self.n1, self.n2, _p(z),
                                             _p(self._t(params, bias_name)), _p(U), _p(V), _p(x), _p(alpha), _p(obar), _p(self.ZB),
                                             _p(self.UB) if mode == L.PIRATE_GATE else None,
                                             _p(self.VB) if mode == L.PIRATE_GATE else None, _p(xbar), _p(pb),
                                             _p(palpha), _sp(self.ZB)))
        ob, nb_ = m._offsets[bias_name]
        self._sum_later(pb, self.achunks, self.H, grad[ob:ob + nb_])
        if alpha_name:
            oa, na = m._offsets[alpha_name]
            self._sum_later(palpha, self.H * self.achunks, 1, grad[oa:oa + na])

    def backward(self, params: torch.Tensor, Ubar_rows: torch.Tensor, grad: torch.Tensor) -> None:
        """`grad`: flat [n_params] slice of the gradient buffer, fully overwritten."""
        if not self._train_ready:
            self._alloc_train()
        m, H, lib = self.model, self.H, L.lib()
        grad = grad.view(-1)
        self._pcall, self._psegs, self._pullbacks = 0, [], []
        # last_fc
        L.check(lib.ppsci_pirate_out_bwd(self.S, self.m, self.n, self.NP, _p(Ubar_rows), _p(self.Ybar), _sp(self.Ybar)))
        ob, nb_ = m._offsets["last_fc.bias"]
        for o in range(self.m):  # bias gradient = sum over points of the value-stream adjoint
            self._sum_later(Ubar_rows[o * self.S], self.n, 1, grad[ob + o:ob + o + 1])
        xlast = self.blocks[-1]["X"] if self.blocks else self.X0
        self._wgrad(xlast, self.Ybar, H, self.m, "last_fc", params, grad)
        cur = 0
        self._dense_t(self.Ybar, self._w(params, "last_fc"), H, self.m, self.XB[cur])
        self.UB.zero_()
        self.VB.zero_()
        for i in range(self.nb - 1, -1, -1):
            b, pre = self.blocks[i], f"blocks.{i}."
            xin = self.blocks[i - 1]["X"] if i > 0 else self.X0
            nxt = 1 - cur
            # x' = alpha h + (1 - alpha) x;  h = act(W3 o2 + b3)
            self._act_bwd(L.PIRATE_RES, b["Z3"], pre + "linear3.bias", self.XB[cur], params, grad, x=xin,
                          alpha_name=pre + "alpha", xbar=self.XB[nxt])
            self._wgrad(b["O2"], self.ZB, H, H, pre + "linear3", params, grad)
            self._dense_t(self.ZB, self._w(params, pre + "linear3"), H, H, self.OB)
            # o2 = gate(act(W2 o1 + b2))
            self._act_bwd(L.PIRATE_GATE, b["Z2"], pre + "linear2.bias", self.OB, params, grad, U=self.U, V=self.V)
            self._wgrad(b["O1"], self.ZB, H, H, pre + "linear2", params, grad)
            self._dense_t(self.ZB, self._w(params, pre + "linear2"), H, H, self.OB)
            # o1 = gate(act(W1 x + b1))
            self._act_bwd(L.PIRATE_GATE, b["Z1"], pre + "linear1.bias", self.OB, params, grad, U=self.U, V=self.V)
            self._wgrad(xin, self.ZB, H, H, pre + "linear1", params, grad)
            self._dense_t(self.ZB, self._w(params, pre + "linear1"), H, H, self.XB[nxt], accumulate=True)
            cur = nxt
        # embeddings U = act(W_u x0 + b_u), V = act(W_v x0 + b_v): their adjoints were accumulated by the gates
        for name, Z, B_ in (("embed_u.0", self.ZU, self.UB), ("embed_v.0", self.ZV, self.VB)):
            self._act_bwd(L.PIRATE_ACT, Z, name + ".bias", B_, params, grad)
            self._wgrad(self.X0, self.ZB, H, H, name, params, grad)
            self._dense_t(self.ZB, self._w(params, name), H, H, self.XB[cur], accumulate=True)
        # Fourier kernel
        L.check(lib.ppsci_pirate_embed_bwd(C.byref(self.desc), self._in_ptrs, _p(self._t(params, "fourier_emb.kernel")),
                                           _p(self.XB[cur]), _p(self.pB), _sp(self.pB)))
        ok, nk = m._offsets["fourier_emb.kernel"]
        self._sum_later(self.pB, self.echunks, nk, grad[ok:ok + nk])
        self._flush_sums()
