"""ppsci.arch.MLP (/root/reference/ppsci/arch/mlp.py:139-315) on the fused HIP kernels.

Parameters live in ONE flat fp32 device buffer in `model.parameters()` order
(linears.0.weight [in,out], linears.0.bias, ..., last_fc.weight, last_fc.bias -- mlp.py:264-277);
`linears[i].weight` etc. are views into it, so the optimizer and the kernels see the same memory.

Calling the model
  * with traced inputs (graph.Sym, during expression compilation) returns traced network outputs;
  * with tensors / numpy arrays runs the forward kernel (no derivative streams) and returns
    [N, 1] tensors per output key -- the reference's eager `model(input_dict)`.
`weight_norm`, `random_weight` and `fourier` keep the reference's trainable tensors (weight_v / weight_g / bias,
fourier_emb.kernel) in `flat_params`; the kernels read `kernel_params`, which `materialize()` rebuilds from
them (csrc/reparam.hip), and `pull_back()` maps the kernel-layout gradient to the trainable tensors.
Options without a HIP kernel yet (stan / swish, per-layer widths, input transforms on the fused path) raise
NotImplementedError."""
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


class _FactoredLinear:
    """WeightNormLinear / RandomWeightFactorization (mlp.py:31-92): views of weight_v [in, out], weight_g [out], bias."""

    def __init__(self, weight_v: torch.Tensor, weight_g: torch.Tensor, bias: torch.Tensor):
        self.weight_v, self.weight_g, self.bias = weight_v, weight_g, bias

    def parameters(self):
        return [self.weight_v, self.weight_g, self.bias]


class _Kernel:
    """FourierEmbedding.kernel [in, dim / 2] (mlp.py:123-126)."""

    def __init__(self, kernel: torch.Tensor):
        self.kernel = kernel

    def parameters(self):
        return [self.kernel]


class PeriodEmbedding:
    """mlp.py:95-114: x_k -> [cos(w x_k), sin(w x_k)], w = 2*pi/period (non-trainable on the HIP path)."""

    def __init__(self, periods: Dict[str, Tuple[float, bool]]):
        for k, (p, trainable) in periods.items():
            if trainable:
                raise NotImplementedError("trainable period embedding has no fused HIP kernel yet")
        self.freqs_dict = {k: float(np.float32(2 * np.pi / float(p))) for k, (p, _) in periods.items()}


class _MLPMeta(type):
    """`ppsci.arch.MLP(...)` may hand back the layer-by-layer class (see MLP.__new__): such an object IS an MLP to user code
    (`isinstance(model, ppsci.arch.MLP)`), although it shares its implementation with PirateNet."""

    def __instancecheck__(cls, obj):
        if type.__instancecheck__(cls, obj):
            return True
        return cls.__name__ == "MLP" and type(obj).__name__ == "LayerwiseMLP"


class MLP(Arch, metaclass=_MLPMeta):
    def __new__(cls, input_keys=None, output_keys=None, num_layers=None, hidden_size=None, *args, **kwargs):
        """Configurations outside the fused kernels' envelope (width > 256, a Fourier embedding whose dim differs from
        hidden_size, per-layer widths with factored layers) are served by the layer-by-layer class (arch/layerwise_mlp.py):
        same constructor, same parameter names."""
        if cls is MLP:
            from .layerwise_mlp import LayerwiseMLP, wants_layerwise

            names = ("activation", "skip_connection", "weight_norm", "input_dim", "output_dim", "periods", "fourier", "random_weight")
            kw = dict(zip(names, args))
            kw.update(kwargs)
            try:
                layerwise = wants_layerwise(num_layers, hidden_size, **kw)
            except TypeError:
                layerwise = False
            if layerwise:
                return LayerwiseMLP(input_keys, output_keys, num_layers, hidden_size, **kw)
        return super().__new__(cls)

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
        if weight_norm:
            random_weight = None  # mlp.py:239-249: weight_norm is tested first
        # mlp.py:217-218, :264-272: input_dim / output_dim replace len(input_keys) / len(output_keys) as the first / last
        # layer's width, for keys that carry multi-column tensors.  Here every key is one [N] column of the SoA batch
        # (base.py:78-148 concat / split), so only the values that agree with the key counts are meaningful.
        if input_dim is not None and int(input_dim) != len(self.input_keys):
            raise NotImplementedError(f"input_dim={input_dim} with {len(self.input_keys)} input key(s): multi-column "
                                      "inputs are not supported on the HIP path (give one key per column)")
        if output_dim is not None and int(output_dim) != len(self.output_keys):
            raise NotImplementedError(f"output_dim={output_dim} with {len(self.output_keys)} output key(s): multi-column "
                                      "outputs are not supported on the HIP path (give one key per column)")
        # per-layer widths (mlp.py:199-201): the kernels run the padded width max(hidden); the trainable tensors keep the
        # reference's shapes and are embedded into zero-filled kernel-layout matrices before every sweep (reparam.hip)
        self._widths = list(hidden)
        padded = len(set(hidden)) != 1
        if padded:
            if weight_norm or random_weight or fourier:
                raise NotImplementedError("per-layer widths together with weight_norm / random_weight / fourier")
            if activation.lower() in L.PARAM_ACTS:
                raise NotImplementedError("per-layer widths together with a learnable activation")
            hidden = [max(hidden)] * len(hidden)
        self.activation = act_mod.get_activation(activation)
        self.skip_connection = bool(skip_connection)
        self.periods = periods
        self.fourier = fourier
        embed = [L.EMBED_NONE] * len(self.input_keys)
        omega = [0.0] * len(self.input_keys)
        if periods:
            self.period_emb = PeriodEmbedding(periods)
            for k, w in self.period_emb.freqs_dict.items():
                j = self.input_keys.index(k)
                embed[j], omega[j] = L.EMBED_PERIOD, w
        fourier_half = 0
        if fourier:
            # mlp.py:233-237 + FourierEmbedding :117-136.  The kernels run the embedding as a hidden layer of
            # width fourier["dim"] (matrix [B, B], cos | sin), so it has to equal the hidden width.
            if int(fourier["dim"]) % 2 != 0:
                raise ValueError(f"out_features must be even, but got {fourier['dim']}.")
            if int(fourier["dim"]) != hidden[0] or self.activation != "tanh":
                raise NotImplementedError("fourier embedding on the HIP path: dim must equal hidden_size, tanh only")
            fourier_half = int(fourier["dim"]) // 2
        self._linear_kind = (L.LINEAR_WEIGHT_NORM if weight_norm else L.LINEAR_RWF if random_weight
                             else L.LINEAR_PLAIN)
        self._rwf = dict(random_weight) if random_weight else None
        self._param_act = self.activation in L.PARAM_ACTS  # swish (scalar beta per layer) / stan (beta[H] per layer)
        if self._param_act and fourier_half:
            raise NotImplementedError("fourier embedding together with a learnable activation")
        self.reparam = bool(fourier_half) or self._linear_kind != L.LINEAR_PLAIN or self._param_act or padded
        self.layout = hp.NetLayout(len(self.input_keys), len(hidden) + (1 if fourier_half else 0), hidden[0],
                                   len(self.output_keys), self.activation, self.skip_connection, embed, omega,
                                   fourier_half)
        dev = get_device()
        # ---- trainable tensors in the reference's parameters() order (fourier_emb, linears, last_fc)
        shapes: List[Tuple[str, Tuple[int, ...]]] = []
        fin = self.layout.d0
        if fourier_half:
            shapes.append(("fourier_emb.kernel", (fin, fourier_half)))
            fin = 2 * fourier_half
        for l in range(len(hidden)):
            wl = self._widths[l]  # == hidden[0] unless the widths differ per layer
            if self._linear_kind == L.LINEAR_PLAIN:
                shapes += [(f"linears.{l}.weight", (fin, wl)), (f"linears.{l}.bias", (wl,))]
            else:  # WeightNormLinear / RandomWeightFactorization: weight_v, weight_g, bias (mlp.py:35-41, :69-75)
                shapes += [(f"linears.{l}.weight_v", (fin, wl)), (f"linears.{l}.weight_g", (wl,)),
                           (f"linears.{l}.bias", (wl,))]
            fin = wl
        if self._param_act:  # self.acts is registered between self.linears and self.last_fc (mlp.py:262-263, :274)
            shapes += [(f"acts.{l}.beta", () if self.activation == "swish" else (hidden[0],)) for l in range(len(hidden))]
        if self._linear_kind == L.LINEAR_RWF:  # mlp.py:266-272: with random_weight the last linear is factorised too
            shapes += [("last_fc.weight_v", (fin, len(self.output_keys))), ("last_fc.weight_g", (len(self.output_keys),)),
                       ("last_fc.bias", (len(self.output_keys),))]
        else:
            shapes += [("last_fc.weight", (fin, len(self.output_keys))), ("last_fc.bias", (len(self.output_keys),))]
        n_train = sum(int(np.prod(shp)) for _, shp in shapes)
        self._shapes = shapes
        self._n_hidden_linears, self._fourier_half = len(hidden), fourier_half
        self._bind_views(torch.zeros(n_train, dtype=torch.float32, device=dev))
        byname = dict(zip(self._names, self._views))
        self.acts = [self.activation] * len(hidden)
        # ---- what the kernels read: for plain nets the very same buffer, otherwise a second buffer in the
        # kernel layout (W0, b0, W1, b1, ...) that materialize() fills from the trainable tensors
        if not self.reparam:
            self.kernel_params = self.flat_params
            self._records = []
        else:
            assert n_train != 0
            self.kernel_params = torch.zeros(self.layout.n_params, dtype=torch.float32, device=dev)
            self._bind_grad(torch.zeros_like(self.flat_params))
            kviews, off = [], 0
            for _, shp in self.layout.param_shapes():
                n = int(np.prod(shp))
                kviews.append((off, n, shp))
                off += n
            # (kind, fin, fout, name of v, of g, of b, kernel slice of W (off, n, shape), of b): the tensors are
            # looked up by name at every call, so re-homing the buffers (ModelList) needs no bookkeeping here
            self._records = []
            kl = 0
            if fourier_half:
                w, b = kviews[0], kviews[1]
                self._records.append((L.LINEAR_FOURIER, w[2][0], w[2][1], "fourier_emb.kernel", None, None, w, b))
                kl = 1
            for l in range(len(hidden)):
                w, b = kviews[2 * (kl + l)], kviews[2 * (kl + l) + 1]
                if padded:  # (kind, source dims, ...): the trainable block is smaller than the kernel-layout slice
                    fs = self.layout.d0 if l == 0 else self._widths[l - 1]
                    self._records.append((L.LINEAR_PADDED, (fs, self._widths[l]), w[2], f"linears.{l}.weight", None,
                                          f"linears.{l}.bias", w, b))
                elif self._linear_kind == L.LINEAR_PLAIN:
                    self._records.append((L.LINEAR_PLAIN, w[2][0], w[2][1], f"linears.{l}.weight", None,
                                          f"linears.{l}.bias", w, b))
                else:
                    self._records.append((self._linear_kind, w[2][0], w[2][1], f"linears.{l}.weight_v",
                                          f"linears.{l}.weight_g", f"linears.{l}.bias", w, b))
            nlin = 2 * (kl + len(hidden))  # kernel layout: the last linear follows the hidden ones ...
            w, b = kviews[nlin], kviews[nlin + 1]
            if self._linear_kind == L.LINEAR_RWF:
                self._records.append((L.LINEAR_RWF, w[2][0], w[2][1], "last_fc.weight_v", "last_fc.weight_g",
                                      "last_fc.bias", w, b))
            elif padded:
                self._records.append((L.LINEAR_PADDED, (self._widths[-1], w[2][1]), w[2], "last_fc.weight", None, "last_fc.bias",
                                      w, b))
            else:
                self._records.append((L.LINEAR_PLAIN, w[2][0], w[2][1], "last_fc.weight", None, "last_fc.bias", w, b))
            if self._param_act:  # ... and then one [H] parameter vector per hidden layer
                none = (0, 0, ())
                for l in range(len(hidden)):
                    # Swish.beta has shape [] (activation.py:52-55): broadcast; Stan.beta [out_features] (:37-40)
                    kind = L.LINEAR_BROADCAST if self.activation == "swish" else L.LINEAR_PLAIN
                    self._records.append((kind, 1, hidden[0], f"acts.{l}.beta", None, None, kviews[nlin + 2 + l], none))
        self._frozen = False
        self._init_parameters()

    def _bind_views(self, flat: torch.Tensor) -> None:
        """(Re)build the per-tensor views of the trainable parameters over `flat` (ModelList re-homes its members'
        parameters into one buffer so that the optimizer and the all-reduce see a single tensor)."""
        self.flat_params = flat
        self._names, self._views = [], []
        off = 0
        for name, shp in self._shapes:
            n = int(np.prod(shp))
            self._names.append(name)
            self._views.append(flat[off:off + n].view(tuple(shp)))
            off += n
        byname = self._byname = dict(zip(self._names, self._views))
        nl = self._n_hidden_linears
        if self._linear_kind == L.LINEAR_PLAIN:
            self.linears = [_Linear(byname[f"linears.{l}.weight"], byname[f"linears.{l}.bias"]) for l in range(nl)]
        else:
            self.linears = [_FactoredLinear(byname[f"linears.{l}.weight_v"], byname[f"linears.{l}.weight_g"],
                                            byname[f"linears.{l}.bias"]) for l in range(nl)]
        if "last_fc.weight_v" in byname:
            self.last_fc = _FactoredLinear(byname["last_fc.weight_v"], byname["last_fc.weight_g"], byname["last_fc.bias"])
        else:
            self.last_fc = _Linear(byname["last_fc.weight"], byname["last_fc.bias"])
        self.fourier_emb = _Kernel(byname["fourier_emb.kernel"]) if self._fourier_half else None
        if not self.reparam:
            self.kernel_params = flat

    def _bind_grad(self, flat_grad: torch.Tensor) -> None:
        """Views of the trainable-layout gradient (reparametrised nets only)."""
        self._grad_train = flat_grad
        self._gviews, off = {}, 0
        for name, shp in self._shapes:
            n = int(np.prod(shp))
            self._gviews[name] = flat_grad[off:off + n].view(tuple(shp))
            off += n

    def rehome(self, flat: torch.Tensor, flat_grad: Optional[torch.Tensor] = None,
               kernel: Optional[torch.Tensor] = None) -> None:
        """Move the trainable parameters into `flat` (a slice of a larger buffer), keeping their values; for a
        reparametrised net also its trainable-layout gradient and its kernel-layout buffer."""
        assert flat.numel() == self.flat_params.numel()
        flat.copy_(self.flat_params)
        self._bind_views(flat)
        if self.reparam:
            assert flat_grad is not None and kernel is not None and kernel.numel() == self.layout.n_params
            self._bind_grad(flat_grad)
            self.kernel_params = kernel

    # ---- trainable tensors <-> kernel layout (csrc/reparam.hip); both are no-ops for a plain MLP
    def materialize(self) -> torch.Tensor:
        """Fill `kernel_params` from the trainable tensors; call before every forward sweep."""
        t = self._byname
        jobs = []
        for kind, fin, fout, v, g, b, (wo, wn, _), (bo, bn, _) in self._records:
            if kind == L.LINEAR_PADDED:
                hp.linear_pad(fin, fout, t[v].reshape(-1), t[b], self.kernel_params[wo:wo + wn], self.kernel_params[bo:bo + bn])
                continue
            jobs.append((kind, fin, fout, t[v].reshape(-1), t[g] if g else None, t[b] if b else None,
                         self.kernel_params[wo:wo + wn], self.kernel_params[bo:bo + bn] if bn else None))
        if jobs:
            hp.linear_multi(jobs, False, self.kernel_params)  # all re-parametrised layers in one launch (16 per launch)
        return self.kernel_params

    def pull_back(self, grad_kernel: torch.Tensor) -> torch.Tensor:
        """Gradient w.r.t. `kernel_params` -> gradient w.r.t. the trainable tensors (`flat_params` order)."""
        if not self.reparam:
            return grad_kernel
        t, gt = self._byname, self._gviews
        jobs = []
        for kind, fin, fout, v, g, b, (wo, wn, _), (bo, bn, _) in self._records:
            if kind == L.LINEAR_PADDED:
                hp.linear_unpad(fin, fout, grad_kernel[wo:wo + wn], grad_kernel[bo:bo + bn], gt[v].reshape(-1), gt[b])
                continue
            jobs.append((kind, fin, fout, t[v].reshape(-1), t[g] if g else None, grad_kernel[wo:wo + wn],
                         grad_kernel[bo:bo + bn] if bn else None, gt[v].reshape(-1), gt[g] if g else None,
                         gt[b] if b else None))
        if jobs:
            hp.linear_multi(jobs, True, grad_kernel)
        return self._grad_train

    # ---- parameters
    def _init_parameters(self):
        """paddle nn.Linear default: Xavier-uniform weight, zero bias (Paddle behaviour);
        drawn from numpy's global RNG so that ppsci.utils.misc.set_random_seed controls it."""
        if self.fourier_emb is not None:  # nn.initializer.Normal(std=scale), mlp.py:123-126
            k = self.fourier_emb.kernel
            k.copy_(torch.from_numpy(np.random.normal(0.0, float(self.fourier["scale"]), size=tuple(k.shape))
                                     .astype(np.float32)))
        if self._param_act:  # Swish(beta=1.0) / Stan: Constant(1) (activation.py:37-40, :50-55)
            for n_, v_ in zip(self._names, self._views):
                if n_.startswith("acts."):
                    v_.fill_(1.0)
        for i, lin in enumerate(self.linears + [self.last_fc]):
            w = lin.weight_v if isinstance(lin, _FactoredLinear) else lin.weight
            fin, fout = w.shape
            lim = math.sqrt(6.0 / (fin + fout))
            if self.activation == "siren" and i < len(self.linears):
                # mlp.py:256-260 / activation.py:106-136: first layer U(-1/in, 1/in), hidden U(+-sqrt(6/in)/w0)
                lim = 1.0 / fin if i == 0 else math.sqrt(6.0 / fin) / L.SIREN_W0
            if isinstance(lin, _FactoredLinear) and self._linear_kind == L.LINEAR_RWF:
                # mlp.py:78-85: v ~ glorot normal, g ~ N(mean, std); g <- exp(g); v <- v / g
                vv = np.random.normal(0.0, math.sqrt(2.0 / (fin + fout)), size=(fin, fout)).astype(np.float32)
                gg = np.exp(np.random.normal(self._rwf["mean"], self._rwf["std"], size=(fout,)).astype(np.float32))
                lin.weight_v.copy_(torch.from_numpy(vv / gg))
                lin.weight_g.copy_(torch.from_numpy(gg))
            else:
                w.copy_(torch.from_numpy(np.random.uniform(-lim, lim, size=(fin, fout)).astype(np.float32)))
                if isinstance(lin, _FactoredLinear):
                    lin.weight_g.fill_(1.0)  # mlp.py:45-46
            lin.bias.zero_()

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
        hp.taylor_fwd(desc, self.materialize(), ins, U, None)
        return {k: U[i].view(n, 1) for i, k in enumerate(self.output_keys)}

    def forward(self, x: Dict[str, object]) -> Dict[str, object]:  # mlp.py:298-315
        traced = any(isinstance(v, Sym) for v in x.values())
        if self._input_transform is not None:
            # mlp.py:299-300: the network sees transform(x); x itself stays in the data dict for the expressions
            xt = self._input_transform(dict(x))
            if traced:
                from ..graph import _lift

                self._traced_features = {k: _lift(xt[k]) for k in self.input_keys}
                y = {k: Sym.net(self, i) for i, k in enumerate(self.output_keys)}
            else:
                y = self._forward_numeric(xt)
            if self._output_transform is not None:
                y = self._output_transform(x, y)
            return y
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
