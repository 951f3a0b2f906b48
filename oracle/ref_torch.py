"""TEST INFRASTRUCTURE (oracle) -- never imported by the product path.

CPU restatement of the reference hot path on torch-CPU, following the reference's
*algorithm* (reverse-over-reverse autodiff on a dynamic graph): every derivative is one
`torch.autograd.grad(y, xs, grad_outputs=ones, create_graph=True)` call, which is what
`paddle.grad(y, xs, create_graph=True)` does with its implicit all-ones cotangent
(ppsci/autodiff/ad.py:73-75).  Run it in float64 ("truth") or float32 ("reference-like").

What it follows in /root/reference (file:line):
  * ppsci/arch/base.py:78-148        concat_to_tensor / split_to_dict
  * ppsci/arch/mlp.py:95-114         PeriodEmbedding
  * ppsci/arch/mlp.py:281-315        MLP.forward_tensor / MLP.forward (incl. the skip quirk)
  * ppsci/arch/mlp.py:530-820        PirateNetBlock / PirateNet
  * ppsci/arch/mlp.py:318-528        ModifiedMLP
  * ppsci/arch/activation.py:77-88   Silu = x * sigmoid(x)
  * ppsci/autodiff/ad.py:30-341      _Jacobian / Jacobians / _Hessian / Hessians / clear
  * ppsci/utils/symbolic.py:111-137  _cvt_to_key
  * ppsci/utils/symbolic.py:184-267  OperatorNode (Add / Mul are left folds in sympy arg order)
  * ppsci/utils/symbolic.py:310-333  DerivativeNode (odd order -> jacobian, pairs -> hessian)
  * ppsci/utils/symbolic.py:507-534  _post_traverse, :791-806 subs(1.0, 1) + dedupe
  * ppsci/equation/pde/laplace.py:40-55, allen_cahn.py:56-64, navier_stokes.py:70-151, poisson.py:40-53
  * ppsci/loss/mse.py:82-105         MSELoss.forward
  * ppsci/loss/mtl/sum.py:45-60      Sum aggregator
  * ppsci/utils/expression.py:60-131 ExpressionSolver.train_forward
  * ppsci/optimizer/optimizer.py:225-248 Adam (paddle.optimizer.Adam, beta1=.9 beta2=.999 eps=1e-8)

Pinning status.  PaddlePaddle cannot be installed here, so the real reference cannot be run end to end.
Pinned are: the stored known answers (MSELoss mse.py:46-68, NS strings equation/pde/base.py:99-111,
tests/test_oracle.py), and -- for the NN / autodiff numerics -- tests/golden/hotpath.npz, which was produced
by executing the reference's OWN Python files (arch/mlp.py, autodiff/ad.py, utils/symbolic.py with
fuse_derivative=True, equation/pde/*.py, loss/mse.py) with `paddle` replaced by a torch-backed shim
(tests/golden/_paddle_shim.py, tests/golden/make_hotpath_golden.py); this restatement reproduces those
fixtures to 1e-10 (tests/test_golden_hotpath.py).  The SPINN / Helmholtz and FNO restatements at the end of
this file are pinned the same way (tests/golden/make_spinn_golden.py -> spinn.npz, make_fno_golden.py ->
fno.npz: the reference's arch/spinn.py, ModifiedMLP, equation/pde/helmholtz.py, arch/fno_block.py, tfnonet.py
executed under the shim; tests/test_golden_spinn.py, tests/test_golden_fno.py), and so are the MLP variants
(weight_norm / random_weight / fourier / swish / stan), CausalMSELoss, ModelList and ParameterNode
(tests/golden/make_variants_golden.py -> variants.npz, tests/test_golden_variants.py).  What remains unpinned is
PaddlePaddle's own kernel arithmetic (covered by the fp32-vs-fp64 tolerance).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import sympy as sp
import torch

from oracle.taylor_np import NetSpec

DETACH_FUNC_NAME = "detach"  # ppsci/equation/pde/base.py:27


# ----------------------------------------------------------------------------- arch
class MLP:
    """Restatement of ppsci.arch.MLP on explicit weights (never default-initialised)."""

    def __init__(self, input_keys, output_keys, net: NetSpec, dtype=torch.float64, factor=None, weight_g=None,
                 fourier_kernel=None, act_beta=None):
        """factor: None | "weight_norm" | "random_weight" -- the hidden `net.weights` are then weight_v and
        `weight_g` lists one [out] vector per hidden layer (mlp.py:31-92); fourier_kernel: FourierEmbedding.kernel
        [in, dim/2] (mlp.py:117-136), `net.weights[0]` then has dim rows."""
        self.input_keys = tuple(input_keys)
        self.output_keys = tuple(output_keys)
        self.dtype = dtype
        self.activation = net.activation
        self.skip_connection = net.skip_connection
        self.periods = {self.input_keys[j]: w for j, w in net.periods.items()}
        self.weights = [torch.tensor(w, dtype=dtype, requires_grad=True) for w in net.weights]
        self.biases = [torch.tensor(b, dtype=dtype, requires_grad=True) for b in net.biases]
        self.factor = factor
        self.weight_g = [torch.tensor(g, dtype=dtype, requires_grad=True) for g in (weight_g or [])]
        # swish: one 0-D beta per hidden layer (activation.py:49-58); stan: one [H] beta per layer (:28-46)
        self.act_beta = [torch.tensor(b, dtype=dtype, requires_grad=True) for b in (act_beta or [])]
        self.fourier_kernel = (None if fourier_kernel is None
                               else torch.tensor(fourier_kernel, dtype=dtype, requires_grad=True))

    def parameters(self) -> List[torch.Tensor]:
        """Registration order of mlp.py:196-277: fourier_emb, linears (weight_v, weight_g, bias), last_fc."""
        out = [] if self.fourier_kernel is None else [self.fourier_kernel]
        n_hidden = len(self.weights) - 1
        for i, (w, b) in enumerate(zip(self.weights, self.biases)):
            if i == n_hidden:
                out += self.act_beta  # self.acts is registered before self.last_fc (mlp.py:262-263, :274)
            out += [w, self.weight_g[i], b] if self._factored(i) else [w, b]
        return out

    def _factored(self, i) -> bool:
        """weight_norm: hidden layers only; random_weight: also last_fc (mlp.py:239-249, :266-272)."""
        n_hidden = len(self.weights) - 1
        return bool(self.factor) and (i < n_hidden or (self.factor == "random_weight" and len(self.weight_g) > n_hidden))

    def _linear(self, i, y):
        w = self.weights[i]
        if self.factor == "weight_norm" and self._factored(i):  # mlp.py:50-54
            norm = torch.linalg.vector_norm(w, ord=2, dim=0, keepdim=True)
            w = self.weight_g[i] * w / norm
        elif self.factor == "random_weight" and self._factored(i):  # mlp.py:91-92
            w = self.weight_g[i] * w
        return y @ w + self.biases[i]

    def _act(self, y, i=0):
        if self.activation == "swish":
            return y * torch.sigmoid(self.act_beta[i] * y)
        if self.activation == "stan":
            return torch.tanh(y) * (1 + self.act_beta[i] * y)
        if self.activation == "tanh":
            return torch.tanh(y)
        if self.activation == "silu":
            return y * torch.sigmoid(y)  # activation.py:87-88
        if self.activation == "sin":
            return torch.sin(y)
        if self.activation == "siren":
            return torch.sin(30.0 * y)
        if self.activation == "cos":
            return torch.cos(y)
        if self.activation == "sigmoid":
            return torch.sigmoid(y)
        if self.activation == "gelu":
            return torch.nn.functional.gelu(y)  # nn.GELU() default: exact erf form
        F = torch.nn.functional  # activation.py:139-154: nn.ReLU(), nn.LeakyReLU() (0.01), nn.ELU() (alpha 1), nn.SELU()
        if self.activation == "relu":
            return F.relu(y)
        if self.activation == "leaky_relu":
            return F.leaky_relu(y, 0.01)
        if self.activation == "elu":
            return F.elu(y, 1.0)
        if self.activation == "selu":
            return F.selu(y)
        if self.activation == "identity":
            return y
        raise ValueError(self.activation)

    def forward_tensor(self, x):  # mlp.py:281-296
        y = x
        skip = None
        n_hidden = len(self.weights) - 1
        for i in range(n_hidden):
            y = self._linear(i, y)
            if self.skip_connection and i % 2 == 0:
                if skip is not None:
                    skip = y
                    y = y + skip
                else:
                    skip = y
            y = self._act(y, i)
        return self._linear(n_hidden, y)  # last_fc (factorised too under random_weight)

    def __call__(self, x: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:  # mlp.py:298-315
        if self.periods:
            y = dict(x)
            for k, w in self.periods.items():
                y[k] = torch.cat([torch.cos(w * x[k]), torch.sin(w * x[k])], dim=-1)
            x = y
        t = torch.cat([x[k] for k in self.input_keys], dim=-1)  # base.py:109-112
        if self.fourier_kernel is not None:  # mlp.py:308-309, :128-136
            t = torch.cat([torch.cos(t @ self.fourier_kernel), torch.sin(t @ self.fourier_kernel)], dim=-1)
        t = self.forward_tensor(t)
        outs = torch.split(t, 1, dim=-1)  # base.py:145-148
        return {k: v for k, v in zip(self.output_keys, outs)}


class PirateNet:
    """Restatement of ppsci.arch.PirateNet (mlp.py:530-820) on explicit named tensors (the reference's parameter names):
    PeriodEmbedding :95-114 -> FourierEmbedding :117-136 -> U, V = act(embed_{u,v}(x0)) :793-796 ->
    PirateNetBlock.forward :614-621 x num_blocks -> last_fc.  `state`: {name: array}; layers are RandomWeightFactorization
    (:57-92, W = weight_g * weight_v) when the names carry weight_v / weight_g, nn.Linear otherwise.  Pinned by
    tests/golden/piratenet.npz (tests/test_golden_piratenet.py)."""

    def __init__(self, input_keys, output_keys, state: Dict[str, np.ndarray], activation="tanh", periods=None,
                 dtype=torch.float64):
        self.input_keys, self.output_keys = tuple(input_keys), tuple(output_keys)
        self.activation, self.dtype = activation, dtype
        self.periods = {k: float(np.float32(2 * np.pi / float(p))) for k, (p, _) in (periods or {}).items()}
        self.names = list(state)
        self.t = {n: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=True) for n, v in state.items()}
        self.num_blocks = 1 + max(int(n.split(".")[1]) for n in state if n.startswith("blocks."))
        self.skip_connection = False

    def parameters(self) -> List[torch.Tensor]:
        return [self.t[n] for n in self.names]

    _act = MLP._act

    def _linear(self, name, y):
        t = self.t
        if name + ".weight_v" in t:
            return y @ (t[name + ".weight_g"] * t[name + ".weight_v"]) + t[name + ".bias"]
        return y @ t[name + ".weight"] + t[name + ".bias"]

    def __call__(self, x: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        if self.periods:
            y = dict(x)
            for k, w in self.periods.items():
                y[k] = torch.cat([torch.cos(w * x[k]), torch.sin(w * x[k])], dim=-1)
            x = y
        e = torch.cat([x[k] for k in self.input_keys], dim=-1)
        kern = self.t["fourier_emb.kernel"]
        h = torch.cat([torch.cos(e @ kern), torch.sin(e @ kern)], dim=-1)
        u = self._act(self._linear("embed_u.0", h))
        v = self._act(self._linear("embed_v.0", h))
        for i in range(self.num_blocks):
            pre = f"blocks.{i}."
            f = self._act(self._linear(pre + "linear1", h))
            z1 = f * u + (1 - f) * v
            g = self._act(self._linear(pre + "linear2", z1))
            z2 = g * u + (1 - g) * v
            hh = self._act(self._linear(pre + "linear3", z2))
            al = self.t[pre + "alpha"]
            h = al * hh + (1 - al) * h
        out = self._linear("last_fc", h)
        return {k: o for k, o in zip(self.output_keys, torch.split(out, 1, dim=-1))}


class ModifiedMLPN(PirateNet):
    """Restatement of ppsci.arch.ModifiedMLP (mlp.py:318-528) on explicit named tensors, any number of inputs:
    x0 = [period-embedded inputs] or its FourierEmbedding; U, V = act(embed_{u,v}(x0)); y <- act(linears.i(y)); y <- y*U + (1-y)*V;
    last_fc (forward_tensor :495-511).  Pinned by tests/golden/modified_mlp.npz (tests/test_golden_modified_mlp.py)."""

    def __init__(self, input_keys, output_keys, state, activation="tanh", periods=None, dtype=torch.float64):
        self.input_keys, self.output_keys = tuple(input_keys), tuple(output_keys)
        self.activation, self.dtype = activation, dtype
        self.periods = {k: float(np.float32(2 * np.pi / float(p))) for k, (p, _) in (periods or {}).items()}
        self.names = list(state)
        self.t = {n: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=True) for n, v in state.items()}
        self.num_layers = 1 + max(int(n.split(".")[1]) for n in state if n.startswith("linears."))
        self.skip_connection = False

    def __call__(self, x):
        if self.periods:
            y = dict(x)
            for k, w in self.periods.items():
                y[k] = torch.cat([torch.cos(w * x[k]), torch.sin(w * x[k])], dim=-1)
            x = y
        h = torch.cat([x[k] for k in self.input_keys], dim=-1)
        if "fourier_emb.kernel" in self.t:
            kern = self.t["fourier_emb.kernel"]
            h = torch.cat([torch.cos(h @ kern), torch.sin(h @ kern)], dim=-1)
        u = self._act(self._linear("embed_u.0", h))
        v = self._act(self._linear("embed_v.0", h))
        for i in range(self.num_layers):
            h = self._act(self._linear(f"linears.{i}", h))
            h = h * u + (1 - h) * v
        out = self._linear("last_fc", h)
        return {k: o for k, o in zip(self.output_keys, torch.split(out, 1, dim=-1))}


class ModelList:
    """ppsci.arch.ModelList (model_list.py:24-72): members share the input dict, outputs are merged."""

    def __init__(self, models: Sequence[MLP]):
        self.model_list = list(models)
        self.output_keys = tuple(k for m in self.model_list for k in m.output_keys)
        self.dtype = self.model_list[0].dtype

    def parameters(self) -> List[torch.Tensor]:
        return [p for m in self.model_list for p in m.parameters()]

    def __call__(self, x: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        y_all: Dict[str, torch.Tensor] = {}
        for m in self.model_list:
            y_all.update(m(x))
        return y_all


# ----------------------------------------------------------------------------- autodiff
def _grad(y, xs, create_graph=True):
    single = not isinstance(xs, (list, tuple))
    xs_l = [xs] if single else list(xs)
    g = torch.autograd.grad(
        y, xs_l, grad_outputs=torch.ones_like(y), create_graph=create_graph, allow_unused=True
    )
    g = [torch.zeros_like(x) if gi is None else gi for gi, x in zip(g, xs_l)]
    return g[0] if single else g


class _Jacobian:  # ad.py:30-77
    def __init__(self, ys, xs, J=None):
        self.ys, self.xs = ys, xs
        self.dim_y, self.dim_x = ys.shape[1], xs.shape[1]
        self.J = {} if J is None else J

    def __call__(self, i=0, j=None):
        if not 0 <= i < self.dim_y:
            raise ValueError(f"i({i}) should in range [0, {self.dim_y}).")
        if j is not None and not 0 <= j < self.dim_x:
            raise ValueError(f"j({j}) should in range [0, {self.dim_x}).")
        if i not in self.J:
            y = self.ys[:, i : i + 1] if self.dim_y > 1 else self.ys
            self.J[i] = _grad(y, self.xs)
        return self.J[i] if (j is None or self.dim_x == 1) else self.J[i][:, j : j + 1]


class Jacobians:  # ad.py:80-165
    def __init__(self):
        self.Js = {}

    def __call__(self, ys, xs, i=0, j=None):
        if not isinstance(xs, (list, tuple)):
            key = (id(ys), id(xs))
            if key not in self.Js:
                self.Js[key] = (_Jacobian(ys, xs), ys, xs)
            return self.Js[key][0](i, j)
        xs_require = [x for x in xs if (id(ys), id(x)) not in self.Js]
        grads_require = _grad(ys, xs_require) if xs_require else []
        idx, out = 0, []
        for k, x in enumerate(xs):
            key = (id(ys), id(x))
            assert x.shape[-1] == 1
            if key not in self.Js:
                self.Js[key] = (_Jacobian(ys, x, {0: grads_require[idx]}), ys, x)
                idx += 1
            out.append(self.Js[key][0](i, j))
        return out

    def _clear(self):
        self.Js = {}


class Hessians:  # ad.py:181-308
    def __init__(self, jac: Jacobians):
        self.Hs = {}
        self.jac = jac

    def __call__(self, ys, xs, component=None, i=0, j=0):
        key = (id(ys), id(xs), component)
        if key not in self.Hs:
            dim_y = ys.shape[1]
            if dim_y > 1:
                if component is None:
                    raise ValueError("component can not be None when dim_y>1.")
                if component >= dim_y:
                    raise ValueError("component should be smaller than dim_y.")
                comp = component
            else:
                if component is not None:
                    raise ValueError("component should be set to None when dim_y=1.")
                comp = 0
            grad_y = self.jac(ys, xs, i=comp, j=None)
            self.Hs[key] = (_Jacobian(grad_y, xs), ys, xs)
        return self.Hs[key][0](i, j)

    def _clear(self):
        self.Hs = {}


jacobian = Jacobians()
hessian = Hessians(jacobian)


def clear():  # ad.py:326-341
    jacobian._clear()
    hessian._clear()


# ----------------------------------------------------------------------------- symbolic
def cvt_to_key(expr) -> str:  # symbolic.py:111-137
    if isinstance(expr, sp.Function) and str(expr.func) == DETACH_FUNC_NAME:
        return f"{cvt_to_key(expr.args[0])}_{DETACH_FUNC_NAME}"
    if isinstance(expr, (sp.Symbol, sp.core.function.UndefinedFunction, sp.Function)):
        return expr.name if hasattr(expr, "name") else str(expr)
    if isinstance(expr, sp.Derivative):
        s = expr.args[0].name
        for symbol, order in expr.args[1:]:
            s += f"__{symbol}" * order
        return s
    return str(expr)


def post_traverse(cur, nodes):  # symbolic.py:507-534
    if isinstance(cur, sp.Function):
        for a in cur.args:
            nodes = post_traverse(a, nodes)
        nodes.append(cur)
    elif isinstance(cur, sp.Derivative):
        nodes = post_traverse(cur.args[0], nodes)
        nodes.append(cur)
    elif isinstance(cur, sp.Symbol):
        nodes.append(cur)
    elif isinstance(cur, sp.Number):
        nodes.append(cur)
    else:
        for a in cur.args:
            nodes = post_traverse(a, nodes)
        nodes.append(cur)
    return nodes


_UNARY = {
    sp.sin: torch.sin, sp.cos: torch.cos, sp.exp: torch.exp, sp.tanh: torch.tanh,
    sp.log: torch.log, sp.sqrt: torch.sqrt, sp.Abs: torch.abs, sp.sinh: torch.sinh,
    sp.cosh: torch.cosh, sp.tan: torch.tan,
    # the rest of SYMPY_TO_PADDLE, symbolic.py:79-108
    sp.asin: torch.asin, sp.acos: torch.acos, sp.atan: torch.atan, sp.asinh: torch.asinh, sp.acosh: torch.acosh,
    sp.atanh: torch.atanh, sp.erf: torch.erf, sp.loggamma: torch.lgamma, sp.sign: torch.sign,
    sp.ceiling: torch.ceil, sp.floor: torch.floor,
}


def lambdify(expr: sp.Basic, model: MLP, dtype=None,
             extra_parameters: Optional[Dict[str, torch.Tensor]] = None) -> Callable[[Dict[str, torch.Tensor]], torch.Tensor]:
    """symbolic.py:681-981 without derivative fusion (fusion does not change values,
    test/utils/test_symbolic.py:93-149)."""
    dtype = dtype or model.dtype
    expr = expr.subs(1.0, 1)  # symbolic.py:791
    nodes = post_traverse(expr, [])
    extra_parameters = extra_parameters or {}  # name -> 0-D tensor (ParameterNode, symbolic.py:471-485)
    nodes = [n for n in nodes if (not n.is_Symbol) or n.name in extra_parameters]  # symbolic.py:797-803
    nodes = list(dict.fromkeys(nodes))  # symbolic.py:806

    def run(data: Dict[str, torch.Tensor]) -> torch.Tensor:
        for n in nodes:
            key = cvt_to_key(n)
            if key in data:
                continue
            if n.is_Symbol:
                data[key] = extra_parameters[n.name]
            elif isinstance(n, sp.Derivative):  # symbolic.py:310-333
                val = data[cvt_to_key(n.args[0])]
                for sym, order in n.args[1:]:
                    order = int(order)
                    x = data[cvt_to_key(sym)]
                    if order & 1:
                        val = jacobian(val, x)
                        order -= 1
                    for _ in range(0, order, 2):
                        val = hessian(val, x)
                data[key] = val
            elif n.func == sp.Add:  # symbolic.py:225-229
                val = data[cvt_to_key(n.args[0])]
                for a in n.args[1:]:
                    val = val + data[cvt_to_key(a)]
                data[key] = val
            elif n.func == sp.Mul:  # symbolic.py:231-235
                val = data[cvt_to_key(n.args[0])]
                for a in n.args[1:]:
                    val = val * data[cvt_to_key(a)]
                data[key] = val
            elif n.func == sp.Pow:
                data[key] = torch.pow(data[cvt_to_key(n.args[0])], data[cvt_to_key(n.args[1])])
            elif isinstance(n, sp.Function) and str(n.func) == DETACH_FUNC_NAME:
                data[key] = data[cvt_to_key(n.args[0])].detach()
            elif n.func in (sp.Max, sp.Min):  # symbolic.py:241-262: pairwise left fold
                f = torch.maximum if n.func == sp.Max else torch.minimum
                val = data[cvt_to_key(n.args[0])]
                for a in n.args[1:]:
                    val = f(val, data[cvt_to_key(a)])
                data[key] = val
            elif n.func == sp.Heaviside:  # symbolic.py:237-239: paddle.heaviside(x, 0)
                x = data[cvt_to_key(n.args[0])]
                data[key] = torch.heaviside(x, torch.zeros((), dtype=x.dtype))
            elif n.func == sp.atan2:
                data[key] = torch.atan2(data[cvt_to_key(n.args[0])], data[cvt_to_key(n.args[1])])
            elif isinstance(n, sp.Function) and n.func in _UNARY:
                data[key] = _UNARY[n.func](data[cvt_to_key(n.args[0])])
            elif isinstance(n, sp.Function):  # LayerNode symbolic.py:406-430
                if str(n.func) in model.output_keys:
                    data.update(model(data))
                elif str(n.func) != "sdf":
                    raise ValueError(f"Node {n} can not match any model")
            elif n.is_Number or n.is_NumberSymbol:  # ConstantNode symbolic.py:433-468: 0-D fp32 tensor
                data[key] = torch.tensor(float(n), dtype=torch.float32).to(dtype)
            else:
                raise NotImplementedError(f"The node {n} is not supported in lambdify.")
        return data[cvt_to_key(nodes[-1])]

    return run


# ----------------------------------------------------------------------------- equations
def laplace_exprs(dim: int) -> Dict[str, sp.Basic]:  # laplace.py:40-55
    invars = sp.symbols("x y z")[:dim]
    u = sp.Function("u")(*invars)
    lap = 0
    for v in invars:
        lap += u.diff(v, 2)
    return {"laplace": lap}


def poisson_exprs(dim: int) -> Dict[str, sp.Basic]:  # poisson.py:40-53
    invars = sp.symbols("x y z")[:dim]
    p = sp.Function("p")(*invars)
    e = 0
    for v in invars:
        e += p.diff(v, 2)
    return {"poisson": e}


def navier_stokes_exprs(nu, rho, dim: int, time: bool) -> Dict[str, sp.Basic]:  # navier_stokes.py:70-151
    t, x, y, z = sp.symbols("t x y z")
    invars = (x, y)
    if time:
        invars = (t,) + invars
    if dim == 3:
        invars += (z,)
    u = sp.Function("u")(*invars)
    v = sp.Function("v")(*invars)
    w = sp.Function("w")(*invars) if dim == 3 else sp.Number(0)
    p = sp.Function("p")(*invars)
    cont = u.diff(x) + v.diff(y) + w.diff(z)

    def mom(k, xk):
        return (
            k.diff(t) + u * k.diff(x) + v * k.diff(y) + w * k.diff(z)
            - ((nu * k.diff(x)).diff(x) + (nu * k.diff(y)).diff(y) + (nu * k.diff(z)).diff(z))
            + 1 / rho * p.diff(xk)
        )

    eqs = {"continuity": cont, "momentum_x": mom(u, x), "momentum_y": mom(v, y)}
    if dim == 3:
        eqs["momentum_z"] = mom(w, z)
    return eqs


def allen_cahn_fn(eps: float):  # allen_cahn.py:56-64
    def allen_cahn(out):
        t, x = out["t"], out["x"]
        u = out["u"]
        u__t, u__x = jacobian(u, [t, x])
        u__x__x = jacobian(u__x, x)
        return u__t - (eps**2) * u__x__x + 5 * u * u * u - 5 * u

    return allen_cahn


# ----------------------------------------------------------------------------- loss
def mse_loss(output_dict, label_dict, weight_dict=None, reduction="mean", weight=None):  # mse.py:82-105
    losses = {}
    for key in label_dict:
        loss = (output_dict[key] - label_dict[key]) ** 2
        if weight_dict and key in weight_dict:
            loss = loss * weight_dict[key]
        if "area" in output_dict:
            loss = loss * output_dict["area"]
        loss = loss.sum() if reduction == "sum" else loss.mean()
        if isinstance(weight, (float, int)):
            loss = loss * weight
        elif isinstance(weight, dict) and key in weight:
            loss = loss * weight[key]
        losses[key] = loss
    return losses


def point_loss(kind, output_dict, label_dict, weight_dict=None, reduction="mean", weight=None):
    """L1Loss (l1.py:93-118), MAELoss (mae.py:85-108), L2Loss (l2.py:88-113), L2RelLoss (l2.py:280-310)."""
    losses = {}
    for key in label_dict:
        x, y = output_dict[key], label_dict[key]
        w = weight_dict[key] if (weight_dict and key in weight_dict) else None
        if kind in ("l1", "mae"):
            loss = (x - y).abs()
            if w is not None:
                loss = loss * w
            if "area" in output_dict:
                loss = loss * output_dict["area"]
            if kind == "l1":
                loss = loss.sum(dim=1)
        elif kind == "l2":
            loss = (x - y) ** 2
            if w is not None:
                loss = loss * w
            if "area" in output_dict:
                loss = loss * output_dict["area"]
            loss = loss.sum(dim=1).sqrt()
        elif kind == "l2rel":
            n = x.shape[0]
            loss = torch.linalg.norm((x - y).reshape(n, -1), dim=1) / torch.linalg.norm(y.reshape(n, -1), dim=1)
            if w is not None:
                loss = loss * w  # l2.py:296-297: [N] * [N, 1] broadcasts to [N, N] (pinned by tests/golden/variants.npz)
        else:
            raise ValueError(kind)
        loss = loss.sum() if reduction == "sum" else loss.mean()
        if isinstance(weight, (float, int)):
            loss = loss * weight
        elif isinstance(weight, dict) and key in weight:
            loss = loss * weight[key]
        losses[key] = loss
    return losses


def periodic_loss(kind, output_dict, label_dict, weight_dict=None, reduction="mean", weight=None):
    """PeriodicMSELoss (mse.py:322-355), PeriodicL1Loss (l1.py:185-218), PeriodicL2Loss (l2.py:181-207): first half of
    every output against its second half."""
    losses = {}
    for key in label_dict:
        n = len(output_dict[key])
        if n % 2 > 0:
            raise ValueError(f"Length of output({n}) of key({key}) should be even.")
        a, b = output_dict[key][:n // 2], output_dict[key][n // 2:]
        loss = (a - b).abs() if kind == "periodic_l1" else (a - b) ** 2
        if weight_dict and key in weight_dict:
            loss = loss * weight_dict[key]
        if "area" in output_dict:
            loss = loss * output_dict["area"]
        if kind == "periodic_l1":
            loss = loss.sum(dim=1)
        elif kind == "periodic_l2":
            loss = loss.sum(dim=1).sqrt()
        loss = loss.sum() if reduction == "sum" else loss.mean()
        if isinstance(weight, (float, int)):
            loss = loss * weight
        elif isinstance(weight, dict) and key in weight:
            loss = loss * weight[key]
        losses[key] = loss
    return losses


def causal_mse_loss(output_dict, label_dict, weight_dict=None, reduction="mean", weight=None, n_chunks=1, tol=1.0):
    """CausalMSELoss.forward, mse.py:158-187."""
    losses = {}
    for key in label_dict:
        loss = (output_dict[key] - label_dict[key]) ** 2  # F.mse_loss(..., "none")
        if weight_dict and key in weight_dict:
            loss = loss * weight_dict[key]
        if "area" in output_dict:
            loss = loss * output_dict["area"]
        acc_mat = torch.tril(torch.ones(n_chunks, n_chunks, dtype=loss.dtype), -1)
        loss_t = loss.reshape(n_chunks, -1)
        weight_t = torch.exp(-tol * (acc_mat @ loss_t.mean(-1, keepdim=True)))
        loss = loss_t * weight_t.detach()
        loss = loss.sum() if reduction == "sum" else loss.mean()
        if isinstance(weight, (float, int)):
            loss = loss * weight
        elif isinstance(weight, dict) and key in weight:
            loss = loss * weight[key]
        losses[key] = loss
    return losses


def loss_sum(losses: Dict[str, torch.Tensor]):  # mtl/sum.py:45-60
    total = 0.0
    for i, k in enumerate(losses):
        total = losses[k] if i == 0 else total + losses[k]
    return total


# ----------------------------------------------------------------------------- train forward
def train_forward(
    model: MLP,
    constraints: Sequence[dict],
):
    """expression.py:60-131.  Each constraint is a dict with
    input (name->np [N,1]), exprs (name->callable(data_dict)), label, weight (name->np [N,1] or None),
    reduction, loss_weight.  Returns (losses_all, losses_constraint, outputs_per_constraint)."""
    losses_all: Dict[str, torch.Tensor] = {}
    losses_constraint: Dict[str, float] = {}
    outputs = []
    for ci, c in enumerate(constraints):
        inp = {k: torch.tensor(np.asarray(v), dtype=model.dtype, requires_grad=True) for k, v in c["input"].items()}
        output_dict = model(inp)
        data = dict(inp)
        data.update(output_dict)
        for name, ex in c["exprs"].items():
            output_dict[name] = ex(data)
        clear()
        label = {k: torch.tensor(np.asarray(v), dtype=model.dtype) for k, v in c["label"].items()}
        wd = None
        if c.get("weight"):
            wd = {k: torch.tensor(np.asarray(v), dtype=model.dtype) for k, v in c["weight"].items()}
        if c.get("loss_kind", "mse") == "causal_mse":
            losses = causal_mse_loss(output_dict, label, wd, c.get("reduction", "mean"), c.get("loss_weight"),
                                     c["n_chunks"], c.get("tol", 1.0))
        elif c.get("loss_kind", "mse").startswith("periodic_"):
            losses = periodic_loss(c["loss_kind"], output_dict, label, wd, c.get("reduction", "mean"), c.get("loss_weight"))
        elif c.get("loss_kind", "mse") == "mse":
            losses = mse_loss(output_dict, label, wd, c.get("reduction", "mean"), c.get("loss_weight"))
        else:
            losses = point_loss(c["loss_kind"], output_dict, label, wd, c.get("reduction", "mean"), c.get("loss_weight"))
        name = c.get("name", f"c{ci}")
        losses_constraint[name] = 0.0
        for k in losses:
            losses_constraint[name] += float(losses[k].item())
            losses_all[k] = losses_all[k] + losses[k] if k in losses_all else losses[k]
        outputs.append(output_dict)
    return losses_all, losses_constraint, outputs


def loss_and_grads(model: MLP, constraints: Sequence[dict]):
    """train.py:117-158: forward, aggregate, backward.  Returns (total, losses_all, flat grad np)."""
    losses_all, losses_cst, outputs = train_forward(model, constraints)
    total = loss_sum(losses_all)
    params = model.parameters()
    grads = torch.autograd.grad(total, params, allow_unused=True)
    flat = np.concatenate(
        [(torch.zeros_like(p) if g is None else g).detach().numpy().ravel() for g, p in zip(grads, params)]
    )
    return float(total.item()), {k: float(v.item()) for k, v in losses_all.items()}, flat, outputs


# ----------------------------------------------------------------------------- optimizer
class Adam:
    """paddle.optimizer.Adam as wrapped by ppsci/optimizer/optimizer.py:225-248 (beta1=0.9, beta2=0.999,
    epsilon=1e-8, no weight decay): p -= lr*sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps*sqrt(1-b2^t))."""

    def __init__(self, n: int, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, dtype=np.float64):
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, eps
        self.m = np.zeros(n, dtype)
        self.v = np.zeros(n, dtype)
        self.t = 0

    def step(self, p: np.ndarray, g: np.ndarray, lr: Optional[float] = None) -> np.ndarray:
        lr = self.lr if lr is None else lr
        self.t += 1
        self.m = self.b1 * self.m + (1 - self.b1) * g
        self.v = self.b2 * self.v + (1 - self.b2) * g * g
        c2 = math.sqrt(1 - self.b2**self.t)
        lr_t = lr * c2 / (1 - self.b1**self.t)
        return p - lr_t * (self.m / (np.sqrt(self.v) + self.eps * c2))


class FirstOrder:
    """paddle.optimizer.{SGD, Momentum, RMSProp, AdamW} update rules on a flat numpy vector, as documented by
    PaddlePaddle for the classes that ppsci/optimizer/optimizer.py:39-176, :326-495 wrap (a float `weight_decay` of
    SGD / Momentum / RMSProp is L2Decay: coeff * p is added to the gradient; AdamW's is decoupled)."""

    def __init__(self, kind: str, n: int, lr=1e-3, dtype=np.float64, **kw):
        self.kind, self.lr, self.kw, self.t = kind, lr, kw, 0
        self.s = [np.zeros(n, dtype) for _ in range(3)]

    def step(self, p: np.ndarray, g: np.ndarray) -> np.ndarray:
        kw, lr = self.kw, self.lr
        self.t += 1
        if self.kind != "adamw":
            g = g + kw.get("weight_decay", 0.0) * p
        if self.kind == "sgd":
            return p - lr * g
        if self.kind == "momentum":
            mu = kw["momentum"]
            self.s[0] = mu * self.s[0] + g
            return p - lr * ((g + mu * self.s[0]) if kw.get("use_nesterov", False) else self.s[0])
        if self.kind == "rmsprop":
            rho, eps, mom = kw.get("rho", 0.95), kw.get("epsilon", 1e-6), kw.get("momentum", 0.0)
            self.s[0] = rho * self.s[0] + (1 - rho) * g * g
            mg = 0.0
            if kw.get("centered", False):
                self.s[2] = rho * self.s[2] + (1 - rho) * g
                mg = self.s[2]
            self.s[1] = mom * self.s[1] + lr * g / np.sqrt(self.s[0] - mg * mg + eps)
            return p - self.s[1]
        if self.kind == "adamw":
            b1, b2, eps, wd = kw.get("beta1", 0.9), kw.get("beta2", 0.999), kw.get("epsilon", 1e-8), kw.get("weight_decay", 0.001)
            p = p * (1.0 - lr * wd)
            self.s[0] = b1 * self.s[0] + (1 - b1) * g
            self.s[1] = b2 * self.s[1] + (1 - b2) * g * g
            c2 = math.sqrt(1 - b2**self.t)
            return p - lr * c2 / (1 - b1**self.t) * (self.s[0] / (np.sqrt(self.s[1]) + eps * c2))
        raise ValueError(self.kind)


def exponential_decay_lr(lr0: float, gamma: float, decay_steps: int, step: int, by_epoch=False) -> float:
    """ppsci/optimizer/lr_scheduler.py:212-269: paddle ExponentialDecay with gamma**(1/decay_steps) per step."""
    return lr0 * (gamma ** (1.0 / decay_steps)) ** step


# ----------------------------------------------------------------------------- SPINN (config 5)
class ModifiedMLP1:
    """One-input ModifiedMLP branch (ppsci/arch/mlp.py:488-507) on explicit weights.
    params: dict with wu,bu,wv,bv, w (list), b (list), wl, bl (numpy, [in,out] weights)."""

    def __init__(self, params: dict, activation: str = "tanh", dtype=torch.float64):
        self.act = activation
        cv = lambda a: torch.tensor(np.asarray(a), dtype=dtype, requires_grad=True)  # noqa: E731
        self.wu, self.bu, self.wv, self.bv = cv(params["wu"]), cv(params["bu"]), cv(params["wv"]), cv(params["bv"])
        self.w = [cv(a) for a in params["w"]]
        self.b = [cv(a) for a in params["b"]]
        self.wl, self.bl = cv(params["wl"]), cv(params["bl"])

    def parameters(self):
        out = [self.wu, self.bu, self.wv, self.bv]
        for w, b in zip(self.w, self.b):
            out += [w, b]
        return out + [self.wl, self.bl]

    def _a(self, y):
        return {"tanh": torch.tanh, "sin": torch.sin, "silu": lambda t: t * torch.sigmoid(t)}[self.act](y)

    def forward_tensor(self, x):  # mlp.py:488-507
        u = self._a(x @ self.wu + self.bu)
        v = self._a(x @ self.wv + self.bv)
        y = x
        for w, b in zip(self.w, self.b):
            y = self._a(y @ w + b)
            y = y * u + (1 - y) * v
        return y @ self.wl + self.bl


def spinn_helmholtz(branches, xs, k: float = 1.0, coeffs=None):
    """u and residual on the tensor-product grid (spinn.py:140-167, helmholtz.py:78-93).  The second derivatives
    are taken per branch output column by double backward -- mathematically what hvp_revrev's nested jvp with
    unit tangents returns per grid point.  Returns (u, res) as [nx,ny,nz] tensors."""
    f, f2 = [], []
    for net, x in zip(branches, xs):
        out = net.forward_tensor(x)  # [n, R]
        d2 = []
        for r in range(out.shape[1]):
            g = torch.autograd.grad(out[:, r].sum(), x, create_graph=True)[0]
            d2.append(torch.autograd.grad(g.sum(), x, create_graph=True)[0])
        f.append(out)
        f2.append(torch.cat(d2, dim=1))
    e = lambda a, b, c: torch.einsum("ir,jr,kr->ijk", a, b, c)  # noqa: E731
    u = e(f[0], f[1], f[2])
    uxx, uyy, uzz = e(f2[0], f[1], f[2]), e(f[0], f2[1], f[2]), e(f[0], f[1], f2[2])
    c = coeffs if coeffs is not None else (k**2, 1.0, 1.0, 1.0)
    return u, c[0] * u + c[1] * uxx + c[2] * uyy + c[3] * uzz


# ----------------------------------------------------------------------------- FNO (BASELINE config 4)
def reference_spectral_conv2d(x, w_re, w_im, n_modes_x, fft_norm="backward", bias=None):
    """Plain torch restatement of FactorizedSpectralConv.forward (fno_block.py:707-796) with the explicit
    fftshift / slicing / four-einsum sequence; the oracle of csrc/spectral_conv.hip."""
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


def fno_forward(x, P, n_layers, n_modes, norm=None, fft_norm="forward", eps=1e-5, domain_padding=None,
                domain_padding_mode="one-sided", stabilizer=None):
    """FNONet.forward (tfnonet.py:179-193) with FNOBlocks.forward_with_postactivation (fno_block.py:1191-1220)
    and fno_block.MLP (:313-320) written out on plain tensors.  P: dict of parameters named like the torch
    modules of paddlescience_amd.arch.fno (lifting.fcs.i.weight [Co,Ci,1,1] ...)."""
    import torch.nn.functional as F

    def conv1x1(x, w, b=None):
        y = torch.einsum("oi,bihw->bohw", w[:, :, 0, 0], x)
        return y if b is None else y + b.view(1, -1, 1, 1)

    def mlp(x, name, n):
        for i in range(n):
            x = conv1x1(x, P[f"{name}.fcs.{i}.weight"], P[f"{name}.fcs.{i}.bias"])
            if i < n - 1:
                x = F.gelu(x)
        return x

    n_lift = sum(1 for k in P if k.startswith("lifting.fcs.") and k.endswith(".weight"))
    x = mlp(x, "lifting", n_lift)
    # DomainPadding (fno_block.py:19-140): round(fraction * resolution) zero rows / columns behind (one-sided) or on both
    # sides (symmetric) of every spatial axis, between lifting and the blocks; removed again in front of the projection
    H0, W0 = x.shape[-2:]
    ph = pw = 0
    if domain_padding is not None:
        fr = [float(domain_padding)] * 2 if not isinstance(domain_padding, (list, tuple)) else [float(v) for v in domain_padding]
        ph, pw = round(fr[0] * H0), round(fr[1] * W0)
        sym = domain_padding_mode == "symmetric"
        x = F.pad(x, [pw if sym else 0, pw, ph if sym else 0, ph])
    for i in range(n_layers):
        skip = conv1x1(x, P[f"fno_blocks.fno_skips.{i}.weight"]) if f"fno_blocks.fno_skips.{i}.weight" in P else x
        xs = torch.tanh(x) if stabilizer == "tanh" else x  # fno_block.py:1199: the skip above sees x, the spectral branch tanh(x)
        y = reference_spectral_conv2d(xs, P[f"fno_blocks.convs.{i}.weight_real"], P[f"fno_blocks.convs.{i}.weight_imag"],
                                      n_modes[0], fft_norm, P[f"fno_blocks.convs.{i}.bias"])
        if norm == "group_norm":  # nn.GroupNorm(num_groups=1): statistics over (C, H, W) per sample
            mu = y.mean(dim=(1, 2, 3), keepdim=True)
            var = y.var(dim=(1, 2, 3), keepdim=True, unbiased=False)
            y = (y - mu) / torch.sqrt(var + eps)
            y = y * P[f"fno_blocks.norm.{i}.weight"].view(1, -1, 1, 1) + P[f"fno_blocks.norm.{i}.bias"].view(1, -1, 1, 1)
        x = y + skip
        if i < n_layers - 1:
            x = F.gelu(x)
    if ph or pw:
        oh, ow = (ph, pw) if domain_padding_mode == "symmetric" else (0, 0)
        x = x[..., oh:oh + H0, ow:ow + W0]
    return mlp(x, "projection", 2)


def uno_forward(x, P, out_channels, n_modes, scalings, skips_map=None, norm=None, eps=1e-5, domain_padding=None,
                domain_padding_mode="one-sided"):
    """UNONet.forward (/root/reference/ppsci/arch/unonet.py:246-289) on plain tensors; P named like paddlescience_amd.arch.uno.
    Per layer (FNOBlocks with n_layers = 1: fno_block.py:1191-1210, no activation behind it)
        x <- norm(SpectralConv_i(x) on the scaled grid) + bicubic(skip_i(x))
    SpectralConv.forward (fno_block.py:707-796): rfftn, fftshift of the rows, the centred block of n_modes rows x n_modes//2+1 columns
    times the weights, fftshift AGAIN, irfftn(s = the output grid) -- the crop / zero-pad at the end of both axes -- plus bias; the
    transforms run with the norm "backward" whatever fft_norm is (FNOBlocks does not hand it on, :1099-1111).  The scaled grid is
    round(size * factor) (:779-786), the last layer's the end-to-end grid (unonet.py:251-254, :273-274).  U skips: unonet.py:259-277."""
    import torch.nn.functional as F

    def conv1x1(x, w, b=None):
        y = torch.einsum("oi,bihw->bohw", w[:, :, 0, 0], x)
        return y if b is None else y + b.view(1, -1, 1, 1)

    n = len(out_channels)
    if skips_map is None:
        skips_map = {n - i - 1: i for i in range(n // 2)}
    x = conv1x1(F.gelu(conv1x1(x, P["lifting.fcs.0.weight"], P["lifting.fcs.0.bias"])), P["lifting.fcs.1.weight"], P["lifting.fcs.1.bias"])
    H0, W0 = x.shape[-2:]
    ph = pw = 0
    if domain_padding is not None:
        fr = [float(domain_padding)] * 2 if not isinstance(domain_padding, (list, tuple)) else [float(v) for v in domain_padding]
        ph, pw = round(fr[0] * H0), round(fr[1] * W0)
        sym = domain_padding_mode == "symmetric"
        x = F.pad(x, [pw if sym else 0, pw, ph if sym else 0, ph])
    e2e = [1.0, 1.0]
    for sc in scalings:
        e2e = [a * b for a, b in zip(e2e, sc)]
    final = (int(round(x.shape[-2] * e2e[0])), int(round(x.shape[-1] * e2e[1])))
    hs = {}
    for i in range(n):
        if i in skips_map:
            t = hs[skips_map[i]]
            if t.shape[-2:] != x.shape[-2:]:
                t = F.interpolate(t, size=tuple(x.shape[-2:]), mode="bicubic", align_corners=True)
            x = torch.cat([x, t], dim=1)
        H, W = x.shape[-2:]
        H2, W2 = final if i == n - 1 else (round(H * scalings[i][0]), round(W * scalings[i][1]))
        pre = f"fno_blocks.{i}."
        sk = conv1x1(x, P[pre + "fno_skips.0.weight"]) if pre + "fno_skips.0.weight" in P else x
        if (H, W) != (H2, W2):
            sk = F.interpolate(sk, size=(H2, W2), mode="bicubic", align_corners=True)
        X = torch.fft.fftshift(torch.fft.rfftn(x, dim=(-2, -1), norm="backward"), dim=-2)
        mx, my = n_modes[i][0], n_modes[i][1] // 2 + 1
        w = torch.complex(P[pre + "convs.0.weight_real"], P[pre + "convs.0.weight_imag"])
        st = H - min(H, mx)
        rows = slice(st // 2, -st // 2) if st else slice(None)
        out = torch.zeros(x.shape[0], w.shape[1], H, W // 2 + 1, dtype=X.dtype)
        out[:, :, rows, :my] = torch.einsum("bihw,iohw->bohw", X[:, :, rows, :my], w)
        out = torch.fft.fftshift(out, dim=-2)
        y = torch.fft.irfftn(out, s=(H2, W2), dim=(-2, -1), norm="backward") + P[pre + "convs.0.bias"].view(1, -1, 1, 1)
        if norm == "group_norm":
            mu = y.mean(dim=(1, 2, 3), keepdim=True)
            var = y.var(dim=(1, 2, 3), keepdim=True, unbiased=False)
            y = (y - mu) / torch.sqrt(var + eps)
            y = y * P[pre + "norm.0.weight"].view(1, -1, 1, 1) + P[pre + "norm.0.bias"].view(1, -1, 1, 1)
        x = y + sk
        if i in skips_map.values():
            hs[i] = conv1x1(x, P[f"horizontal_skips.{i}.weight"])
    if ph or pw:
        oh, ow = (ph, pw) if domain_padding_mode == "symmetric" else (0, 0)
        x = x[..., oh:oh + H0, ow:ow + W0]
    return conv1x1(F.gelu(conv1x1(x, P["projection.fcs.0.weight"], P["projection.fcs.0.bias"])), P["projection.fcs.1.weight"],
                   P["projection.fcs.1.bias"])


def sht_matrices(nlat, nlon, lmax, mmax):
    """The transform pair of /root/reference/ppsci/arch/paddle_harmonics/sht.py (grid "equiangular", norm "ortho") as dense real tables,
    built independently of the product's arch/sht_tables.py: orthonormal associated Legendre functions from scipy.special.lpmv (which
    carries the Condon-Shortley phase) times sqrt((2l+1)/(4 pi) (l-m)!/(l+m)!), Clenshaw-Curtis weights on theta_k = k pi/(nlat-1) from
    the moment equations sum_k w_k T_n(x_k) = int_{-1}^{1} T_n (exact for n < nlat).  Returns (A [m][l][k] for the analysis incl. the
    2 pi / nlon of rfft(norm="forward") * 2 pi, P [m][l][k] for the synthesis)."""
    import math

    import numpy as np
    import scipy.special

    theta = np.pi * np.arange(nlat) / (nlat - 1)
    x = np.cos(theta)
    n = np.arange(nlat)
    V = np.cos(np.outer(n, theta))  # T_n(x_k)
    mom = np.where(n % 2 == 0, 2.0 / (1.0 - n.astype(np.float64) ** 2), 0.0)
    w = np.linalg.solve(V, mom)
    P = np.zeros((mmax, lmax, nlat))
    for m in range(mmax):
        for l in range(m, lmax):
            nrm = math.sqrt((2 * l + 1) / (4 * math.pi) * math.factorial(l - m) / math.factorial(l + m))
            P[m, l] = nrm * scipy.special.lpmv(m, l, x)
    return torch.tensor(P * w[None, None, :] * (2.0 * math.pi / nlon)), torch.tensor(P)


def sfno_forward(x, P, n_layers, n_modes, norm=None, eps=1e-5):
    """SFNONet.forward (/root/reference/ppsci/arch/sfnonet.py:553-568): FNONet's lifting / blocks / projection with SphericalConv
    (:322-360) as the spectral convolution -- RealSHT onto n_modes[0] degrees x n_modes[1] // 2 orders (sht.py:118-150), complex weights
    per degree (`dhconv`, sfnonet.py:45-74), InverseRealSHT (sht.py:216-232), bias.  P named like paddlescience_amd.arch.fno.SFNONet."""
    import torch.nn.functional as F

    def conv1x1(x, w, b=None):
        y = torch.einsum("oi,bihw->bohw", w[:, :, 0, 0], x)
        return y if b is None else y + b.view(1, -1, 1, 1)

    x = conv1x1(F.gelu(conv1x1(x, P["lifting.fcs.0.weight"], P["lifting.fcs.0.bias"])), P["lifting.fcs.1.weight"], P["lifting.fcs.1.bias"])
    H, W = x.shape[-2:]
    L, M = n_modes[0], n_modes[1] // 2
    A, Pl = sht_matrices(H, W, L, M)
    A, Pl = A.to(x.dtype), Pl.to(x.dtype)
    for i in range(n_layers):
        skip = conv1x1(x, P[f"fno_blocks.fno_skips.{i}.weight"])
        X = torch.fft.rfft(x, dim=-1)[..., :M]  # unscaled: A carries 2 pi / W
        coef = torch.complex(torch.einsum("bckm,mlk->bclm", X.real, A), torch.einsum("bckm,mlk->bclm", X.imag, A))
        w = torch.complex(P[f"fno_blocks.convs.{i}.weight_real"], P[f"fno_blocks.convs.{i}.weight_imag"])
        coef = torch.einsum("bilm,iol->bolm", coef, w)
        T = torch.complex(torch.einsum("bclm,mlk->bckm", coef.real, Pl), torch.einsum("bclm,mlk->bckm", coef.imag, Pl))
        y = torch.fft.irfft(T, n=W, dim=-1, norm="forward") + P[f"fno_blocks.convs.{i}.bias"].view(1, -1, 1, 1)
        if norm == "group_norm":
            mu = y.mean(dim=(1, 2, 3), keepdim=True)
            var = y.var(dim=(1, 2, 3), keepdim=True, unbiased=False)
            y = (y - mu) / torch.sqrt(var + eps)
            y = y * P[f"fno_blocks.norm.{i}.weight"].view(1, -1, 1, 1) + P[f"fno_blocks.norm.{i}.bias"].view(1, -1, 1, 1)
        x = y + skip
        if i < n_layers - 1:
            x = F.gelu(x)
    return conv1x1(F.gelu(conv1x1(x, P["projection.fcs.0.weight"], P["projection.fcs.0.bias"])), P["projection.fcs.1.weight"],
                   P["projection.fcs.1.bias"])


def field_rel_error(x, y, order=0, p=2, spacing=(1.0, 1.0), fix=(False, False)):
    """Per-row relative error of /root/reference/examples/neuraloperator/metric.py on [B, C, H, W] tensors: LpLoss.rel
    (:148-160; order 0) and H1Loss.rel (:330-352; order 1: the squared norms of the central differences of :36-55 --
    periodic wrap-around, one-sided at the first / last sample under fix -- are added before the square roots).  Returns
    the [B, C] matrix of row terms (the classes then reduce it over `reduce_dims`)."""

    def diff(v, dim, h, fx):
        d = (torch.roll(v, -1, dims=dim) - torch.roll(v, 1, dims=dim)) / (2.0 * h)
        if fx:
            d = d.clone()
            first, second = v.select(dim, 0), v.select(dim, 1)
            last, prev = v.select(dim, -1), v.select(dim, -2)
            d.select(dim, 0).copy_((second - first) / h)
            d.select(dim, -1).copy_((last - prev) / h)
        return d

    def norm_p(v):
        return v.flatten(-2).abs().pow(p).sum(-1)

    e = x - y
    if order == 0:
        return (norm_p(e) / norm_p(y)) ** (1.0 / p) if p != 1 else norm_p(e) / norm_p(y)
    sd, sy = norm_p(e), norm_p(y)
    for dim, h, fx in ((-2, spacing[0], fix[0]), (-1, spacing[1], fix[1])):
        sd = sd + norm_p(diff(x, dim, h, fx) - diff(y, dim, h, fx))
        sy = sy + norm_p(diff(y, dim, h, fx))
    return sd.sqrt() / sy.sqrt()

