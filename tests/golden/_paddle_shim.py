"""TEST INFRASTRUCTURE, container-only.  A torch-backed stand-in for the handful of PaddlePaddle APIs that the
reference's HOT-PATH modules use (ppsci/arch/{base,mlp,activation}.py, ppsci/autodiff/ad.py,
ppsci/utils/{symbolic,expression}.py, ppsci/equation/pde/*.py, ppsci/loss/{mse,mtl/sum}.py), so that the
reference's own Python code can be executed here -- where PaddlePaddle cannot be installed -- to generate
golden vectors (tests/golden/make_hotpath_golden.py).

Mapping (paddle -> torch), each semantically equivalent for the calls on this path:
  paddle.grad(ys, xs, create_graph, retain_graph)  -> torch.autograd.grad(ys, xs, grad_outputs=ones_like(ys), ...)
                                                      (paddle's implicit cotangent is all-ones)
  nn.Linear(in, out): weight [in, out], y = x @ W + b (paddle layout)
  paddle.concat / split / to_tensor / cos / sin / ..., F.mse_loss(x, y, "none"), F.sigmoid, nn.Tanh ...
  Tensor.stop_gradient = False                      -> requires_grad_(True) on leaf tensors
Everything else resolves to an inert dummy so that unrelated import-time references do not fail."""
import sys
import types

import numpy as np
import torch

import _ref_import as RI

DTYPE = torch.float64  # fixture precision; paddle.get_default_dtype() still reports float32 to the geometry code


def _t(x):
    return x


class Layer:
    def __init__(self, *a, **k):
        object.__setattr__(self, "_params", {})
        object.__setattr__(self, "_subs", {})
        self.training = True

    def __setattr__(self, k, v):
        if isinstance(v, torch.Tensor) and getattr(v, "_is_param", False):
            self._params[k] = v
        elif isinstance(v, Layer):
            self._subs[k] = v
        object.__setattr__(self, k, v)

    def __call__(self, *a, **k):
        return self.forward(*a, **k)

    def create_parameter(self, shape, attr=None, dtype=None, is_bias=False, default_initializer=None):
        t = torch.zeros(tuple(shape), dtype=DTYPE)
        if default_initializer is not None:
            default_initializer(t)
        trainable = True if attr is None else getattr(attr, "trainable", True)
        t.requires_grad_(bool(trainable))
        t._is_param = True
        t.stop_gradient = not trainable
        return t

    def parameters(self):
        out = list(self._params.values())
        for s in self._subs.values():
            out += s.parameters()
        return out

    def named_parameters(self, prefix=""):
        out = [(prefix + k, v) for k, v in self._params.items()]
        for n, s in self._subs.items():
            out += s.named_parameters(prefix + n + ".")
        return out

    def add_sublayer(self, name, layer):
        self._subs[name] = layer
        return layer

    def train(self):
        self.training = True

    def eval(self):
        self.training = False


class LayerList(Layer):
    def __init__(self, layers=()):
        super().__init__()
        self._list = list(layers)
        for i, l in enumerate(self._list):
            self._subs[str(i)] = l

    def __iter__(self):
        return iter(self._list)

    def __getitem__(self, i):
        return self._list[i]

    def __len__(self):
        return len(self._list)

    def append(self, l):
        self._subs[str(len(self._list))] = l
        self._list.append(l)


class Sequential(Layer):
    """nn.Sequential(*layers): sublayers named "0", "1", ... applied in order."""

    def __init__(self, *layers):
        super().__init__()
        self._seq = list(layers)
        for i, l in enumerate(self._seq):
            self._subs[str(i)] = l

    def forward(self, x):
        for l in self._seq:
            x = l(x)
        return x


class ParameterList(Layer):
    def __init__(self, params=()):
        super().__init__()
        self._plist = list(params)
        for i, p in enumerate(self._plist):
            self._params[str(i)] = p

    def __iter__(self):
        return iter(self._plist)

    def __len__(self):
        return len(self._plist)

    def parameters(self):
        return list(self._plist)

    def state_dict(self):
        return {str(i): p for i, p in enumerate(self._plist)}


class Linear(Layer):
    def __init__(self, in_features, out_features, weight_attr=None, bias_attr=None, name=None):
        super().__init__()
        lim = float(np.sqrt(6.0 / (in_features + out_features)))
        w = (torch.rand(in_features, out_features, dtype=DTYPE) * 2 - 1) * lim
        w.requires_grad_(True)
        w._is_param = True
        b = torch.zeros(out_features, dtype=DTYPE, requires_grad=True)
        b._is_param = True
        self.weight, self.bias = w, b

    def forward(self, x):
        return x @ self.weight + self.bias


def _act(fn):
    class _A(Layer):
        def __init__(self, *a, **k):
            super().__init__()

        def forward(self, x):
            return fn(x)

    return _A


class _Const:
    def __init__(self, value=0.0):
        self.value = value

    def __call__(self, t):
        with torch.no_grad():  # parameters are created in paddle's default dtype, float32
            t.fill_(float(np.float32(self.value)))


def grad(outputs, inputs, grad_outputs=None, retain_graph=None, create_graph=False, only_inputs=True,
         allow_unused=False, no_grad_vars=None):
    single_out = not isinstance(outputs, (list, tuple))
    outs = [outputs] if single_out else list(outputs)
    single_in = not isinstance(inputs, (list, tuple))
    ins = [inputs] if single_in else list(inputs)
    go = [torch.ones_like(o) for o in outs] if grad_outputs is None else grad_outputs
    if retain_graph is None:
        retain_graph = create_graph
    g = torch.autograd.grad(outs, ins, grad_outputs=go, retain_graph=retain_graph, create_graph=create_graph,
                            allow_unused=True)
    g = [torch.zeros_like(x) if gi is None else gi for gi, x in zip(g, ins)]
    return g


def to_tensor(data, dtype=None, place=None, stop_gradient=True):
    if isinstance(data, torch.Tensor):
        return data
    if isinstance(data, (float, int)):
        # paddle.to_tensor(python float) -> default dtype float32 (this is how ConstantNode rounds constants)
        return torch.tensor(float(np.float32(data)), dtype=DTYPE)
    return torch.tensor(np.asarray(data), dtype=DTYPE)


def install():
    RI.install()  # dummy finder for everything not defined below + bare `ppsci` packages
    paddle = sys.modules.get("paddle") or __import__("paddle")
    nn = __import__("paddle.nn", fromlist=["x"])
    F = __import__("paddle.nn.functional", fromlist=["x"])
    jit = __import__("paddle.jit", fromlist=["x"])

    paddle.Tensor = torch.Tensor
    paddle.get_default_dtype = lambda: "float32"
    paddle.grad = grad
    paddle.to_tensor = to_tensor
    paddle.concat = lambda xs, axis=0: torch.cat(list(xs), dim=axis)
    paddle.split = lambda x, n, axis=0: list(torch.split(x, x.shape[axis] // n if isinstance(n, int) else n, dim=axis))
    for name in ("sin", "cos", "exp", "tanh", "log", "sqrt", "abs", "sinh", "cosh", "tan", "sign", "ceil", "floor",
                 "maximum", "minimum", "pow", "zeros", "ones", "zeros_like", "ones_like", "full_like"):
        setattr(paddle, name, getattr(torch, name))
    paddle.heaviside = lambda x, y: torch.heaviside(x, y.to(x.dtype))
    paddle.broadcast_to = lambda x, shape: torch.broadcast_to(x, tuple(shape))
    paddle.no_grad = torch.no_grad
    paddle.ParamAttr = type("ParamAttr", (), {"__init__": lambda self, trainable=True, **k: setattr(self, "trainable", trainable)})
    nn.Layer, nn.LayerList, nn.ParameterList, nn.Linear = Layer, LayerList, ParameterList, Linear
    nn.Sequential = Sequential
    nn.Tanh, nn.Sigmoid, nn.Identity = _act(torch.tanh), _act(torch.sigmoid), _act(lambda x: x)
    nn.ReLU, nn.ELU, nn.SELU, nn.GELU = _act(torch.relu), _act(torch.nn.functional.elu), _act(torch.selu), _act(torch.nn.functional.gelu)
    nn.LeakyReLU = _act(torch.nn.functional.leaky_relu)
    init = __import__("paddle.nn.initializer", fromlist=["x"])
    init.Constant = _Const
    nn.initializer = init
    F.sigmoid, F.tanh = torch.sigmoid, torch.tanh
    F.mse_loss = lambda x, y, reduction="mean": ((x - y) ** 2 if reduction == "none" else torch.nn.functional.mse_loss(x, y, reduction=reduction))
    F.linear = lambda x, w, b=None: x @ w + (0 if b is None else b)
    jit.to_static = lambda f=None, **k: (f if f is not None else (lambda g: g))
    nn.functional = F
    paddle.nn, paddle.jit = nn, jit
    # paddle.Tensor has no in-place `+=` / `*=` dunder methods: `a += b` rebinds `a = a + b` (this is what makes
    # OperatorNode._add/_mul_operator_func, symbolic.py:225-235, safe).  torch's are in-place, so they are
    # redirected for this fixture-generation process only.
    torch.Tensor.__iadd__ = lambda a, b: a + b
    torch.Tensor.__imul__ = lambda a, b: a * b
    torch.Tensor.__isub__ = lambda a, b: a - b
    torch.Tensor.__itruediv__ = lambda a, b: a / b
    # torch tensors accept arbitrary attributes, so `x.stop_gradient = False` is harmless; leaf inputs are
    # created with requires_grad=True by the fixture generator.
    return paddle


def import_hotpath():
    """Imports the reference hot-path modules (real files under /root/reference/ppsci)."""
    import importlib

    install()
    init = types.ModuleType("ppsci.utils.initializer")  # only used by unrelated layers at call time
    for n in ("zeros_", "ones_", "constant_", "uniform_", "normal_", "trunc_normal_", "kaiming_uniform_",
              "kaiming_normal_", "xavier_uniform_", "xavier_normal_", "linear_init_", "conv_init_", "glorot_normal_"):
        setattr(init, n, lambda *a, **k: None)
    sys.modules["ppsci.utils.initializer"] = init
    sys.modules["ppsci.utils"].initializer = init
    ad = importlib.import_module("ppsci.autodiff.ad")
    autod = sys.modules["ppsci.autodiff"]
    autod.jacobian, autod.hessian, autod.clear = ad.jacobian, ad.hessian, ad.clear
    base = importlib.import_module("ppsci.arch.base")
    act = importlib.import_module("ppsci.arch.activation")
    mlp = importlib.import_module("ppsci.arch.mlp")
    arch = sys.modules["ppsci.arch"]
    arch.Arch, arch.MLP, arch.activation, arch.base, arch.mlp = base.Arch, mlp.MLP, act, base, mlp
    arch.ModelList = type("ModelList", (), {})
    sys.modules["ppsci"].arch = arch
    pde_base = importlib.import_module("ppsci.equation.pde.base")
    eq = sys.modules["ppsci.equation"]
    eq.DETACH_FUNC_NAME = pde_base.DETACH_FUNC_NAME
    eq.PDE = pde_base.PDE
    sys.modules["ppsci"].equation = eq
    sys.modules["ppsci"].autodiff = autod
    symbolic = importlib.import_module("ppsci.utils.symbolic")
    sys.modules["ppsci.utils"].symbolic = symbolic
    out = dict(ad=ad, mlp=mlp, symbolic=symbolic, pde_base=pde_base)
    for name in ("laplace", "allen_cahn", "navier_stokes", "poisson"):
        out[name] = importlib.import_module(f"ppsci.equation.pde.{name}")
    loss_base = importlib.import_module("ppsci.loss.base")
    sys.modules["ppsci.loss"].base = loss_base
    out["mse"] = importlib.import_module("ppsci.loss.mse")
    return out
