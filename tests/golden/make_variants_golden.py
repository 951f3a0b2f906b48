"""Generates tests/golden/variants.npz by executing the REFERENCE's own code for the MLP variants and helpers built
after the core path (/root/reference/ppsci/arch/mlp.py WeightNormLinear / RandomWeightFactorization /
FourierEmbedding, arch/activation.py Swish / Stan, arch/model_list.py, loss/mse.py CausalMSELoss,
equation/pde/viv.py + utils/symbolic.py ParameterNode) in this container, PaddlePaddle replaced by the torch-backed
shim (tests/golden/_paddle_shim.py + the few additions below), in float64.

    python tests/golden/make_variants_golden.py

Per case: explicit parameter values in the REFERENCE's `parameters()` order (so the fixture also pins that order),
points, residuals per point, loss terms and d(total loss)/d(parameters)."""
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import _paddle_shim as S  # noqa: E402

D = torch.float64


class NamedParam(torch.Tensor):
    """A leaf tensor with a writable `.name` (paddle parameters have one; ParameterNode matches symbols by it)."""

    name = None


def extend_shim(paddle):
    nn = sys.modules["paddle.nn"]
    init = sys.modules["paddle.nn.initializer"]

    class Normal:
        def __init__(self, mean=0.0, std=1.0):
            self.mean, self.std = mean, std

        def __call__(self, t):
            with torch.no_grad():
                t.copy_(torch.randn(t.shape, dtype=t.dtype) * self.std + self.mean)

    init.Normal = Normal

    def assign(src, dst):
        with torch.no_grad():
            dst.copy_(src)

    paddle.assign = assign
    paddle.tril = lambda x, diagonal=0: torch.tril(x, diagonal)
    paddle.ones = lambda shape, dtype=None: torch.ones(tuple(shape), dtype=D)
    _norm = torch.Tensor.norm
    torch.Tensor.norm = lambda self, p=2, axis=None, keepdim=False, **k: _norm(self, p=p, dim=axis, keepdim=keepdim)
    S.Layer.register_buffer = lambda self, name, t, persistable=True: object.__setattr__(self, name, t)
    S.ParameterList.append = lambda self, p: (self._plist.append(p), self._params.__setitem__(str(len(self._plist) - 1), p))[0]
    count = [0]

    def create_parameter(shape, dtype=None, default_initializer=None, **k):
        t = torch.zeros(tuple(shape), dtype=D)
        if default_initializer is not None:
            default_initializer(t)
        t = torch.Tensor._make_subclass(NamedParam, t, True)
        t.name = f"create_parameter_{count[0]}.w_0"
        t._is_param = True
        count[0] += 1
        return t

    paddle.create_parameter = create_parameter
    nn.functional.sigmoid = torch.sigmoid


def set_params(params, rng, scale=0.3):
    """Fill the trainable parameters (reference order) with fp32-representable values; returns the flat vector."""
    vals = []
    with torch.no_grad():
        for p in params:
            v = rng.uniform(-scale, scale, tuple(p.shape)).astype(np.float32).astype(np.float64)
            if p.dim() <= 1 and p.numel() > 0 and getattr(p, "_positive", False):
                v = np.abs(v) + 0.7
            p.copy_(torch.tensor(v))
            vals.append(np.asarray(v).ravel())
    return np.concatenate(vals)


def grads_flat(total, params):
    g = torch.autograd.grad(total, params, allow_unused=True, retain_graph=True)
    return np.concatenate([(torch.zeros_like(p) if gi is None else gi).detach().numpy().ravel() for gi, p in zip(g, params)])


def main():
    import sympy as sp

    mods = S.import_hotpath()
    paddle = sys.modules["paddle"]
    extend_shim(paddle)
    MLP, lambdify, clear = mods["mlp"].MLP, mods["symbolic"].lambdify, mods["ad"].clear
    mse = mods["mse"]
    out = {}

    def trainable(model):
        return [p for p in model.parameters() if p.requires_grad]

    def allen_cahn_case(name, model, n, rng, loss_obj, sort_t=False, positive=(), scale=0.3):
        ps = trainable(model)
        names = [k for k, p in model.named_parameters() if p.requires_grad]
        for k, p in zip(names, ps):
            p._positive = any(tag in k for tag in positive)
        flat = set_params(ps, rng, scale)
        X = rng.uniform([0, -1], [1, 1], (n, 2)).astype(np.float32).astype(np.float64)
        if sort_t:
            X[:, 0] = np.sort(X[:, 0])
        data = {"t": torch.tensor(X[:, :1], requires_grad=True), "x": torch.tensor(X[:, 1:], requires_grad=True)}
        eq = mods["allen_cahn"].AllenCahn(0.05)
        od = model(data)
        dd = dict(data)
        dd.update(od)
        od["allen_cahn"] = eq.equations["allen_cahn"](dd)
        clear()
        label = {"allen_cahn": torch.tensor(rng.standard_normal((n, 1)).astype(np.float32).astype(np.float64) * 0.05)}
        losses = loss_obj(od, label, None)
        total = losses["allen_cahn"]
        out[f"{name}/X"], out[f"{name}/params"] = X, flat
        out[f"{name}/names"] = np.array(names)
        out[f"{name}/res"] = od["allen_cahn"].detach().numpy()[:, 0]
        out[f"{name}/label"] = label["allen_cahn"].numpy()[:, 0]
        out[f"{name}/loss"] = np.asarray(float(total.detach()))
        out[f"{name}/grad"] = grads_flat(total, ps)
        print(name, float(total.detach()), np.linalg.norm(out[f"{name}/grad"]))

    rng = np.random.default_rng(77)
    allen_cahn_case("weight_norm", MLP(("t", "x"), ("u",), None, (24, 24, 24), "tanh", weight_norm=True), 40, rng,
                    mse.MSELoss("mean"), positive=("weight_g",))
    allen_cahn_case("fourier_rwf_periods",
                    MLP(("t", "x"), ("u",), None, (24, 24, 24), "tanh", periods={"x": (2.0, False)},
                        fourier={"dim": 24, "scale": 1.0}, random_weight={"mean": 0.5, "std": 0.1}), 44, rng,
                    mse.MSELoss("mean"), positive=("weight_g",))
    allen_cahn_case("swish", MLP(("t", "x"), ("u",), None, (20, 20, 20), "swish"), 36, rng, mse.MSELoss("sum"),
                    positive=("beta",))
    allen_cahn_case("stan", MLP(("t", "x"), ("u",), None, (20, 20), "stan"), 36, rng, mse.MSELoss("mean"),
                    positive=("beta",))
    allen_cahn_case("causal", MLP(("t", "x"), ("u",), None, (16, 16), "tanh"), 48, rng,
                    mse.CausalMSELoss(8, "mean", tol=1.5), sort_t=True)

    for act in ("siren", "gelu", "sigmoid", "cos"):
        allen_cahn_case(f"act_{act}", MLP(("t", "x"), ("u",), None, (20, 20, 20), act), 30, rng, mse.MSELoss("mean"),
                        scale=0.03 if act == "siren" else 0.3)  # siren multiplies by w0 = 30 (activation.py Siren)

    # ---- the other point losses (loss/l1.py, loss/mae.py, loss/l2.py) on an Allen-Cahn residual + a data term
    F = sys.modules["paddle.nn.functional"]
    F.l1_loss = lambda x, y, reduction="mean": ((x - y).abs() if reduction == "none" else torch.nn.functional.l1_loss(x, y, reduction=reduction))
    _sum, _mean = torch.Tensor.sum, torch.Tensor.mean
    torch.Tensor.sum = lambda self, axis=None, keepdim=False, **k: _sum(self) if axis is None and not k else _sum(self, dim=k.get("dim", axis), keepdim=keepdim)
    paddle.linalg = type("linalg", (), {"norm": staticmethod(lambda x, p=2, axis=None: torch.linalg.norm(x, ord=p, dim=axis))})
    paddle.norm = lambda x, p=2, axis=None: torch.linalg.norm(x, ord=p, dim=axis)
    l1m, maem, l2m = (importlib.import_module(f"ppsci.loss.{m}") for m in ("l1", "mae", "l2"))
    for lname, cls in (("l1", l1m.L1Loss), ("mae", maem.MAELoss), ("l2", l2m.L2Loss), ("l2rel", l2m.L2RelLoss)):
        for red in ("mean", "sum"):
            model = MLP(("t", "x"), ("u",), None, (16, 16), "tanh")
            ps = trainable(model)
            flat = set_params(ps, rng)
            n = 37
            X = rng.uniform([0, -1], [1, 1], (n, 2)).astype(np.float32).astype(np.float64)
            data = {"t": torch.tensor(X[:, :1], requires_grad=True), "x": torch.tensor(X[:, 1:], requires_grad=True)}
            eq = mods["allen_cahn"].AllenCahn(0.05)
            od = model(data)
            dd = dict(data)
            dd.update(od)
            od["allen_cahn"] = eq.equations["allen_cahn"](dd)
            clear()
            lab = {"allen_cahn": torch.tensor(rng.standard_normal((n, 1)).astype(np.float32).astype(np.float64) + 3.0),
                   "u": torch.tensor(rng.standard_normal((n, 1)).astype(np.float32).astype(np.float64) + 2.0)}
            wts = {k: torch.tensor(rng.uniform(0.5, 2.0, (n, 1)).astype(np.float32).astype(np.float64)) for k in lab}
            losses = cls(red, weight={"u": 0.7})(od, lab, wts)
            total = losses["allen_cahn"] + losses["u"]
            nm = f"loss_{lname}_{red}"
            out[f"{nm}/X"], out[f"{nm}/params"] = X, flat
            for k in lab:
                out[f"{nm}/label/{k}"], out[f"{nm}/weight/{k}"] = lab[k].numpy()[:, 0], wts[k].numpy()[:, 0]
                out[f"{nm}/loss/{k}"] = np.asarray(float(losses[k].detach()))
            out[f"{nm}/grad"] = grads_flat(total, ps)
            print(nm, float(total.detach()))

    # ---- Periodic{MSE,L1,L2}Loss (mse.py:269-355, l1.py:123-218, l2.py:118-207): batch = [x = -1 side ; x = +1 side]
    for lname, cls in (("mse", mse.PeriodicMSELoss), ("l1", l1m.PeriodicL1Loss), ("l2", l2m.PeriodicL2Loss)):
        for red in ("mean", "sum"):
            model = MLP(("t", "x"), ("u",), None, (16, 16), "tanh")
            ps = trainable(model)
            flat = set_params(ps, rng)
            h = 19
            tcol = rng.uniform(0, 1, (h, 1)).astype(np.float32).astype(np.float64)
            X = np.concatenate([np.concatenate([tcol, np.full((h, 1), -1.0)], 1), np.concatenate([tcol, np.full((h, 1), 1.0)], 1)])
            data = {"t": torch.tensor(X[:, :1], requires_grad=True), "x": torch.tensor(X[:, 1:], requires_grad=True)}
            eq = mods["allen_cahn"].AllenCahn(0.05)
            od = model(data)
            dd = dict(data)
            dd.update(od)
            od["allen_cahn"] = eq.equations["allen_cahn"](dd)
            clear()
            lab = {k: torch.zeros(2 * h, 1, dtype=D) for k in ("u", "allen_cahn")}
            losses = cls(red, weight={"u": 0.7})(od, lab, None)
            total = losses["allen_cahn"] + losses["u"]
            nm = f"periodic_{lname}_{red}"
            out[f"{nm}/X"], out[f"{nm}/params"] = X, flat
            for k in lab:
                out[f"{nm}/loss/{k}"] = np.asarray(float(losses[k].detach()))
            out[f"{nm}/grad"] = grads_flat(total, ps)
            print(nm, float(total.detach()))

    # ---- every function of SYMPY_TO_PADDLE (symbolic.py:79-108) in one residual
    # (the dummy paddle module handed the map placeholders for the functions the shim had not defined)
    for sf, tname in ((sp.asin, "asin"), (sp.acos, "acos"), (sp.atan, "atan"), (sp.atan2, "atan2"), (sp.asinh, "asinh"),
                      (sp.acosh, "acosh"), (sp.atanh, "atanh"), (sp.erf, "erf"), (sp.loggamma, "lgamma")):
        mods["symbolic"].SYMPY_TO_PADDLE[sf] = getattr(torch, tname)
    model = MLP(("x", "y"), ("u",), None, (20, 20), "tanh")
    ps = trainable(model)
    flat = set_params(ps, rng)
    n = 48
    X = rng.uniform(0.1, 0.9, (n, 2)).astype(np.float32).astype(np.float64)
    data = {"x": torch.tensor(X[:, :1], requires_grad=True), "y": torch.tensor(X[:, 1:], requires_grad=True)}
    x, y = sp.symbols("x y")
    u = sp.Function("u")(x, y)
    expr = (sp.atan(u.diff(x)) + sp.erf(u) * sp.asinh(u.diff(y, 2)) + sp.atan2(u.diff(x, 2), 1 + x * x)
            + sp.asin(u / 4) * sp.acos(x / 2) + sp.atanh(y / 2) * u + sp.acosh(2 + u * u)
            + sp.loggamma(2 + y) * u.diff(y) + sp.floor(4 * x) * u + sp.ceiling(3 * y) * u.diff(x)
            + sp.Max(u, u.diff(x), 0.1) + sp.Min(u.diff(y), x) + sp.Heaviside(x - 0.5) * u + sp.sign(y - 0.4) * u)
    od = model(data)
    dd = dict(data)
    dd.update(od)
    od["r"] = lambdify(expr, model, fuse_derivative=True)(dd)
    clear()
    losses = mse.MSELoss("mean")(od, {"r": torch.zeros(n, 1, dtype=D)}, None)
    out["sympy_map/X"], out["sympy_map/params"] = X, flat
    out["sympy_map/res"] = od["r"].detach().numpy()[:, 0]
    out["sympy_map/loss"] = np.asarray(float(losses["r"].detach()))
    out["sympy_map/grad"] = grads_flat(losses["r"], ps)
    print("sympy_map", float(losses["r"].detach()))

    # ---- ModelList: two nets coupled in sympy residuals
    ml = importlib.import_module("ppsci.arch.model_list")
    ma = MLP(("x", "y"), ("u", "v"), None, (20, 20), "tanh")
    mb = MLP(("x", "y"), ("p",), None, (16, 16, 16), "silu")
    model = ml.ModelList((ma, mb))
    ps = trainable(model)
    flat = set_params(ps, rng)
    n = 39
    X = rng.uniform(-1, 1, (n, 2)).astype(np.float32).astype(np.float64)
    data = {"x": torch.tensor(X[:, :1], requires_grad=True), "y": torch.tensor(X[:, 1:], requires_grad=True)}
    x, y = sp.symbols("x y")
    u, v, p = (sp.Function(k)(x, y) for k in ("u", "v", "p"))
    nu = 0.05
    exprs = {"continuity": u.diff(x) + v.diff(y),
             "momentum_x": u * u.diff(x) + v * u.diff(y) - nu * (u.diff(x, 2) + u.diff(y, 2)) + p.diff(x)}
    od = model(data)
    dd = dict(data)
    dd.update(od)
    for k, e in exprs.items():
        od[k] = lambdify(e, model, fuse_derivative=True)(dd)
    clear()
    losses = mse.MSELoss("sum")(od, {k: torch.zeros(n, 1, dtype=D) for k in exprs}, None)
    total = losses["continuity"] + losses["momentum_x"]
    out["model_list/X"], out["model_list/params"] = X, flat
    out["model_list/names"] = np.array([k for k, q in model.named_parameters() if q.requires_grad])
    for k in exprs:
        out[f"model_list/res/{k}"] = od[k].detach().numpy()[:, 0]
        out[f"model_list/loss/{k}"] = np.asarray(float(losses[k].detach()))
    out["model_list/grad"] = grads_flat(total, ps)
    print("model_list", float(total.detach()))

    # ---- learnable equation parameters (Vibration)
    viv = importlib.import_module("ppsci.equation.pde.viv")
    eq = viv.Vibration(2.0, 0.7, -0.4)
    model = MLP(("t_f",), ("eta",), None, (20, 20), "tanh")
    ps = trainable(model)
    flat = set_params(ps, rng)
    n = 41
    t = rng.uniform(0, 1, (n, 1)).astype(np.float32).astype(np.float64)
    data = {"t_f": torch.tensor(t, requires_grad=True)}
    od = model(data)
    dd = dict(data)
    dd.update(od)
    od["f"] = lambdify(eq.equations["f"], model, extra_parameters=list(eq.learnable_parameters), fuse_derivative=True)(dd)
    clear()
    lab = {"eta": torch.tensor(rng.standard_normal((n, 1)).astype(np.float32).astype(np.float64) * 0.1),
           "f": torch.tensor(rng.standard_normal((n, 1)).astype(np.float32).astype(np.float64))}
    losses = mse.MSELoss("mean")(od, lab, None)
    total = losses["eta"] + losses["f"]
    ks = list(eq.learnable_parameters)
    out["viv/X"], out["viv/params"] = t, flat
    out["viv/label/eta"], out["viv/label/f"] = lab["eta"].numpy()[:, 0], lab["f"].numpy()[:, 0]
    out["viv/res/f"] = od["f"].detach().numpy()[:, 0]
    out["viv/loss/eta"], out["viv/loss/f"] = np.asarray(float(losses["eta"].detach())), np.asarray(float(losses["f"].detach()))
    out["viv/grad"] = grads_flat(total, ps)
    out["viv/grad_k"] = grads_flat(total, ks)
    print("viv", float(total.detach()), out["viv/grad_k"])
    np.savez_compressed(os.path.join(HERE, "variants.npz"), **out)


if __name__ == "__main__":
    main()
