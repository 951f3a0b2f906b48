"""Generates tests/golden/piratenet.npz by executing the REFERENCE's own PirateNet (ppsci/arch/mlp.py:530-820 -- PeriodEmbedding,
FourierEmbedding, RandomWeightFactorization, PirateNetBlock), autodiff/ad.py, utils/symbolic.py and loss/mse.py in float64 under
the torch-backed paddle shim (tests/golden/_paddle_shim.py): per-point residuals, loss terms and the gradient of the total loss
with respect to every named parameter, for seeded parameter values (alpha != 0, so that both branches of the residual
connection are live).

    python tests/golden/make_piratenet_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

CASES = {
    # allen_cahn_piratenet.py in small: periodic x, Fourier features, random weight factorisation, u_t and u_xx
    "allen_cahn_rwf": dict(inputs=("t", "x"), outputs=("u",), blocks=2, hidden=32, act="tanh", periods={"x": (2.0, False)},
                           fourier={"dim": 32, "scale": 2.0}, rwf={"mean": 1.0, "std": 0.1}, eq="allen_cahn", n=40,
                           lo=[0, -1], hi=[1, 1], reduction="mean"),
    "two_out_gelu": dict(inputs=("x", "y"), outputs=("u", "v"), blocks=1, hidden=16, act="gelu", periods=None,
                         fourier={"dim": 16, "scale": 1.0}, rwf=None, eq="two_out", n=33, lo=[-1, -1], hi=[1, 1],
                         reduction="sum"),
    "three_blocks_silu": dict(inputs=("x", "y", "t"), outputs=("u",), blocks=3, hidden=24, act="silu",
                              periods={"x": (1.5, False), "y": (3.0, False)}, fourier={"dim": 24, "scale": 1.0}, rwf=None,
                              eq="heat", n=21, lo=[-1, -1, 0], hi=[1, 1, 1], reduction="mean"),
    "values_only_sin": dict(inputs=("x",), outputs=("u",), blocks=1, hidden=16, act="sin", periods=None,
                            fourier={"dim": 16, "scale": 3.0}, rwf={"mean": 0.5, "std": 0.1}, eq="value", n=19, lo=[-2],
                            hi=[2], reduction="mean"),
}


def equations(c):
    import sympy as sp

    syms = sp.symbols(" ".join(c["inputs"]))
    syms = syms if isinstance(syms, tuple) else (syms,)
    if c["eq"] == "allen_cahn":
        t, x = syms
        u = sp.Function("u")(t, x)
        return {"allen_cahn": u.diff(t) - 0.0001 * u.diff(x, 2) + 5 * u**3 - 5 * u}
    if c["eq"] == "two_out":
        x, y = syms
        u, v = sp.Function("u")(x, y), sp.Function("v")(x, y)
        return {"continuity": u.diff(x) + v.diff(y), "momentum": u * v.diff(x, 2) + u.diff(y, 2) - v}
    if c["eq"] == "heat":
        x, y, t = syms
        u = sp.Function("u")(x, y, t)
        return {"heat": u.diff(t) - 0.1 * (u.diff(x, 2) + u.diff(y, 2))}
    (x,) = syms
    u = sp.Function("u")(x)
    return {"value": u * 1}


def draw_params(named, c, seed):
    """Seeded values for every trainable tensor, by name (the same arrays go to the HIP model through set_state_dict)."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, p in named:
        shp = tuple(p.shape)
        if name.startswith("period_emb"):
            continue
        if name.endswith("alpha"):
            v = rng.uniform(0.2, 0.8, shp)
        elif name.endswith("kernel"):
            v = rng.normal(0.0, c["fourier"]["scale"], shp)
        elif name.endswith("weight_g"):
            v = np.exp(rng.normal(c["rwf"]["mean"], c["rwf"]["std"], shp))
        elif name.endswith("bias"):
            v = rng.normal(0.0, 0.1, shp)
        else:
            fin, fout = shp
            v = rng.normal(0.0, np.sqrt(2.0 / (fin + fout)), shp)
            if name.endswith("weight_v"):
                v = v * 1.2
        out[name] = v.astype(np.float32).astype(np.float64)
    return out


def main():
    import _paddle_shim as S

    mods = S.import_hotpath()
    PirateNet = mods["mlp"].PirateNet
    lambdify = mods["symbolic"].lambdify
    MSELoss = mods["mse"].MSELoss
    clear = mods["ad"].clear
    out = {}
    for ci, (name, c) in enumerate(CASES.items()):
        model = PirateNet(c["inputs"], c["outputs"], c["blocks"], c["hidden"], c["act"], periods=c["periods"],
                          fourier=c["fourier"], random_weight=c["rwf"])
        named = [(n, p) for n, p in model.named_parameters()]
        vals = draw_params(named, c, 500 + ci)
        train = []
        with torch.no_grad():
            for n, p in named:
                if n in vals:
                    p.copy_(torch.tensor(vals[n]))
                    train.append((n, p))
        rng = np.random.default_rng(5000 + ci)
        X = rng.uniform(c["lo"], c["hi"], (c["n"], len(c["inputs"]))).astype(np.float32).astype(np.float64)
        data = {k: torch.tensor(X[:, j:j + 1], requires_grad=True) for j, k in enumerate(c["inputs"])}
        eqs = equations(c)
        output_dict = model(data)
        data_dict = dict(data)
        data_dict.update(output_dict)
        for k, ex in eqs.items():
            output_dict[k] = lambdify(ex, model, fuse_derivative=True)(data_dict)
        clear()
        keys = list(eqs.keys())
        label = {k: torch.tensor(rng.standard_normal((c["n"], 1)).astype(np.float32).astype(np.float64) * 0.05) for k in keys}
        losses = MSELoss(c["reduction"])(output_dict, label, None)
        total = 0.0
        for i, k in enumerate(losses):
            total = losses[k] if i == 0 else total + losses[k]
        grads = torch.autograd.grad(total, [p for _, p in train], allow_unused=True)
        out[f"{name}/X"] = X
        out[f"{name}/total"] = np.asarray(float(total.detach()))
        for (n, p), g in zip(train, grads):
            out[f"{name}/param/{n}"] = vals[n]
            out[f"{name}/grad/{n}"] = (torch.zeros_like(p) if g is None else g).detach().numpy()
        for k in c["outputs"]:
            out[f"{name}/out/{k}"] = output_dict[k].detach().numpy()[:, 0]
        for k in keys:
            out[f"{name}/res/{k}"] = output_dict[k].detach().numpy()[:, 0]
            out[f"{name}/loss/{k}"] = np.asarray(float(losses[k].detach()))
            out[f"{name}/label/{k}"] = label[k].numpy()[:, 0]
        gn = np.sqrt(sum(float((out[f"{name}/grad/{n}"] ** 2).sum()) for n, _ in train))
        print(name, [n for n, _ in train][:6], "total loss", float(total.detach()), "|grad|", gn, flush=True)
    np.savez_compressed(os.path.join(HERE, "piratenet.npz"), **out)


if __name__ == "__main__":
    main()
