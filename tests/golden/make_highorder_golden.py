"""Generates tests/golden/highorder.npz by executing the REFERENCE's own hot-path Python (ppsci/arch/mlp.py with
arch/activation.py's act_func_dict, autodiff/ad.py, utils/symbolic.py -- DerivativeNode of any order, :310-333 --,
equation/pde/biharmonic.py, loss/mse.py) in float64 under the torch-backed paddle shim (tests/golden/_paddle_shim.py):

  * third / fourth-order residuals: Biharmonic(dim=1) (examples/euler_beam), Biharmonic(dim=2) incl. the mixed u_xxyy
    (examples/biharmonic2d), a KdV-type residual u_t + u u_x + 0.0025 u_xxx;
  * the activations of act_func_dict added after round 1: relu, leaky_relu, elu, selu, identity.

    python tests/golden/make_highorder_golden.py
"""
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import taylor_np as T  # noqa: E402  (only make_net / flat_params: the seeded weight draw)

CASES = {
    "euler_beam_3x20": dict(eq="biharmonic1", inputs=("x",), outputs=("u",), hidden=[20] * 3, act="tanh", n=40,
                            lo=[0.0], hi=[1.0], reduction="mean"),
    "biharmonic2d_5x20": dict(eq="biharmonic2", inputs=("x", "y"), outputs=("u",), hidden=[20] * 5, act="tanh", n=24,
                              lo=[0, 0], hi=[1, 1], reduction="mean"),
    "kdv_3x32_silu": dict(eq="kdv", inputs=("t", "x"), outputs=("u",), hidden=[32] * 3, act="silu", n=36,
                          lo=[0, -1], hi=[1, 1], reduction="sum"),
    "beam_bc_2x24_sin": dict(eq="beam_bc", inputs=("x",), outputs=("u",), hidden=[24] * 2, act="sin", n=20,
                             lo=[0.0], hi=[1.0], reduction="sum"),
    "act_relu": dict(eq="first_second", inputs=("x", "y"), outputs=("u",), hidden=[20] * 2, act="relu", n=33,
                     lo=[-1, -1], hi=[1, 1], reduction="mean"),
    "act_leaky_relu": dict(eq="first_second", inputs=("x", "y"), outputs=("u",), hidden=[20] * 2, act="leaky_relu", n=33,
                           lo=[-1, -1], hi=[1, 1], reduction="mean"),
    "act_elu": dict(eq="first_second", inputs=("x", "y"), outputs=("u",), hidden=[20] * 3, act="elu", n=33,
                    lo=[-1, -1], hi=[1, 1], reduction="mean"),
    "act_selu": dict(eq="first_second", inputs=("x", "y"), outputs=("u",), hidden=[20] * 2, act="selu", n=33,
                     lo=[-1, -1], hi=[1, 1], reduction="mean"),
    "act_identity": dict(eq="first_second", inputs=("x", "y"), outputs=("u",), hidden=[20] * 2, act="identity", n=33,
                         lo=[-1, -1], hi=[1, 1], reduction="mean"),
}


def equations(mods, c):
    """name -> sympy expression (the objects the reference's lambdify consumes)."""
    import sympy as sp

    if c["eq"] in ("biharmonic1", "biharmonic2"):
        return mods["biharmonic"].Biharmonic(int(c["eq"][-1]), -1.0, 1.0).equations
    syms = sp.symbols(" ".join(c["inputs"]))
    syms = syms if isinstance(syms, tuple) else (syms,)
    u = sp.Function("u")(*syms)
    if c["eq"] == "kdv":
        t, x = syms
        return {"kdv": u.diff(t) + u * u.diff(x) + 0.0025 * u.diff(x, 3)}
    if c["eq"] == "beam_bc":  # the four boundary quantities of euler_beam.py:49-54 as per-point expressions
        (x,) = syms
        return {"u__x": u.diff(x), "u__x__x": u.diff(x, 2), "u__x__x__x": u.diff(x, 3)}
    x, y = syms
    return {"r": u.diff(x) + u * u.diff(y) + 0.5 * u.diff(x, 2) + u.diff(y, 2)}


def main():
    import sympy as sp

    import _paddle_shim as S

    mods = S.import_hotpath()
    mods["biharmonic"] = importlib.import_module("ppsci.equation.pde.biharmonic")
    MLP = mods["mlp"].MLP
    lambdify = mods["symbolic"].lambdify
    MSELoss = mods["mse"].MSELoss
    clear = mods["ad"].clear
    out = {}
    for ci, (name, c) in enumerate(CASES.items()):
        model = MLP(c["inputs"], c["outputs"], None, tuple(c["hidden"]), c["act"])
        net = T.make_net(len(c["inputs"]), c["hidden"], len(c["outputs"]), seed=300 + ci, activation=c["act"], bias_scale=0.1)
        net = net.astype(np.float32).astype(np.float64)
        flat = T.flat_params(net)
        lin = [p for p in model.parameters() if p.dim() > 0]
        off = 0
        with torch.no_grad():
            for p in lin:
                k = p.numel()
                p.copy_(torch.tensor(flat[off:off + k].reshape(tuple(p.shape))))
                off += k
        assert off == flat.size
        rng = np.random.default_rng(3000 + ci)
        X = rng.uniform(c["lo"], c["hi"], (c["n"], len(c["inputs"]))).astype(np.float32).astype(np.float64)
        data = {k: torch.tensor(X[:, j:j + 1], requires_grad=True) for j, k in enumerate(c["inputs"])}
        eqs = equations(mods, c)
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
        grads = torch.autograd.grad(total, lin, allow_unused=True)
        g = np.concatenate([(torch.zeros_like(p) if gi is None else gi).detach().numpy().ravel() for gi, p in zip(grads, lin)])
        out[f"{name}/X"] = X
        out[f"{name}/params"] = flat
        out[f"{name}/grad"] = g
        out[f"{name}/total"] = np.asarray(float(total.detach()))
        for k in keys:
            out[f"{name}/res/{k}"] = output_dict[k].detach().numpy()[:, 0]
            out[f"{name}/loss/{k}"] = np.asarray(float(losses[k].detach()))
            out[f"{name}/label/{k}"] = label[k].numpy()[:, 0]
        print(name, "total loss", float(total.detach()), "|grad|", float(np.linalg.norm(g)), flush=True)
    np.savez_compressed(os.path.join(HERE, "highorder.npz"), **out)


if __name__ == "__main__":
    main()
