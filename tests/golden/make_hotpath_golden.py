"""Generates tests/golden/hotpath.npz by executing the REFERENCE's own hot-path Python code
(/root/reference/ppsci/arch/mlp.py, autodiff/ad.py, utils/symbolic.py with fuse_derivative=True -- what
Solver.__init__ uses, solver.py:496-535 --, equation/pde/*.py, loss/mse.py, loss/mtl/sum.py) in this
container, with PaddlePaddle replaced by the torch-backed shim of tests/golden/_paddle_shim.py, in float64.

    python tests/golden/make_hotpath_golden.py

Per case: explicit weights (oracle.taylor_np.make_net, copied into the reference MLP), points, the residual
of every equation per point, every loss term, and d(total loss)/d(parameters) via `total_loss.backward()`-
equivalent autograd through the reference's double-backward graph."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import _paddle_shim as S  # noqa: E402

from oracle import taylor_np as T  # noqa: E402  (only make_net / flat_params: the seeded weight draw)

CASES = {
    "laplace2d_3x20": dict(eq="laplace", inputs=("x", "y"), outputs=("u",), hidden=[20, 20, 20], act="tanh", n=64,
                           lo=[0, 0], hi=[1, 1], reduction="sum"),
    "laplace2d_5x20_skip": dict(eq="laplace", inputs=("x", "y"), outputs=("u",), hidden=[20] * 5, act="tanh", n=48,
                                lo=[0, 0], hi=[1, 1], reduction="sum", skip=True),
    "allen_cahn_4x64_period": dict(eq="allen_cahn", inputs=("t", "x"), outputs=("u",), hidden=[64] * 4, act="tanh", n=96,
                                   lo=[0, -1], hi=[1, 1], reduction="mean", periods={"x": (2.0, False)}),
    "ns2d_3x20_detach": dict(eq="navier_stokes", inputs=("x", "y"), outputs=("u", "v", "p"), hidden=[20] * 3, act="tanh",
                             n=56, lo=[-0.05, -0.05], hi=[0.05, 0.05], reduction="sum", detach=("u", "v__y"), weights=True),
    "ns2d_3x32_silu": dict(eq="navier_stokes", inputs=("x", "y"), outputs=("u", "v", "p"), hidden=[32] * 3, act="silu",
                           n=40, lo=[-0.05, -0.05], hi=[0.05, 0.05], reduction="mean", weights=True),
    "poisson2d_2x24_sin": dict(eq="poisson", inputs=("x", "y"), outputs=("p",), hidden=[24, 24], act="sin", n=33,
                               lo=[0, 0], hi=[1, 1], reduction="mean"),
}


def main():
    import sympy as sp

    mods = S.import_hotpath()
    MLP = mods["mlp"].MLP
    lambdify = mods["symbolic"].lambdify
    MSELoss = mods["mse"].MSELoss
    clear = mods["ad"].clear
    out = {}
    for ci, (name, c) in enumerate(CASES.items()):
        periods = c.get("periods")
        model = MLP(c["inputs"], c["outputs"], None, tuple(c["hidden"]), c["act"], skip_connection=c.get("skip", False),
                    periods=periods)
        pidx = {c["inputs"].index(k): float(np.float32(2 * np.pi / p[0])) for k, p in (periods or {}).items()}
        net = T.make_net(len(c["inputs"]), c["hidden"], len(c["outputs"]), seed=100 + ci, activation=c["act"],
                         periods=pidx, skip_connection=c.get("skip", False), bias_scale=0.1)
        net = net.astype(np.float32).astype(np.float64)  # fp32-representable weights
        flat = T.flat_params(net)
        lin = [p for p in model.parameters() if p.dim() > 0]  # period frequencies are 0-d and non-trainable
        off = 0
        with torch.no_grad():
            for p in lin:
                k = p.numel()
                p.copy_(torch.tensor(flat[off:off + k].reshape(tuple(p.shape))))
                off += k
        assert off == flat.size
        rng = np.random.default_rng(1000 + ci)
        X = rng.uniform(c["lo"], c["hi"], (c["n"], len(c["inputs"]))).astype(np.float32).astype(np.float64)
        data = {k: torch.tensor(X[:, j:j + 1], requires_grad=True) for j, k in enumerate(c["inputs"])}
        if c["eq"] == "laplace":
            eq = mods["laplace"].Laplace(2)
        elif c["eq"] == "poisson":
            eq = mods["poisson"].Poisson(2)
        elif c["eq"] == "allen_cahn":
            eq = mods["allen_cahn"].AllenCahn(0.01)
        else:
            eq = mods["navier_stokes"].NavierStokes(0.01, 1.0, 2, False, detach_keys=c.get("detach"))
        output_dict = model(data)  # expression.py:96-102
        data_dict = dict(data)
        data_dict.update(output_dict)
        for k, ex in eq.equations.items():
            fn = lambdify(ex, model, fuse_derivative=True) if isinstance(ex, sp.Basic) else ex
            output_dict[k] = fn(data_dict)
        clear()
        keys = list(eq.equations.keys())
        label = {k: torch.tensor(rng.standard_normal((c["n"], 1)).astype(np.float32).astype(np.float64) * 0.05) for k in keys}
        weight = None
        if c.get("weights"):
            weight = {k: torch.tensor(rng.uniform(0.5, 1.5, (c["n"], 1)).astype(np.float32).astype(np.float64)) for k in keys}
        losses = MSELoss(c["reduction"])(output_dict, label, weight)
        total = 0.0
        for i, k in enumerate(losses):  # mtl/sum.py:53-60
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
            if weight:
                out[f"{name}/weight/{k}"] = weight[k].numpy()[:, 0]
        print(name, "total loss", float(total.detach()), "|grad|", float(np.linalg.norm(g)))
    np.savez_compressed(os.path.join(HERE, "hotpath.npz"), **out)


if __name__ == "__main__":
    main()
