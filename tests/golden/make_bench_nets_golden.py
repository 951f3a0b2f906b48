"""Generates tests/golden/bench_nets.npz: the EXACT networks and point clouds bench.py times (SURVEY.md 8(d) table,
BASELINE.json configs[0..2]), evaluated by the REFERENCE's own hot-path Python code (ppsci/arch/mlp.py,
autodiff/ad.py, utils/symbolic.py with fuse_derivative=True, equation/pde/*.py, loss/mse.py) in float64 with
PaddlePaddle replaced by the torch-backed shim of tests/golden/_paddle_shim.py.

    python tests/golden/make_bench_nets_golden.py

Per case: the first N_FIX points of the bench batch, the residual of every equation per point, every loss term
(over these N_FIX points) and d(total loss)/d(parameters).  The weights are NOT stored (1 MB of noise for NS 5x128):
they are the seeded draw of SURVEY 8(d) -- W ~ U(+-sqrt(6/(in+out))), b = 0, default_rng(1234), layer by layer, W then
b, rounded to fp32 -- which bench.py, the tests and this script all regenerate with `bench_weights()` below; a
checksum of the flat fp32 parameter vector is stored to catch drift.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

N_FIX = 2048

CASES = {
    # BASELINE configs[0]: Laplace2D 3x20, 10 k points rng(42).random, u_xx + u_yy, label 0, MSE-sum
    "laplace2d_3x20": dict(eq="laplace", inputs=("x", "y"), outputs=("u",), hidden=[20] * 3, reduction="sum",
                           points=lambda n: np.random.default_rng(42).random((n, 2), dtype=np.float32)),
    # BASELINE configs[1]: Allen-Cahn 4x64, 100 k points uniform([0,-1],[1,1]), eps = 0.01, label 0, MSE-mean
    "allen_cahn_4x64": dict(eq="allen_cahn", inputs=("t", "x"), outputs=("u",), hidden=[64] * 4, reduction="mean",
                            points=lambda n: np.random.default_rng(42).uniform([0, -1], [1, 1], (n, 2)).astype(np.float32)),
    # BASELINE configs[2]: LDC NavierStokes 5x128, 1 M points uniform(-0.05, 0.05), nu = 0.01, rho = 1,
    # weights 1e-4, MSE-sum
    # configs[1] at the shape of the reference's OWN yaml (examples/allen_cahn/conf/allen_cahn.yaml:38-42): 4 x 256 tanh with the
    # period embedding x -> (cos(pi x), sin(pi x)) (periods: {x: [2.0, false]}, arch/mlp.py:95-114): the width-256 kernels
    "allen_cahn_4x256_period": dict(eq="allen_cahn", inputs=("t", "x"), outputs=("u",), hidden=[256] * 4, reduction="mean",
                                    periods={"x": [2.0, False]}, d_in=3,
                                    points=lambda n: np.random.default_rng(42).uniform([0, -1], [1, 1], (n, 2)).astype(np.float32)),
    "ns2d_5x128": dict(eq="navier_stokes", inputs=("x", "y"), outputs=("u", "v", "p"), hidden=[128] * 5, reduction="sum",
                       weight=1e-4,
                       points=lambda n: np.random.default_rng(42).uniform(-0.05, 0.05, (n, 2)).astype(np.float32)),
}


def bench_weights(d_in, hidden, d_out, seed=1234):
    """flat fp32 parameter vector in `parameters()` order (W0 b0 W1 b1 ... W_last b_last)."""
    rng = np.random.default_rng(seed)
    sizes = [d_in] + list(hidden) + [d_out]
    out = []
    for fi, fo in zip(sizes[:-1], sizes[1:]):
        lim = np.sqrt(6.0 / (fi + fo))
        out.append(rng.uniform(-lim, lim, size=(fi, fo)).astype(np.float32).ravel())
        out.append(np.zeros(fo, np.float32))
    return np.concatenate(out)


def main():
    import sympy as sp

    import _paddle_shim as S

    mods = S.import_hotpath()
    MLP = mods["mlp"].MLP
    lambdify = mods["symbolic"].lambdify
    MSELoss = mods["mse"].MSELoss
    clear = mods["ad"].clear
    out = {}
    for name, c in CASES.items():
        model = MLP(c["inputs"], c["outputs"], None, tuple(c["hidden"]), "tanh", periods=c.get("periods"))
        flat = bench_weights(c.get("d_in", len(c["inputs"])), c["hidden"], len(c["outputs"]))
        lin = [p for p in model.parameters() if p.dim() > 0]
        off = 0
        with torch.no_grad():
            for p in lin:
                k = p.numel()
                p.copy_(torch.tensor(flat[off:off + k].astype(np.float64).reshape(tuple(p.shape))))
                off += k
        assert off == flat.size
        X = c["points"](N_FIX).astype(np.float64)
        data = {k: torch.tensor(X[:, j:j + 1], requires_grad=True) for j, k in enumerate(c["inputs"])}
        if c["eq"] == "laplace":
            eq = mods["laplace"].Laplace(2)
        elif c["eq"] == "allen_cahn":
            eq = mods["allen_cahn"].AllenCahn(0.01)
        else:
            eq = mods["navier_stokes"].NavierStokes(0.01, 1.0, 2, False)
        output_dict = model(data)  # expression.py:96-102
        data_dict = dict(data)
        data_dict.update(output_dict)
        for k, ex in eq.equations.items():
            fn = lambdify(ex, model, fuse_derivative=True) if isinstance(ex, sp.Basic) else ex
            output_dict[k] = fn(data_dict)
        clear()
        keys = list(eq.equations.keys())
        label = {k: torch.zeros((N_FIX, 1), dtype=torch.float64) for k in keys}
        weight = None
        if c.get("weight"):
            w32 = float(np.float32(c["weight"]))
            weight = {k: torch.full((N_FIX, 1), w32, dtype=torch.float64) for k in keys}
        losses = MSELoss(c["reduction"])(output_dict, label, weight)
        total = 0.0
        for i, k in enumerate(losses):  # mtl/sum.py:53-60
            total = losses[k] if i == 0 else total + losses[k]
        grads = torch.autograd.grad(total, lin, allow_unused=True)
        g = np.concatenate([(torch.zeros_like(p) if gi is None else gi).detach().numpy().ravel() for gi, p in zip(grads, lin)])
        out[f"{name}/X"] = X.astype(np.float32)
        out[f"{name}/param_checksum"] = np.asarray([float(flat.astype(np.float64).sum()),
                                                    float(np.abs(flat.astype(np.float64)).sum())])
        out[f"{name}/grad"] = g
        out[f"{name}/total"] = np.asarray(float(total.detach()))
        for k in keys:
            out[f"{name}/res/{k}"] = output_dict[k].detach().numpy()[:, 0]
            out[f"{name}/loss/{k}"] = np.asarray(float(losses[k].detach()))
        print(name, "total loss", float(total.detach()), "|grad|", float(np.linalg.norm(g)), flush=True)
    np.savez_compressed(os.path.join(HERE, "bench_nets.npz"), **out)


if __name__ == "__main__":
    main()
