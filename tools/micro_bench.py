"""Micro-benchmarks (run on the GPU box) for the secondary configs: SPINN Helmholtz3D step on a 128^3 grid
(BASELINE config 5) and the FNO spectral convolution (config 4).  Prints one JSON line each."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ppsci  # noqa: E402


def timeit(fn, reps=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def spinn(nc=128):
    np.random.seed(111)
    model = ppsci.arch.SPINN(("x", "y", "z"), ("u",), 32, 4, 64, "tanh")
    eq = ppsci.equation.Helmholtz(3, 1.0)
    eq.model = model
    rng = np.random.default_rng(42)
    xs = [rng.uniform(-1, 1, (nc, 1)).astype(np.float32) for _ in range(3)]
    uc = rng.standard_normal((nc, nc, nc, 1)).astype(np.float32)
    data = {"x": xs[0], "y": xs[1], "z": xs[2], "uc": uc}
    lab = {"helmholtz": uc}
    pde = ppsci.constraint.SupervisedConstraint(
        {"dataset": {"name": "ContinuousNamedArrayDataset", "input": lambda: data, "label": lambda d: lab}},
        output_expr=eq.equations, loss=ppsci.loss.MSELoss("mean"), name="PDE")
    opt = ppsci.optimizer.Adam(1e-3)(model)
    solver = ppsci.solver.Solver(model, {"PDE": pde}, "/tmp/out_spinn", opt, epochs=1, iters_per_epoch=1)
    cc = solver._compiled["PDE"]
    cc.bind(data, lab)

    def step():
        solver.engine.forward_backward([cc])
        opt.step(solver.engine.grad)

    t = timeit(step)
    pts = nc**3
    print(json.dumps({"bench": "spinn_helmholtz3d_step", "nc": nc, "ms": t * 1e3, "grid_points_per_s": pts / t,
                      "algorithmic_label_GBps": pts * 4 / t / 1e9}), flush=True)


def ns(n=125_000):
    """BASELINE config 3 per-GPU shard: LDC NavierStokes 2-D steady, MLP 2 -> 128 x 5 -> 3 tanh, 125 000
    collocation points (1 M / 8 GPUs), continuity + momentum_x + momentum_y, MSE-sum with weight 1e-4, Adam."""
    torch.manual_seed(0)
    np.random.seed(0)
    model = ppsci.arch.MLP(("x", "y"), ("u", "v", "p"), 5, 128, "tanh")
    eq = ppsci.equation.NavierStokes(0.01, 1.0, 2, False)
    X = np.random.default_rng(42).uniform(-0.05, 0.05, (n, 2)).astype(np.float32)
    keys = ("continuity", "momentum_x", "momentum_y")
    cfg = {"dataset": {"name": "NamedArrayDataset", "input": {"x": X[:, :1], "y": X[:, 1:]},
                       "label": {k: np.zeros((n, 1), np.float32) for k in keys},
                       "weight": {k: np.full((n, 1), 1e-4, np.float32) for k in keys}},
           "batch_size": n, "sampler": {"name": "BatchSampler", "shuffle": False, "drop_last": True}}
    pde = ppsci.constraint.SupervisedConstraint(cfg, ppsci.loss.MSELoss("sum"), eq.equations, name="EQ")
    opt = ppsci.optimizer.Adam(1e-3)(model)
    solver = ppsci.solver.Solver(model, {"EQ": pde}, "/tmp/out_ns", opt, epochs=1, iters_per_epoch=1)
    cc = solver._compiled["EQ"]

    def step():
        solver.engine.forward_backward([cc.fused])
        opt.step(solver.engine.grad)

    t = timeit(step, reps=10, warm=3)
    P, S = 2 * 128 + 4 * 128 * 128 + 128 * 3, 5
    print(json.dumps({"bench": "ldc_navier_stokes_5x128_step", "points": n, "ms": t * 1e3, "points_per_s": n / t,
                      "matrix_TFLOPs": 6.0 * P * S * n / t / 1e12}), flush=True)


if __name__ == "__main__":
    ns()
    spinn(int(sys.argv[1]) if len(sys.argv) > 1 else 128)
