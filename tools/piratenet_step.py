"""allen_cahn_piratenet.yaml's network (3 blocks x 256, periodic x, Fourier 256, RWF) on a batch of 8192 points: the
u_t / u_xx residual step (forward + reverse + Adam) in isolation, for timing and rocprofv3.

    python tools/piratenet_step.py [steps] [batch]"""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ppsci  # noqa: E402

if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    np.random.seed(1)
    model = ppsci.arch.PirateNet(("t", "x"), ("u",), 3, 256, "tanh", periods={"x": [2.0, False]},
                                 fourier={"dim": 256, "scale": 2.0}, random_weight={"mean": 1.0, "std": 0.1})
    eq = ppsci.equation.AllenCahn(eps=0.01)
    rng = np.random.default_rng(0)
    tx = rng.uniform([0, -1], [1, 1], (n, 2)).astype(np.float32)
    inp = {"t": tx[:, 0:1], "x": tx[:, 1:2]}
    cfg = {"dataset": {"name": "IterableNamedArrayDataset", "input": inp, "label": {"allen_cahn": np.zeros((n, 1), np.float32)}}}
    cst = ppsci.constraint.SupervisedConstraint(cfg, ppsci.loss.MSELoss("mean"), eq.equations, name="PDE")
    with tempfile.TemporaryDirectory() as tmp:
        opt = ppsci.optimizer.Adam(1e-3)(model)
        solver = ppsci.solver.Solver(model, {"PDE": cst}, tmp, opt, epochs=1, iters_per_epoch=1)
        fused = solver._compiled["PDE"].fused

        def step():
            solver.engine.forward_backward([fused])
            opt.step(solver.engine.grad)

        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    S = fused.streams.S
    H, nb = 256, 3
    dense = 2 + 3 * nb  # HxH layers
    flops = 3 * 2 * dense * H * H * S * n  # fwd + data gradient + weight gradient
    print(json.dumps({"ms_per_step": dt * 1e3, "points_per_s": n / dt, "streams": S, "dense_TFLOPs": flops / dt / 1e12,
                      "loss": fused.losses()}))
