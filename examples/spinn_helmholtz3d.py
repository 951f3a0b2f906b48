"""SPINN Helmholtz3D (BASELINE config 5) -- /root/reference/examples/spinn/helmholtz3d.py:118-213 with key=value
arguments instead of hydra: one PDE constraint on the nc^3 tensor-product grid + six boundary faces.

    python examples/spinn_helmholtz3d.py nc=64 epochs=1 iters_per_epoch=1000
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ppsci  # noqa: E402
from examples._args import parse  # noqa: E402
from ppsci.utils import logger  # noqa: E402

dtype = "float32"


def exact_u(a, x, y, z):
    return np.sin(a[0] * np.pi * x) * np.sin(a[1] * np.pi * y) * np.sin(a[2] * np.pi * z)


def source_term(a, x, y, z, lda=1.0):
    u = exact_u(a, x, y, z)[..., None]
    return -((a[0] * np.pi) ** 2 + (a[1] * np.pi) ** 2 + (a[2] * np.pi) ** 2) * u + lda * u


def main():
    cfg = parse(dict(seed=111, output_dir="./output_spinn", epochs=1, iters_per_epoch=1000, nc=64, nc_test=100, r=32,
                     num_layers=4, hidden_size=64, learning_rate=1e-3, gamma=0.9, decay_steps=1000, log_freq=100,
                     a1=4, a2=4, a3=3, resample_every=100))
    ppsci.utils.misc.set_random_seed(cfg["seed"])
    logger.init_logger("ppsci", os.path.join(cfg["output_dir"], "train.log"))
    a = (cfg["a1"], cfg["a2"], cfg["a3"])
    model = ppsci.arch.SPINN(("x", "y", "z"), ("u",), cfg["r"], cfg["num_layers"], cfg["hidden_size"], "tanh")
    equation = {"Helmholtz": ppsci.equation.Helmholtz(3, 1.0)}
    equation["Helmholtz"].model = model
    state = {"iter": 0}

    def gen():
        nc = cfg["nc"]
        xc, yc, zc = (np.random.uniform(-1.0, 1.0, [nc, 1]).astype(dtype) for _ in range(3))
        xm, ym, zm = np.meshgrid(xc, yc, zc, indexing="ij")
        state.update(xc=xc, yc=yc, zc=zc, uc=source_term(a, xm, ym, zm).astype(dtype))
        one, mone = np.asarray([[1.0]], dtype), np.asarray([[-1.0]], dtype)
        state["faces"] = [(one, yc, zc), (mone, yc, zc), (xc, one, zc), (xc, mone, zc), (xc, yc, one), (xc, yc, mone)]

    gen()

    def interior():
        state["iter"] += 1
        if state["iter"] % cfg["resample_every"] == 0:
            gen()
        return {"x": state["xc"], "y": state["yc"], "z": state["zc"], "uc": state["uc"]}

    constraint = {"PDE": ppsci.constraint.SupervisedConstraint(
        {"dataset": {"name": "ContinuousNamedArrayDataset", "input": interior, "label": lambda d: {"helmholtz": d["uc"]}},
         "shard_in_engine": True},  # multi-GPU: every rank draws the same grid, the engine keeps its x-slab
        output_expr=equation["Helmholtz"].equations, loss=ppsci.loss.MSELoss("mean"), name="PDE")}
    for i in range(6):
        constraint[f"BC{i}"] = ppsci.constraint.SupervisedConstraint(
            {"dataset": {"name": "ContinuousNamedArrayDataset",
                         "input": (lambda i=i: dict(zip(("x", "y", "z"), state["faces"][i]))),
                         "label": lambda d: {"u": np.zeros([len(d["x"]), len(d["y"]), len(d["z"]), 1], dtype)}},
             "shard_in_engine": True},
            output_expr={"u": lambda out: out["u"]}, loss=ppsci.loss.MSELoss("mean"), name=f"BC{i}")
    sched = ppsci.optimizer.lr_scheduler.ExponentialDecay(cfg["epochs"], cfg["iters_per_epoch"], cfg["learning_rate"],
                                                          cfg["gamma"], cfg["decay_steps"])()
    optimizer = ppsci.optimizer.Adam(sched)(model)
    solver = ppsci.solver.Solver(model, constraint, cfg["output_dir"], optimizer, sched, cfg["epochs"], cfg["iters_per_epoch"],
                                 log_freq=cfg["log_freq"], equation=equation)
    solver.train()
    t = np.linspace(-1.0, 1.0, cfg["nc_test"], dtype=dtype)
    xm, ym, zm = np.meshgrid(t, t, t, indexing="ij")
    u_gt = exact_u(a, xm, ym, zm).reshape(-1)
    u = solver.predict({"x": t.reshape(-1, 1), "y": t.reshape(-1, 1), "z": t.reshape(-1, 1)}, batch_size=None,
                       return_numpy=True)["u"].reshape(-1)
    logger.message(f"l2_err = {np.linalg.norm(u - u_gt) / np.linalg.norm(u_gt):.4f}, rmse = {np.sqrt(np.mean((u - u_gt) ** 2)):.4f}")


if __name__ == "__main__":
    main()
