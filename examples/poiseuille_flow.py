"""LabelFree-DNN-Surrogate pipe flow -- /root/reference/examples/pipe/poiseuille_flow.py:36-158 with the hydra config
(conf/poiseuille_flow.yaml) replaced by key=value arguments.  Everything it needs runs on the fused HIP path:
three `swish` MLPs (trainable beta per layer) in a ModelList, a registered input transform (sin / cos features of x;
multiplied by X_IN = 0 as in the reference), output transforms that hard-wire the wall and pressure conditions,
NavierStokes with the viscosity as an input variable, an InteriorConstraint on a PointCloud of the (x, y, nu) grid.
The data is the script's own meshgrid, so the run is reproducible here; the end reports the error of u against the
analytic Poiseuille profile (R^2 - y^2) dP / (2 L nu rho).

    python examples/poiseuille_flow.py epochs=300
"""
import copy
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ppsci  # noqa: E402
import ppsci.functional as F  # noqa: E402
from examples._args import parse  # noqa: E402
from ppsci.utils import logger  # noqa: E402

dtype = "float32"


def main():
    cfg = parse(dict(seed=42, output_dir="./output_poiseuille_flow", epochs=300, batch_size=128, learning_rate=5e-3,
                     log_freq=195, NU_MEAN=0.001, NU_STD=0.9, L=1.0, R=0.05, RHO=1.0, P_OUT=0.0, P_IN=0.1, N_x=10, N_y=50,
                     N_p=50, X_IN=0.0))
    ppsci.utils.misc.set_random_seed(cfg["seed"])
    logger.init_logger("ppsci", os.path.join(cfg["output_dir"], "train.log"))
    X_IN, L, R = cfg["X_IN"], cfg["L"], cfg["R"]
    X_OUT = X_IN + L
    nu0, nu1 = cfg["NU_MEAN"] * (1 - cfg["NU_STD"]), cfg["NU_MEAN"] * (1 + cfg["NU_STD"])
    xs = np.linspace(X_IN, X_OUT, cfg["N_x"], endpoint=True, dtype=dtype)
    ys = np.linspace(-R, R, cfg["N_y"], endpoint=True, dtype=dtype)
    nus = np.linspace(nu0, nu1, cfg["N_p"], endpoint=True, dtype=dtype)
    grid = np.array(np.meshgrid(xs, ys, nus)).reshape(3, -1).T
    pts = copy.deepcopy(grid)
    np.random.shuffle(pts)
    geom = ppsci.geometry.PointCloud({"x": pts[:, 0:1], "y": pts[:, 1:2], "nu": pts[:, 2:3]}, ("x", "y", "nu"))

    keys = ("sin(x)", "cos(x)", "y", "nu")
    model_u = ppsci.arch.MLP(keys, ("u",), 3, 50, "swish")
    model_v = ppsci.arch.MLP(keys, ("v",), 3, 50, "swish")
    model_p = ppsci.arch.MLP(keys, ("p",), 3, 50, "swish")
    b = 2 * np.pi / (X_OUT - X_IN)
    c = np.pi * (X_IN + X_OUT) / (X_IN - X_OUT)

    def input_trans(d):
        return {"sin(x)": X_IN * F.sin(b * d["x"] + c), "cos(x)": X_IN * F.cos(b * d["x"] + c), "y": d["y"], "nu": d["nu"]}

    model_u.register_output_transform(lambda d, out: {"u": out["u"] * (R**2 - d["y"] ** 2)})
    model_v.register_output_transform(lambda d, out: {"v": (R**2 - d["y"] ** 2) * out["v"]})
    model_p.register_output_transform(lambda d, out: {
        "p": (cfg["P_IN"] - cfg["P_OUT"]) * (X_OUT - d["x"]) / L + (X_IN - d["x"]) * (X_OUT - d["x"]) * out["p"]})
    for m in (model_u, model_v, model_p):
        m.register_input_transform(input_trans)
    model = ppsci.arch.ModelList((model_u, model_v, model_p))
    optimizer = ppsci.optimizer.Adam(cfg["learning_rate"])(model)
    equation = {"NavierStokes": ppsci.equation.NavierStokes(nu="nu", rho=cfg["RHO"], dim=2, time=False)}
    iters = int(len(pts) / cfg["batch_size"])
    pde = ppsci.constraint.InteriorConstraint(
        equation["NavierStokes"].equations, {"continuity": 0, "momentum_x": 0, "momentum_y": 0}, geom=geom,
        dataloader_cfg={"dataset": "NamedArrayDataset", "num_workers": 1, "batch_size": cfg["batch_size"],
                        "iters_per_epoch": iters, "sampler": {"name": "BatchSampler", "shuffle": False, "drop_last": False}},
        loss=ppsci.loss.MSELoss("mean"), evenly=True, name="EQ")
    solver = ppsci.solver.Solver(model, {pde.name: pde}, cfg["output_dir"], optimizer, epochs=cfg["epochs"],
                                 iters_per_epoch=iters, eval_during_train=False, save_freq=0, log_freq=cfg["log_freq"],
                                 equation=equation)
    solver.train()
    pred = solver.predict({"x": grid[:, 0:1], "y": grid[:, 1:2], "nu": grid[:, 2:3]}, batch_size=None, return_numpy=True)
    dP = cfg["P_IN"] - cfg["P_OUT"]
    u_exact = (R**2 - grid[:, 1:2] ** 2) * dP / (2 * L * grid[:, 2:3] * cfg["RHO"])
    err = np.linalg.norm(pred["u"] - u_exact) / np.linalg.norm(u_exact)
    logger.message(f"relative L2 error of u against the Poiseuille profile: {err:.4f}")


if __name__ == "__main__":
    main()
