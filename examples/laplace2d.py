"""Laplace2D (BASELINE config 1) -- the training script of /root/reference/examples/laplace/laplace2d.py:
20-126 with the hydra config (examples/laplace/conf/laplace2d.yaml) replaced by key=value arguments.

    python examples/laplace2d.py epochs=2000 num_layers=5 hidden_size=20
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ppsci  # noqa: E402
from examples._args import parse  # noqa: E402
from ppsci.utils import logger  # noqa: E402


def main():
    cfg = parse(dict(seed=42, output_dir="./output_laplace2d", epochs=2000, iters_per_epoch=1, num_layers=5,
                     hidden_size=20, learning_rate=1e-3, npoint_interior=9801, npoint_bc=400, eval_freq=200,
                     log_freq=200))
    ppsci.utils.misc.set_random_seed(cfg["seed"])
    logger.init_logger("ppsci", os.path.join(cfg["output_dir"], "train.log"))
    model = ppsci.arch.MLP(("x", "y"), ("u",), cfg["num_layers"], cfg["hidden_size"])
    equation = {"laplace": ppsci.equation.Laplace(dim=2)}
    geom = {"rect": ppsci.geometry.Rectangle((0.0, 0.0), (1.0, 1.0))}

    def u_solution_func(out):
        x, y = out["x"], out["y"]
        return np.cos(x) * np.cosh(y)

    dl = {"dataset": "IterableNamedArrayDataset", "iters_per_epoch": cfg["iters_per_epoch"]}
    n_total = cfg["npoint_interior"] + cfg["npoint_bc"]
    pde = ppsci.constraint.InteriorConstraint(equation["laplace"].equations, {"laplace": 0}, geom["rect"],
                                              {**dl, "batch_size": n_total}, ppsci.loss.MSELoss("sum"), evenly=True,
                                              name="EQ")
    bc = ppsci.constraint.BoundaryConstraint({"u": lambda out: out["u"]}, {"u": u_solution_func}, geom["rect"],
                                             {**dl, "batch_size": cfg["npoint_bc"]}, ppsci.loss.MSELoss("sum"), name="BC")
    optimizer = ppsci.optimizer.Adam(learning_rate=cfg["learning_rate"])(model)
    mse_metric = ppsci.validate.GeometryValidator({"u": lambda out: out["u"]}, {"u": u_solution_func}, geom["rect"],
                                                  {"dataset": "IterableNamedArrayDataset", "total_size": n_total},
                                                  ppsci.loss.MSELoss(), evenly=True, metric={"MSE": ppsci.metric.MSE()},
                                                  with_initial=True, name="MSE_Metric")
    # visualizer (laplace2d.py:95-104 of the reference): the prediction on the evenly sampled points as a .vtu point cloud
    vis_points = geom["rect"].sample_interior(n_total, evenly=True)
    visualizer = {"visualize_u": ppsci.visualize.VisualizerVtu(vis_points, {"u": lambda d: d["u"]}, num_timestamps=1,
                                                               prefix="result_u")}
    solver = ppsci.solver.Solver(model, {pde.name: pde, bc.name: bc}, cfg["output_dir"], optimizer, epochs=cfg["epochs"],
                                 iters_per_epoch=cfg["iters_per_epoch"], eval_during_train=True, eval_freq=cfg["eval_freq"],
                                 log_freq=cfg["log_freq"], equation=equation, geom=geom,
                                 validator={mse_metric.name: mse_metric}, visualizer=visualizer)
    solver.train()
    solver.eval()
    solver.visualize()


if __name__ == "__main__":
    main()
