"""Euler beam, after /root/reference/examples/euler_beam/euler_beam.py (+ conf/euler_beam.yaml): u_xxxx + 1 = 0 on (0, 1),
clamped at x = 0 (u = u_x = 0), free at x = 1 (u_xx = u_xxx = 0); exact solution -x^4/24 + x^3/6 - x^2/4.

The PDE constraint (Biharmonic(dim=1), 100 Hammersley points) runs on the fused kernels with fourth-order derivative streams;
the boundary constraint of the reference picks ROWS of its four-point batch (`d["u"][0:1]`, `jacobian(...)[1:2]`, ...): one-row
slices as whole outputs are lowered too (a per-point weight mask, paddlescience_amd/compile.py), so both constraints are fused.

    python examples/euler_beam.py epochs=2000
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ppsci  # noqa: E402
from examples._args import parse  # noqa: E402
from ppsci.autodiff import hessian, jacobian  # noqa: E402
from ppsci.utils import logger  # noqa: E402

DEFAULTS = dict(seed=42, output_dir="./output_euler_beam", epochs=10000, iters_per_epoch=1, q=-1.0, D=1.0, num_layers=3,
                hidden_size=20, learning_rate=1e-3, batch_pde=100, batch_bc=4, eval_total=100, log_freq=500)


def build(cfg):
    """model, constraints, validator and Solver of the case (also used by bench.py's `extra` entry)."""
    ppsci.utils.misc.set_random_seed(cfg["seed"])
    geom = {"interval": ppsci.geometry.Interval(0, 1)}
    model = ppsci.arch.MLP(("x",), ("u",), cfg["num_layers"], cfg["hidden_size"])
    equation = {"biharmonic": ppsci.equation.Biharmonic(dim=1, q=cfg["q"], D=cfg["D"])}
    dl = {"dataset": "IterableNamedArrayDataset", "iters_per_epoch": cfg["iters_per_epoch"]}
    pde = ppsci.constraint.InteriorConstraint(equation["biharmonic"].equations, {"biharmonic": 0}, geom["interval"],
                                              {**dl, "batch_size": cfg["batch_pde"]}, ppsci.loss.MSELoss(), random="Hammersley",
                                              name="EQ")
    bc = ppsci.constraint.BoundaryConstraint(
        {"u0": lambda d: d["u"][0:1], "u__x": lambda d: jacobian(d["u"], d["x"])[1:2],
         "u__x__x": lambda d: hessian(d["u"], d["x"])[2:3],
         "u__x__x__x": lambda d: jacobian(hessian(d["u"], d["x"]), d["x"])[3:4]},
        {"u0": 0, "u__x": 0, "u__x__x": 0, "u__x__x__x": 0}, geom["interval"], {**dl, "batch_size": cfg["batch_bc"]},
        ppsci.loss.MSELoss("sum"), evenly=True, name="BC")

    def u_solution(out):
        x = out["x"]
        return -(x ** 4) / 24 + x ** 3 / 6 - x ** 2 / 4

    val = ppsci.validate.GeometryValidator({"u": lambda out: out["u"]}, {"u": u_solution}, geom["interval"],
                                           {"dataset": "IterableNamedArrayDataset", "total_size": cfg["eval_total"]},
                                           ppsci.loss.MSELoss(), evenly=True, metric={"MSE": ppsci.metric.MSE()}, name="L2Rel_Validator")
    opt = ppsci.optimizer.Adam(cfg["learning_rate"])(model)
    # visualizer (euler_beam.py:88-100 of the reference): exact and predicted deflection over x as a scatter plot
    visu_points = geom["interval"].sample_interior(cfg["eval_total"], evenly=True)
    visualizer = {"visualize_u": ppsci.visualize.VisualizerScatter1D(
        visu_points, ("x",), {"u_label": lambda d: u_solution(d), "u_pred": lambda d: d["u"]}, num_timestamps=1, prefix="result_u")}
    solver = ppsci.solver.Solver(model, {pde.name: pde, bc.name: bc}, cfg["output_dir"], opt, epochs=cfg["epochs"],
                                 iters_per_epoch=cfg["iters_per_epoch"], log_freq=cfg["log_freq"], equation=equation, geom=geom,
                                 validator={val.name: val}, visualizer=visualizer)
    return solver


if __name__ == "__main__":
    cfg = parse(dict(DEFAULTS))
    logger.init_logger("ppsci", os.path.join(cfg["output_dir"], "train.log"))
    solver = build(cfg)
    solver.train()
    solver.eval()
    solver.visualize()
