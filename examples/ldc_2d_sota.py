"""Lid-driven cavity with the curriculum / ModifiedMLP configuration of /root/reference/examples/ldc/ldc_2d_Re3200_sota.py
(+ conf/ldc_2d_Re3200_sota.yaml): ModifiedMLP (x, y) -> (u, v, p), 5 x 256 tanh, Fourier features dim 128 scale 10, random weight
factorisation mean 1.0 std 0.1; per Reynolds number of the curriculum [100, 400, 1000, 3200]: NavierStokes(1/Re) residuals on
8 192 fresh uniform points per iteration + 4 x 256 boundary points (lid u = 1), GradNorm over the five loss terms, ONE optimizer
and ExponentialDecay schedule across the whole curriculum.  The reference validates against ./data/ldc_Re*.mat (not shipped, no
network): here the evaluation reports the PDE residual and the lid boundary error instead.

    python examples/ldc_2d_sota.py epochs=2,2,2,2 iters_per_epoch=200
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ppsci  # noqa: E402
from examples._args import parse  # noqa: E402
from ppsci.loss import mtl  # noqa: E402
from ppsci.utils import logger  # noqa: E402

dtype = "float32"


def sample_points_on_square_boundary(num_pts_per_side, eps):  # ldc_2d_Re3200_sota.py:55-73
    lin, cut = np.linspace(0, 1, num_pts_per_side), np.linspace(0, 1 - eps, num_pts_per_side)
    top = np.column_stack((lin, np.ones_like(lin)))
    bottom = np.column_stack((lin, np.zeros_like(lin)))
    left = np.column_stack((np.zeros_like(cut), cut))
    right = np.column_stack((np.ones_like(cut), cut))
    return np.vstack((top, bottom, left, right))


if __name__ == "__main__":
    cfg = parse(dict(n_eval=101, seed=42, output_dir="./output_ldc_2d_sota", Re="100,400,1000,3200", epochs="2,2,2,2", iters_per_epoch=200,
                     num_layers=5, hidden_size=256, fourier_dim=128, fourier_scale=10.0, batch_pde=8192, batch_bc=256,
                     learning_rate=1e-3, gamma=0.9, decay_steps=10000, grad_norm_update_freq=1000, grad_norm_momentum=0.9,
                     log_freq=100))
    as_list = lambda v, typ: [typ(x) for x in (v if isinstance(v, (tuple, list)) else str(v).split(","))]  # noqa: E731
    Re_list, epochs = as_list(cfg["Re"], float), as_list(cfg["epochs"], int)
    ppsci.utils.misc.set_random_seed(cfg["seed"])
    logger.init_logger("ppsci", os.path.join(cfg["output_dir"], "train.log"))
    model = ppsci.arch.ModifiedMLP(("x", "y"), ("u", "v", "p"), cfg["num_layers"], cfg["hidden_size"], "tanh",
                                   fourier={"dim": cfg["fourier_dim"], "scale": cfg["fourier_scale"]},
                                   random_weight={"mean": 1.0, "std": 0.1})
    lr_scheduler = ppsci.optimizer.lr_scheduler.ExponentialDecay(
        epochs=sum(epochs), iters_per_epoch=cfg["iters_per_epoch"], learning_rate=cfg["learning_rate"], gamma=cfg["gamma"],
        decay_steps=cfg["decay_steps"], by_epoch=False)()
    optimizer = ppsci.optimizer.Adam(lr_scheduler)(model)
    grad_norm = mtl.GradNorm(model, 5, update_freq=cfg["grad_norm_update_freq"], momentum=cfg["grad_norm_momentum"])
    x_bc = sample_points_on_square_boundary(cfg["batch_bc"], eps=0.01).astype(dtype)
    v_bc = np.zeros((cfg["batch_bc"] * 4, 1), dtype)
    u_bc = v_bc.copy()
    u_bc[: cfg["batch_bc"]] = 1.0

    for idx, (Re, ep) in enumerate(zip(Re_list, epochs)):  # train_curriculum, ldc_2d_Re3200_sota.py:75-200
        logger.message(f"Training curriculum {idx + 1}/{len(epochs)} Re={Re:.5g} epochs={ep}")
        equation = {"NavierStokes": ppsci.equation.NavierStokes(1 / Re, 1, dim=2, time=False)}

        def gen_input_batch():
            tx = np.random.uniform([0.0, 0.0], [1.0, 1.0], (cfg["batch_pde"], 2)).astype(dtype)
            return {"x": tx[:, 0:1], "y": tx[:, 1:2]}

        def gen_label_batch(input_batch):
            z = np.zeros([cfg["batch_pde"], 1], dtype)
            return {"continuity": z, "momentum_x": z, "momentum_y": z}

        pde = ppsci.constraint.SupervisedConstraint(
            {"dataset": {"name": "ContinuousNamedArrayDataset", "input": gen_input_batch, "label": gen_label_batch}},
            output_expr=equation["NavierStokes"].equations, loss=ppsci.loss.MSELoss("mean"), name="PDE")
        bc = ppsci.constraint.SupervisedConstraint(
            {"dataset": {"name": "IterableNamedArrayDataset", "input": {"x": x_bc[:, 0:1], "y": x_bc[:, 1:2]},
                         "label": {"u": u_bc, "v": v_bc}}},
            output_expr={"u": lambda out: out["u"], "v": lambda out: out["v"]}, loss=ppsci.loss.MSELoss("mean"), name="BC")
        solver = ppsci.solver.Solver(model, {pde.name: pde, bc.name: bc}, os.path.join(cfg["output_dir"], f"Re_{int(Re)}"),
                                     optimizer, lr_scheduler, ep, cfg["iters_per_epoch"], log_freq=cfg["log_freq"],
                                     equation=equation, loss_aggregator=grad_norm)
        solver.train()
        g = np.linspace(0.0, 1.0, cfg["n_eval"], dtype=dtype)
        xy = ppsci.utils.misc.cartesian_product(g, g)
        res = solver.predict({"x": xy[:, 0:1], "y": xy[:, 1:2]}, equation["NavierStokes"].equations, batch_size=None,
                             return_numpy=True)
        lid = solver.predict({"x": x_bc[: cfg["batch_bc"], 0:1], "y": x_bc[: cfg["batch_bc"], 1:2]}, batch_size=None,
                             return_numpy=True)
        logger.info(f"Re={Re:.5g}: residual RMS continuity {np.sqrt(np.mean(res['continuity'] ** 2)):.3e}, momentum_x "
                    f"{np.sqrt(np.mean(res['momentum_x'] ** 2)):.3e}; lid |u - 1| mean {np.abs(lid['u'] - 1).mean():.3e}")
