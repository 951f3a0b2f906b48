"""Lid-driven cavity, steady Navier-Stokes 2-D (BASELINE config 3) -- structure of
/root/reference/examples/ldc/ldc2d_steady_Re10.py with key=value arguments instead of hydra.  Multi-GPU:

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/ldc2d_steady.py npoint_pde=1000000
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ppsci  # noqa: E402
from examples._args import parse  # noqa: E402
from ppsci.utils import logger  # noqa: E402


def main():
    cfg = parse(dict(seed=42, output_dir="./output_ldc2d", epochs=20, iters_per_epoch=100, num_layers=5, hidden_size=128,
                     learning_rate=1e-3, npoint_pde=9801, npoint_bc=400, nu=0.01, rho=1.0, weight_pde=1e-4, log_freq=50))
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        torch.distributed.init_process_group("nccl")
    ppsci.utils.misc.set_random_seed(cfg["seed"])
    logger.init_logger("ppsci", os.path.join(cfg["output_dir"], "train.log"))
    model = ppsci.arch.MLP(("x", "y"), ("u", "v", "p"), cfg["num_layers"], cfg["hidden_size"], "tanh")
    equation = {"NavierStokes": ppsci.equation.NavierStokes(cfg["nu"], cfg["rho"], 2, False)}
    geom = {"rect": ppsci.geometry.Rectangle((-0.05, -0.05), (0.05, 0.05))}
    dl = {"dataset": "NamedArrayDataset", "iters_per_epoch": 1,
          "sampler": {"name": "BatchSampler", "drop_last": False, "shuffle": False}}
    pde = ppsci.constraint.InteriorConstraint(
        equation["NavierStokes"].equations, {"continuity": 0, "momentum_x": 0, "momentum_y": 0}, geom["rect"],
        {**dl, "batch_size": cfg["npoint_pde"]}, ppsci.loss.MSELoss("sum"), evenly=True,
        weight_dict={k: cfg["weight_pde"] for k in ("continuity", "momentum_x", "momentum_y")}, name="EQ")
    # with world_size > 1 the batch sampler hands every rank a rank-strided shard of this one global batch
    # (paddle DistributedBatchSampler semantics, /root/reference/ppsci/data/__init__.py:76-99)

    def bc(name, crit, uval):
        return ppsci.constraint.BoundaryConstraint(
            {"u": lambda out: out["u"], "v": lambda out: out["v"]}, {"u": uval, "v": 0}, geom["rect"],
            {**dl, "batch_size": cfg["npoint_bc"]}, ppsci.loss.MSELoss("sum"), criteria=crit, name=name)

    top = bc("BC_top", lambda x, y: np.isclose(y, 0.05), 1)
    walls = bc("BC_walls", lambda x, y: ~np.isclose(y, 0.05), 0)
    optimizer = ppsci.optimizer.Adam(cfg["learning_rate"])(model)
    solver = ppsci.solver.Solver(model, {c.name: c for c in (pde, top, walls)}, cfg["output_dir"], optimizer,
                                 epochs=cfg["epochs"], iters_per_epoch=cfg["iters_per_epoch"], log_freq=cfg["log_freq"],
                                 equation=equation, geom=geom)
    solver.train()


if __name__ == "__main__":
    main()
