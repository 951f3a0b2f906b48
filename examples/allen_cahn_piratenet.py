"""Allen-Cahn with PirateNet -- /root/reference/examples/allen_cahn/allen_cahn_piratenet.py:56-180 +
conf/allen_cahn_piratenet.yaml:35-75 (3 blocks x 256, tanh, periodic x, Fourier features dim 256 scale 2, random weight
factorisation mean 1.0 std 0.1, time-sorted batches of 8192 + CausalMSELoss with 32 windows, GradNorm loss weights), with the
hydra config replaced by key=value arguments.  dataset/allen_cahn.mat is not shipped (no network): the initial condition is its
analytic form u(0,x) = x^2 cos(pi x) and the evaluation reports the PDE residual.

    python examples/allen_cahn_piratenet.py epochs=2 iters_per_epoch=500
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


def main():
    cfg = parse(dict(seed=42, output_dir="./output_allen_cahn_piratenet", epochs=2, iters_per_epoch=500, batch_size=8192,
                     num_blocks=3, hidden_size=256, learning_rate=1e-3, gamma=0.9, decay_steps=5000, log_freq=100,
                     n_chunks=32, tol=1.0, grad_norm_update_freq=1000, grad_norm_momentum=0.9, grad_norm=True,
                     n_x=512, n_t_eval=101))
    ppsci.utils.misc.set_random_seed(cfg["seed"])
    logger.init_logger("ppsci", os.path.join(cfg["output_dir"], "train.log"))
    model = ppsci.arch.PirateNet(("t", "x"), ("u",), cfg["num_blocks"], cfg["hidden_size"], "tanh", periods={"x": [2.0, False]},
                                 fourier={"dim": cfg["hidden_size"], "scale": 2.0}, random_weight={"mean": 1.0, "std": 0.1})
    equation = {"AllenCahn": ppsci.equation.AllenCahn(eps=0.01)}
    t0, t1, x0, x1 = 0.0, 1.0, -1.0, 1.0
    x_star = np.linspace(x0, x1, cfg["n_x"], endpoint=False, dtype=dtype)

    def gen_input_batch():  # allen_cahn_piratenet.py:83-92: the causal loss needs the batch ordered in time
        tx = np.random.uniform([t0, x0], [t1, x1], (cfg["batch_size"], 2)).astype(dtype)
        return {"t": np.sort(tx[:, 0:1], axis=0), "x": tx[:, 1:2]}

    def gen_label_batch(input_batch):
        return {"allen_cahn": np.zeros([cfg["batch_size"], 1], dtype)}

    pde = ppsci.constraint.SupervisedConstraint(
        {"dataset": {"name": "ContinuousNamedArrayDataset", "input": gen_input_batch, "label": gen_label_batch}},
        output_expr=equation["AllenCahn"].equations, loss=ppsci.loss.CausalMSELoss(cfg["n_chunks"], "mean", tol=cfg["tol"]),
        name="PDE")
    ic_input = {"t": np.full([len(x_star), 1], t0, dtype), "x": x_star.reshape([-1, 1])}
    ic_label = {"u": (x_star**2 * np.cos(np.pi * x_star)).reshape([-1, 1]).astype(dtype)}
    ic = ppsci.constraint.SupervisedConstraint(
        {"dataset": {"name": "IterableNamedArrayDataset", "input": ic_input, "label": ic_label}},
        output_expr={"u": lambda out: out["u"]}, loss=ppsci.loss.MSELoss("mean"), name="IC")
    constraint = {pde.name: pde, ic.name: ic}
    lr_scheduler = ppsci.optimizer.lr_scheduler.ExponentialDecay(
        epochs=cfg["epochs"], iters_per_epoch=cfg["iters_per_epoch"], learning_rate=cfg["learning_rate"],
        gamma=cfg["gamma"], decay_steps=cfg["decay_steps"], by_epoch=False)()
    optimizer = ppsci.optimizer.Adam(lr_scheduler)(model)
    aggregator = (mtl.GradNorm(model, len(constraint), cfg["grad_norm_update_freq"], cfg["grad_norm_momentum"])
                  if cfg["grad_norm"] else None)  # allen_cahn_piratenet.py:165-170
    solver = ppsci.solver.Solver(model, constraint, cfg["output_dir"], optimizer, lr_scheduler, cfg["epochs"],
                                 cfg["iters_per_epoch"], log_freq=cfg["log_freq"], equation=equation, loss_aggregator=aggregator)
    solver.train()
    tx = ppsci.utils.misc.cartesian_product(np.linspace(t0, t1, cfg["n_t_eval"], dtype=dtype), x_star)
    res = solver.predict({"t": tx[:, 0:1], "x": tx[:, 1:2]}, equation["AllenCahn"].equations, batch_size=None,
                         return_numpy=True)
    u0 = solver.predict(ic_input, batch_size=None, return_numpy=True)["u"]
    logger.info(f"PDE residual RMS on a {cfg['n_t_eval']}x{cfg['n_x']} grid: {float(np.sqrt(np.mean(res['allen_cahn'] ** 2))):.5e}; "
                f"initial condition rel-L2: {float(np.linalg.norm(u0 - ic_label['u']) / np.linalg.norm(ic_label['u'])):.5e}")


if __name__ == "__main__":
    main()
