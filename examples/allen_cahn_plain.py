"""Allen-Cahn (BASELINE config 2) -- /root/reference/examples/allen_cahn/allen_cahn_plain.py:62-170 with the
hydra config replaced by key=value arguments.  The reference reads the initial condition and the reference
solution from dataset/allen_cahn.mat (not shipped, no network); here the initial condition is its analytic
form u(0,x) = x^2 cos(pi x) and evaluation reports the PDE residual.

    python examples/allen_cahn_plain.py epochs=5 iters_per_epoch=200 batch_size=4096

`causal=True fourier=True rwf=True` gives the configuration of allen_cahn_causal.py with
conf/allen_cahn_causal_fourier_rwf.yaml:35-69 (time-sorted batches + CausalMSELoss with 32 windows,
FourierEmbedding dim 256 scale 1.0, RandomWeightFactorization mean 0.5 std 0.1).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ppsci  # noqa: E402
from examples._args import parse  # noqa: E402
from ppsci.utils import logger  # noqa: E402

dtype = "float32"


def main():
    cfg = parse(dict(n_x=512, n_t_eval=101, seed=42, output_dir="./output_allen_cahn", epochs=5, iters_per_epoch=200, batch_size=4096,
                     num_layers=4, hidden_size=256, learning_rate=1e-3, gamma=0.9, decay_steps=2000, log_freq=100,  # allen_cahn.yaml:38-42
                     period_x=True, causal=False, n_chunks=32, tol=1.0, fourier=False, rwf=False))
    ppsci.utils.misc.set_random_seed(cfg["seed"])
    logger.init_logger("ppsci", os.path.join(cfg["output_dir"], "train.log"))
    periods = {"x": [2.0, False]} if cfg["period_x"] else None
    model = ppsci.arch.MLP(("t", "x"), ("u",), cfg["num_layers"], cfg["hidden_size"], "tanh", periods=periods,
                           fourier={"dim": cfg["hidden_size"], "scale": 1.0} if cfg["fourier"] else None,
                           random_weight={"mean": 0.5, "std": 0.1} if cfg["rwf"] else None)
    equation = {"AllenCahn": ppsci.equation.AllenCahn(eps=0.01)}
    t0, t1, x0, x1 = 0.0, 1.0, -1.0, 1.0
    x_star = np.linspace(x0, x1, cfg["n_x"], endpoint=False, dtype=dtype)

    def gen_input_batch():
        tx = np.random.uniform([t0, x0], [t1, x1], (cfg["batch_size"], 2)).astype(dtype)
        # allen_cahn_causal.py:88-91: the causal loss needs the batch ordered in time
        return {"t": np.sort(tx[:, 0:1], axis=0) if cfg["causal"] else tx[:, 0:1], "x": tx[:, 1:2]}

    def gen_label_batch(input_batch):
        return {"allen_cahn": np.zeros([cfg["batch_size"], 1], dtype)}

    pde = ppsci.constraint.SupervisedConstraint(
        {"dataset": {"name": "ContinuousNamedArrayDataset", "input": gen_input_batch, "label": gen_label_batch}},
        output_expr=equation["AllenCahn"].equations,
        loss=(ppsci.loss.CausalMSELoss(cfg["n_chunks"], "mean", tol=cfg["tol"]) if cfg["causal"]
              else ppsci.loss.MSELoss("mean")), name="PDE")
    ic_input = {"t": np.full([len(x_star), 1], t0, dtype), "x": x_star.reshape([-1, 1])}
    ic_label = {"u": (x_star**2 * np.cos(np.pi * x_star)).reshape([-1, 1]).astype(dtype)}
    ic = ppsci.constraint.SupervisedConstraint(
        {"dataset": {"name": "IterableNamedArrayDataset", "input": ic_input, "label": ic_label}},
        output_expr={"u": lambda out: out["u"]}, loss=ppsci.loss.MSELoss("mean"), name="IC")
    lr_scheduler = ppsci.optimizer.lr_scheduler.ExponentialDecay(
        epochs=cfg["epochs"], iters_per_epoch=cfg["iters_per_epoch"], learning_rate=cfg["learning_rate"],
        gamma=cfg["gamma"], decay_steps=cfg["decay_steps"], by_epoch=False)()
    optimizer = ppsci.optimizer.Adam(lr_scheduler)(model)
    solver = ppsci.solver.Solver(model, {pde.name: pde, ic.name: ic}, cfg["output_dir"], optimizer, lr_scheduler,
                                 cfg["epochs"], cfg["iters_per_epoch"], log_freq=cfg["log_freq"], equation=equation)
    solver.train()
    tx = ppsci.utils.misc.cartesian_product(np.linspace(t0, t1, cfg["n_t_eval"], dtype=dtype), x_star)
    res = solver.predict({"t": tx[:, 0:1], "x": tx[:, 1:2]}, equation["AllenCahn"].equations, batch_size=None,
                         return_numpy=True)
    logger.info(f"PDE residual RMS on a 101x512 grid: {float(np.sqrt(np.mean(res['allen_cahn'] ** 2))):.5e}")


if __name__ == "__main__":
    main()
