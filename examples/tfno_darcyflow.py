"""TFNO-2D on Darcy flow, after /root/reference/examples/neuraloperator/train_tfno.py (+ conf/tfno_darcyflow_pretrain.yaml):
DarcyFlowDataset (positional-encoding channels, unit-Gaussian output encoding) -> TFNO2dNet -> LpLoss_train / H1Loss_train,
validated at 16x16 and 32x32 with the H1 / L2 metrics.

The reference reads `darcy_train_16.npy` / `darcy_test_{16,32}.npy`; there is no network here, so when `data_dir` does not
hold them, files of the same format with a synthetic stand-in are written first (smoothed random two-phase permeability
a(x) -> a few Jacobi sweeps of -div(a grad u) = 1): enough to exercise the whole path.

    python examples/tfno_darcyflow.py epochs=5
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ppsci  # noqa: E402
from examples._args import parse  # noqa: E402
from ppsci.utils import logger  # noqa: E402


def synthetic_darcy(n, res, seed):
    rng = np.random.default_rng(seed)
    k = np.fft.fftfreq(res)[:, None] ** 2 + np.fft.fftfreq(res)[None, :] ** 2
    filt = np.exp(-k * (res * 0.35) ** 2)
    a = np.fft.ifft2(np.fft.fft2(rng.standard_normal((n, res, res))) * filt).real
    a = np.where(a > 0, 12.0, 3.0).astype(np.float32)  # two-phase medium like the reference data
    u = np.zeros_like(a)
    h2 = (1.0 / res) ** 2
    for _ in range(200):  # Jacobi sweeps, zero Dirichlet boundary
        up = np.pad(u, ((0, 0), (1, 1), (1, 1)))
        u = (up[:, :-2, 1:-1] + up[:, 2:, 1:-1] + up[:, 1:-1, :-2] + up[:, 1:-1, 2:] + h2 / a) / 4.0
    return a, u.astype(np.float32)


def ensure_data(data_dir, n_train, n_test):
    os.makedirs(data_dir, exist_ok=True)
    for name, n, res, seed in (("darcy_train_16", n_train, 16, 1), ("darcy_test_16", n_test, 16, 2), ("darcy_test_32", n_test, 32, 3)):
        f = os.path.join(data_dir, name + ".npy")
        if not os.path.exists(f):
            logger.warning(f"{f} not found: writing a synthetic Darcy-like stand-in of the same format")
            a, u = synthetic_darcy(n, res, seed)
            np.save(f, {"x": a, "y": u}, allow_pickle=True)


if __name__ == "__main__":
    cfg = parse(dict(seed=666, output_dir="./output_tfno", data_dir="./datasets/darcyflow", epochs=10, n_train=256, n_test=64,
                     batch_size=16, n_modes=16, hidden_channels=32, lifting_channels=256, projection_channels=64, n_layers=4,
                     norm="group_norm", learning_rate=5e-3, log_freq=8, training_loss="h1"))
    ppsci.utils.misc.set_random_seed(cfg["seed"])
    logger.init_logger("ppsci", os.path.join(cfg["output_dir"], "train.log"))
    ensure_data(cfg["data_dir"], cfg["n_train"], cfg["n_test"])

    def loader(split, shuffle):
        return {"dataset": {"name": "DarcyFlowDataset", "data_dir": cfg["data_dir"], "input_keys": ("x",), "label_keys": ("y",),
                            "train_resolution": 16, "test_resolutions": [16, 32], "grid_boundaries": [[0, 1], [0, 1]],
                            "encode_input": False, "encode_output": True, "encoding": "channel-wise", "channel_dim": 1,
                            "data_split": split},
                "sampler": {"name": "BatchSampler", "drop_last": False, "shuffle": shuffle}, "batch_size": cfg["batch_size"]}

    train_loss = ppsci.loss.LpLoss_train(d=2, p=2) if cfg["training_loss"] == "l2" else ppsci.loss.H1Loss_train(d=2)
    sup = ppsci.constraint.SupervisedConstraint(loader("train", True), loss=ppsci.loss.FunctionalLoss(train_loss), name="Sup")
    metric = {"h1": ppsci.metric.FunctionalMetric(ppsci.loss.H1Loss(d=2)), "l2": ppsci.metric.FunctionalMetric(ppsci.loss.LpLoss(d=2, p=2))}
    validator = {
        "Sup_Validator_16x16": ppsci.validate.SupervisedValidator(loader("test_16x16", False), ppsci.loss.FunctionalLoss(train_loss),
                                                                  metric=metric, name="Sup_Validator_16x16"),
        "Sup_Validator_32x32": ppsci.validate.SupervisedValidator(loader("test_32x32", False), ppsci.loss.FunctionalLoss(train_loss),
                                                                  metric=metric, name="Sup_Validator_32x32"),
    }
    model = ppsci.arch.TFNO2dNet(("x",), ("y",), cfg["n_modes"], cfg["n_modes"], cfg["hidden_channels"], 3, 1,
                                 cfg["lifting_channels"], cfg["projection_channels"], cfg["n_layers"], norm=cfg["norm"])
    opt = ppsci.optimizer.Adam(cfg["learning_rate"])(model)
    solver = ppsci.solver.Solver(model, {sup.name: sup}, cfg["output_dir"], opt, epochs=cfg["epochs"],
                                 iters_per_epoch=len(sup.data_loader), log_freq=cfg["log_freq"], eval_during_train=True,
                                 eval_freq=max(1, cfg["epochs"] // 2), validator=validator)
    solver.train()
    solver.eval()
