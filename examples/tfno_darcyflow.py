"""TFNO-2D on Darcy flow, after /root/reference/examples/neuraloperator/train_tfno.py (+ conf/tfno_darcyflow_pretrain.yaml).

The reference reads `darcy_train_16.npy` / `darcy_test_{16,32}.npy`; there is no network here, so when
`data_dir` does not hold them a synthetic stand-in of the same shapes is generated (smoothed random
permeability field a(x) -> a few Jacobi sweeps of -div(a grad u) = 1), enough to exercise the whole path:
positional-encoding channels, TFNO2dNet through the HIP spectral kernel, FunctionalLoss, validators.

    python examples/tfno_darcyflow.py epochs=5 resolution=16
"""
import os
import sys

import numpy as np
import torch

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
    return a[:, None], u[:, None].astype(np.float32)


def with_grid(a):
    """Positional encoding of the reference dataset (ppsci/data/dataset/darcyflow_dataset.py): x- and y-coordinate
    channels appended to the permeability."""
    n, _, h, w = a.shape
    gy, gx = np.meshgrid(np.linspace(0, 1, h, dtype=np.float32), np.linspace(0, 1, w, dtype=np.float32), indexing="ij")
    return np.concatenate([a, np.broadcast_to(gx, (n, 1, h, w)), np.broadcast_to(gy, (n, 1, h, w))], 1).astype(np.float32)


def lp_loss(output_dict, label_dict, weight_dict=None):
    """metric.LpLoss_train(d=2, p=2) of the reference example: relative L2 per sample, summed over the batch."""
    d = (output_dict["y"] - label_dict["y"]).flatten(1)
    return {"l2": (torch.linalg.norm(d, dim=1) / torch.linalg.norm(label_dict["y"].flatten(1), dim=1)).sum()}


def lp_metric(output_dict, label_dict):
    d = (output_dict["y"] - label_dict["y"]).flatten(1)
    return {"y": (torch.linalg.norm(d, dim=1) / torch.linalg.norm(label_dict["y"].flatten(1), dim=1)).mean()}


if __name__ == "__main__":
    cfg = parse(dict(seed=666, output_dir="./output_tfno", data_dir="./datasets/darcyflow", epochs=10, resolution=16,
                     n_train=256, n_test=64, batch_size=16, n_modes=16, hidden_channels=32, lifting_channels=256,
                     projection_channels=64, n_layers=4, norm="group_norm", learning_rate=5e-3, log_freq=8))
    ppsci.utils.misc.set_random_seed(cfg["seed"])
    logger.init_logger("ppsci", os.path.join(cfg["output_dir"], "train.log"))
    res = cfg["resolution"]
    f = os.path.join(cfg["data_dir"], f"darcy_train_{res}.npy")
    if os.path.exists(f):
        raw = np.load(f, allow_pickle=True).item()
        a_tr, u_tr = np.asarray(raw["x"], np.float32)[:, None], np.asarray(raw["y"], np.float32)[:, None]
        raw = np.load(os.path.join(cfg["data_dir"], f"darcy_test_{res}.npy"), allow_pickle=True).item()
        a_te, u_te = np.asarray(raw["x"], np.float32)[:, None], np.asarray(raw["y"], np.float32)[:, None]
    else:
        logger.warning(f"{f} not found: using synthetic Darcy-like fields")
        a_tr, u_tr = synthetic_darcy(cfg["n_train"], res, 1)
        a_te, u_te = synthetic_darcy(cfg["n_test"], res, 2)
    mu, sd = u_tr.mean(), u_tr.std() + 1e-8  # UnitGaussianNormalizer of the reference (encode_output)
    train_cfg = {"dataset": {"name": "NamedArrayDataset", "input": {"x": with_grid(a_tr)}, "label": {"y": (u_tr - mu) / sd}},
                 "batch_size": cfg["batch_size"], "sampler": {"name": "BatchSampler", "shuffle": True, "drop_last": True}}
    test_cfg = {"dataset": {"name": "NamedArrayDataset", "input": {"x": with_grid(a_te)}, "label": {"y": (u_te - mu) / sd}},
                "batch_size": cfg["batch_size"], "sampler": {"name": "BatchSampler", "shuffle": False, "drop_last": False}}
    model = ppsci.arch.TFNO2dNet(("x",), ("y",), cfg["n_modes"], cfg["n_modes"], cfg["hidden_channels"], 3, 1,
                                 cfg["lifting_channels"], cfg["projection_channels"], cfg["n_layers"], norm=cfg["norm"])
    sup = ppsci.constraint.SupervisedConstraint(train_cfg, ppsci.loss.FunctionalLoss(lp_loss), name="Sup")
    val = ppsci.validate.SupervisedValidator(test_cfg, ppsci.loss.FunctionalLoss(lp_loss),
                                             metric={"l2": ppsci.metric.FunctionalMetric(lp_metric)}, name="Sup_Validator")
    opt = ppsci.optimizer.Adam(cfg["learning_rate"])(model)
    solver = ppsci.solver.Solver(model, {sup.name: sup}, cfg["output_dir"], opt, epochs=cfg["epochs"],
                                 iters_per_epoch=len(sup.data_loader), log_freq=cfg["log_freq"], eval_during_train=True,
                                 eval_freq=max(1, cfg["epochs"] // 2), validator={val.name: val})
    solver.train()
    solver.eval()
