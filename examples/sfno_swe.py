"""SFNO on the spherical shallow-water equations, after /root/reference/examples/neuraloperator/train_sfno.py (+
conf/sfno_swe_pretrain.yaml): SphericalSWEDataset (3 fields on a 32 x 64 latitude-longitude grid) -> SFNONet (32 degrees x 16 orders,
hidden 32, 4 layers, GroupNorm) -> LpLoss, validated on the 32 x 64 and 64 x 128 grids with the L2 metric.

The reference reads `train_SWE_32x64.npy` / `test_SWE_{32x64,64x128}.npy`; there is no network here, so when `data_dir` does not hold
them, files of the same format with a synthetic stand-in are written first (band-limited random fields advected by a solid-body
rotation: the label is the input rotated in longitude and damped) -- enough to exercise the whole path.

    python examples/sfno_swe.py epochs=5
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ppsci  # noqa: E402
from examples._args import parse  # noqa: E402
from ppsci.utils import logger  # noqa: E402


def synthetic_swe(n, nlat, nlon, seed):
    rng = np.random.default_rng(seed)
    lat = np.linspace(0, np.pi, nlat)[:, None]
    lon = np.linspace(0, 2 * np.pi, nlon, endpoint=False)[None, :]
    x = np.zeros((n, 3, nlat, nlon), np.float32)
    y = np.zeros_like(x)
    for k in range(1, 5):
        a = rng.standard_normal((n, 3, 1, 1)).astype(np.float32) / k
        ph = rng.uniform(0, 2 * np.pi, (n, 3, 1, 1)).astype(np.float32)
        x += a * np.sin(lat) ** k * np.cos(k * lon + ph)
        y += 0.9 * a * np.sin(lat) ** k * np.cos(k * (lon - 0.3) + ph)
    return x, y


def ensure_data(data_dir, n_train, n_test):
    os.makedirs(data_dir, exist_ok=True)
    for name, n, (nlat, nlon), seed in (("train_SWE_32x64", n_train, (32, 64), 1), ("test_SWE_32x64", n_test, (32, 64), 2),
                                         ("test_SWE_64x128", n_test, (64, 128), 3)):
        f = os.path.join(data_dir, name + ".npy")
        if not os.path.exists(f):
            logger.warning(f"{f} not found: writing a synthetic stand-in of the same format")
            x, y = synthetic_swe(n, nlat, nlon, seed)
            np.save(f, {"x": x, "y": y}, allow_pickle=True)


if __name__ == "__main__":
    cfg = parse(dict(seed=666, output_dir="./output_sfno", data_dir="./datasets/SWE", epochs=10, n_train=64, n_test=16, batch_size=4,
                     n_modes=32, hidden_channels=32, lifting_channels=256, projection_channels=64, n_layers=4, norm="group_norm",
                     learning_rate=5e-3, log_freq=8))
    ppsci.utils.misc.set_random_seed(cfg["seed"])
    logger.init_logger("ppsci", os.path.join(cfg["output_dir"], "train.log"))
    ensure_data(cfg["data_dir"], cfg["n_train"], cfg["n_test"])

    def loader(split, shuffle):
        return {"dataset": {"name": "SphericalSWEDataset", "data_dir": cfg["data_dir"], "input_keys": ("x",), "label_keys": ("y",),
                            "train_resolution": "32x64", "test_resolutions": ["32x64", "64x128"], "data_split": split},
                "sampler": {"name": "BatchSampler", "drop_last": False, "shuffle": shuffle}, "batch_size": cfg["batch_size"]}

    train_loss = ppsci.loss.LpLoss_train(d=2, p=2, reduce_dims=[0, 1])  # (three output fields: summed over batch and channel)
    sup = ppsci.constraint.SupervisedConstraint(loader("train", True), loss=ppsci.loss.FunctionalLoss(train_loss), name="Sup")
    metric = {"l2": ppsci.metric.FunctionalMetric(ppsci.loss.LpLoss(d=2, p=2, reduce_dims=[0, 1]))}
    validator = {
        "Sup_Validator_32x64": ppsci.validate.SupervisedValidator(loader("test_32x64", False), ppsci.loss.FunctionalLoss(train_loss),
                                                                  metric=metric, name="Sup_Validator_32x64"),
        "Sup_Validator_64x128": ppsci.validate.SupervisedValidator(loader("test_64x128", False), ppsci.loss.FunctionalLoss(train_loss),
                                                                   metric=metric, name="Sup_Validator_64x128"),
    }
    model = ppsci.arch.SFNONet(("x",), ("y",), (cfg["n_modes"], cfg["n_modes"]), cfg["hidden_channels"], 3, 3, cfg["lifting_channels"],
                               cfg["projection_channels"], cfg["n_layers"], norm=cfg["norm"])
    opt = ppsci.optimizer.Adam(cfg["learning_rate"])(model)
    solver = ppsci.solver.Solver(model, {sup.name: sup}, cfg["output_dir"], opt, epochs=cfg["epochs"],
                                 iters_per_epoch=len(sup.data_loader), log_freq=cfg["log_freq"], eval_during_train=True,
                                 eval_freq=max(1, cfg["epochs"] // 2), validator=validator)
    solver.train()
    solver.eval()
