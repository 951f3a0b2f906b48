"""UNO on Darcy flow, after /root/reference/examples/neuraloperator/train_uno.py (+ conf/uno_darcyflow_pretrain.yaml):
DarcyFlowDataset (positional-encoding channels, no output encoding) -> UNONet (five Fourier layers 32-64-64-64-32 on the grids
19 -> 19 -> 10 -> 10 -> 20 -> 19 of the padded 16 x 16 training resolution, U skips 0 -> 4 and 1 -> 3) -> H1Loss_train, validated at
16x16 and 32x32 with the H1 / L2 metrics.  Data: see examples/tfno_darcyflow.py (a synthetic stand-in is written when the
reference's .npy files are absent).

    python examples/uno_darcyflow.py epochs=5
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ppsci  # noqa: E402
from examples._args import parse  # noqa: E402
from examples.tfno_darcyflow import ensure_data  # noqa: E402
from ppsci.utils import logger  # noqa: E402

if __name__ == "__main__":
    cfg = parse(dict(seed=666, output_dir="./output_uno", data_dir="./datasets/darcyflow", epochs=10, n_train=256, n_test=64,
                     batch_size=16, hidden_channels=64, lifting_channels=256, projection_channels=64, width=1.0, modes=1.0,
                     norm="group_norm", domain_padding=0.2, learning_rate=5e-3, log_freq=8, training_loss="h1"))
    ppsci.utils.misc.set_random_seed(cfg["seed"])
    logger.init_logger("ppsci", os.path.join(cfg["output_dir"], "train.log"))
    ensure_data(cfg["data_dir"], cfg["n_train"], cfg["n_test"])

    def loader(split, shuffle):
        return {"dataset": {"name": "DarcyFlowDataset", "data_dir": cfg["data_dir"], "input_keys": ("x",), "label_keys": ("y",),
                            "train_resolution": 16, "test_resolutions": [16, 32], "grid_boundaries": [[0, 1], [0, 1]],
                            "encode_input": False, "encode_output": False, "encoding": "channel-wise", "channel_dim": 1,
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
    w, mo = cfg["width"], cfg["modes"]  # (scale the channel counts / mode counts of the reference config down for a quick run)
    ch = [max(2, int(c * w)) for c in (32, 64, 64, 64, 32)]
    modes = [[max(2, int(m * mo))] * 2 for m in (16, 8, 8, 8, 16)]
    model = ppsci.arch.UNONet(("x",), ("y",), 3, 1, max(2, int(cfg["hidden_channels"] * w)), cfg["lifting_channels"],
                              cfg["projection_channels"], n_layers=5, uno_out_channels=ch, uno_n_modes=modes,
                              uno_scalings=[[1.0, 1.0], [0.5, 0.5], [1, 1], [2, 2], [1, 1]], norm=cfg["norm"],
                              domain_padding=cfg["domain_padding"], domain_padding_mode="one-sided", fft_norm="forward")
    opt = ppsci.optimizer.Adam(cfg["learning_rate"])(model)
    solver = ppsci.solver.Solver(model, {sup.name: sup}, cfg["output_dir"], opt, epochs=cfg["epochs"],
                                 iters_per_epoch=len(sup.data_loader), log_freq=cfg["log_freq"], eval_during_train=True,
                                 eval_freq=max(1, cfg["epochs"] // 2), validator=validator)
    solver.train()
    solver.eval()
