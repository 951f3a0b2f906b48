"""Operator-learning data path against tests/golden/neuralop.npz -- produced by executing the REFERENCE's own
ppsci/data/dataset/darcyflow_dataset.py and examples/neuraloperator/metric.py under the torch-backed paddle shim
(tests/golden/make_neuralop_golden.py): DarcyFlowDataset items of every split (positional-encoding channels, unit-Gaussian
encoders), PositionalEmbedding2D, LpLoss / H1Loss and their training variants."""
import os

import numpy as np
import pytest
import torch

import ppsci
from ppsci.data.dataset.darcyflow_dataset import DarcyFlowDataset, PositionalEmbedding2D
from tests.common import make_dev_fixture, rel

dev = make_dev_fixture()

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "neuralop.npz"))


@pytest.fixture()
def darcy_dir(tmp_path):
    for k in ("train_16", "test_16", "test_32"):
        np.save(tmp_path / f"darcy_{k}.npy", {"x": G[f"raw/{k}/x"], "y": G[f"raw/{k}/y"]}, allow_pickle=True)
    return str(tmp_path)


@pytest.mark.parametrize("tag,kw", [("default", {}), ("enc_in", dict(encode_input=True))])
def test_darcy_flow_dataset_matches_reference_run(darcy_dir, tag, kw):
    for split in ("train", "test_16x16", "test_32x32"):
        ds = DarcyFlowDataset(("x",), ("y",), darcy_dir, test_resolutions=[16, 32], train_resolution=16, data_split=split, **kw)
        assert len(ds) == int(G[f"{tag}/{split}/len"])
        for i in (0, len(ds) - 1):
            inp, lab, w = ds[i]
            assert inp["x"].dtype == np.float32 and inp["x"].shape == G[f"{tag}/{split}/{i}/x"].shape
            np.testing.assert_allclose(inp["x"], G[f"{tag}/{split}/{i}/x"], rtol=2e-5, atol=2e-6)
            np.testing.assert_allclose(lab["y"], G[f"{tag}/{split}/{i}/y"], rtol=2e-5, atol=2e-6)
        # batch-index path of this framework: a whole batch in one call equals the stacked items
        idx = np.arange(min(3, len(ds)))
        inp, lab, _ = ds[idx]
        np.testing.assert_array_equal(inp["x"][2 if len(idx) > 2 else 0], ds[int(idx[-1])][0]["x"])
    np.testing.assert_allclose(ds.output_encoder.mean, G[f"{tag}/out_mean"], rtol=1e-6)
    np.testing.assert_allclose(ds.output_encoder.std, G[f"{tag}/out_std"], rtol=1e-5)
    z = G["raw/test_16/y"][:2, None]
    np.testing.assert_allclose(ds.output_encoder.decode(ds.output_encoder.encode(z.copy())), G[f"{tag}/decode"], rtol=1e-5, atol=1e-7)
    if "encode_input" in kw:
        np.testing.assert_allclose(ds.input_encoder.mean, G[f"{tag}/in_mean"], rtol=1e-6)
        np.testing.assert_allclose(ds.input_encoder.std, G[f"{tag}/in_std"], rtol=1e-5)
    with pytest.raises(ValueError):
        DarcyFlowDataset(("x",), ("y",), darcy_dir, test_resolutions=[16, 64])


def test_positional_embedding_2d():
    pe = PositionalEmbedding2D([[0, 1], [-1, 1]])
    out = pe(np.arange(60, dtype=np.float32).reshape(3, 5, 4))
    np.testing.assert_array_equal(out, G["posenc/3x5x4"])
    assert pe(np.zeros((2, 3, 5, 4), np.float32)).shape == (2, 5, 5, 4)


def test_oracle_field_errors_match_reference_run():
    """The fp64 restatement of LpLoss / H1Loss (oracle/ref_torch.field_rel_error) against values computed by the
    reference's own metric.py (tests/golden/make_neuralop_golden.py)."""
    import math

    from oracle import ref_torch as R

    x, y = torch.tensor(G["metric/x"]), torch.tensor(G["metric/y"])
    h = tuple(2 * math.pi / n for n in x.shape[-2:])
    np.testing.assert_allclose(R.field_rel_error(x, y).sum(0).squeeze().numpy() / x.shape[0], G["metric/lp_d2/l2"], rtol=1e-12)
    np.testing.assert_allclose(R.field_rel_error(x, y, p=1).mean(0).squeeze().numpy() / x.shape[0], G["metric/lp_d2_p1_mean/l2"], rtol=1e-12)
    np.testing.assert_allclose(R.field_rel_error(x, y, 1, 2, h).sum(0).squeeze().numpy(), G["metric/h1_train_d2/y"], rtol=1e-12)
    hf = tuple(1.0 / n for n in x.shape[-2:])
    np.testing.assert_allclose(R.field_rel_error(x, y, 1, 2, hf, (True, True)).sum(0).squeeze().numpy() / x.shape[0],
                               G["metric/h1_d2_fix/h1"], rtol=1e-12)


def test_lp_and_h1_losses_match_reference_run(dev):
    """LpLoss / H1Loss on the field-loss kernels (csrc/field_loss.hip) against the reference-run values (fp32 vs fp64), and
    the adjoint they hand the FNO engine against autograd through the fp64 restatement."""
    from oracle import ref_torch as R
    from paddlescience_amd.device import get_device

    d = get_device()
    x64, y64 = torch.tensor(G["metric/x"]), torch.tensor(G["metric/y"])
    if x64.dim() == 3:
        x64, y64 = x64.unsqueeze(1), y64.unsqueeze(1)
    x, y = x64.float().to(d), y64.float().to(d)
    L = ppsci.loss
    cases = {"lp_d2": L.LpLoss(d=2, p=2), "lp_d2_p1_mean": L.LpLoss(d=2, p=1, reductions="mean"),
             "lp_train_d2": L.LpLoss_train(d=2, p=2), "h1_d2": L.H1Loss(d=2),
             "h1_d2_fix": L.H1Loss(d=2, L=1.0, fix_x_bnd=True, fix_y_bnd=True), "h1_train_d2": L.H1Loss_train(d=2)}
    for name, fn in cases.items():
        res = fn({"y": x}, {"y": y})
        for k, v in res.items():
            np.testing.assert_allclose(v.cpu().numpy(), G[f"metric/{name}/{k}"], rtol=3e-6, err_msg=f"{name}/{k}")
    np.testing.assert_allclose(cases["lp_d2"].abs(x, y).cpu().numpy(), G["metric/lp_d2/abs"], rtol=3e-6)
    np.testing.assert_allclose(cases["h1_d2_fix"].abs(x, y).cpu().numpy(), G["metric/h1_d2_fix/abs"], rtol=3e-6)
    # the adjoint w.r.t. the network output (what the operator engine feeds the hand-written backward)
    import math

    for name, order, p, sp, fix in (("h1_train_d2", 1, 2, tuple(2 * math.pi / n for n in x.shape[-2:]), (False, False)),
                                    ("lp_train_d2", 0, 2, (1.0, 1.0), (False, False))):
        xr = x64.clone().requires_grad_(True)
        R.field_rel_error(xr, y64, order, p, sp, fix).sum().backward()
        losses, g = cases[name].value_and_grad(x, y, "y")
        np.testing.assert_allclose(losses["y"].cpu().numpy(), G[f"metric/{name}/y"], rtol=3e-6)
        assert rel(g.cpu().numpy(), xr.grad.numpy()) < 5e-6, name
    hf = tuple(1.0 / n for n in x.shape[-2:])
    xr = x64.clone().requires_grad_(True)
    R.field_rel_error(xr, y64, 1, 2, hf, (True, True)).sum().backward()
    _, g = cases["h1_d2_fix"].rel_and_grad(x, y)
    assert rel(g.cpu().numpy(), xr.grad.numpy()) < 5e-6


def test_tfno_trains_on_the_darcy_dataset_end_to_end(darcy_dir, tmp_path, dev):
    """examples/tfno_darcyflow.py in small: DarcyFlowDataset by name -> SupervisedConstraint(FunctionalLoss(H1Loss_train))
    -> TFNO2dNet -> Solver.train / eval with the H1 / L2 validators at both test resolutions -- on the CPU emulator AND on the
    device (the `dev` fixture injects the emulator for [emu] only; training at 16 x 16 with evaluation at 16 x 16 and 32 x 32
    between epochs is also the executor's buffer-set switch under a replayed HIP graph)."""
    from paddlescience_amd import device

    try:
        ppsci.utils.misc.set_random_seed(3)

        def loader(split, shuffle):
            return {"dataset": {"name": "DarcyFlowDataset", "data_dir": darcy_dir, "input_keys": ("x",), "label_keys": ("y",),
                                "train_resolution": 16, "test_resolutions": [16, 32], "data_split": split},
                    "sampler": {"name": "BatchSampler", "drop_last": False, "shuffle": shuffle}, "batch_size": 3}

        loss = ppsci.loss.FunctionalLoss(ppsci.loss.H1Loss_train(d=2))
        sup = ppsci.constraint.SupervisedConstraint(loader("train", True), loss=loss, name="Sup")
        metric = {"h1": ppsci.metric.FunctionalMetric(ppsci.loss.H1Loss(d=2)),
                  "l2": ppsci.metric.FunctionalMetric(ppsci.loss.LpLoss(d=2, p=2))}
        val = {n: ppsci.validate.SupervisedValidator(loader(s, False), loss, metric=metric, name=n)
               for n, s in (("V16", "test_16x16"), ("V32", "test_32x32"))}
        model = ppsci.arch.TFNO2dNet(("x",), ("y",), 4, 4, 8, 3, 1, 16, 16, 2, norm="group_norm")
        opt = ppsci.optimizer.Adam(2e-3)(model)
        solver = ppsci.solver.Solver(model, {"Sup": sup}, str(tmp_path / "out"), opt, epochs=3, iters_per_epoch=len(sup.data_loader),
                                     log_freq=1, validator=val, eval_during_train=True, eval_freq=1)
        solver.train()
        target, group = solver.eval()
        assert set(group) == {"V16", "V32"} and all(np.isfinite(v) for g in group.values() for v in g.values())
        assert set(group["V16"]) == {"h1.h1", "l2.l2"}
        assert str(model.flat_params.device).startswith("cuda" if dev == "gpu" else "cpu")
    finally:
        pass
