"""FNO network path (BASELINE config 4): TFNO2dNet forward / parameter gradients against the
plain-torch fp64 restatement of FNONet.forward + FNOBlocks.forward_with_postactivation (oracle/ref_torch.py,
/root/reference/ppsci/arch/tfnonet.py:179-193, fno_block.py:1191-1220), and a few Solver steps against the
oracle's Adam.  Tolerances: fp32 vs fp64, forward rel-L2 <= 2e-5, gradients <= 2e-4 (GroupNorm amplifies)."""
import numpy as np
import pytest
import torch

from oracle import ref_torch as R
from tests.common import make_dev_fixture, rel

dev = make_dev_fixture()


def _model(norm, seed=0, grid=8):
    import ppsci

    torch.manual_seed(seed)
    return ppsci.arch.TFNO2dNet(("x",), ("y",), 4, 4, hidden_channels=8, in_channels=3, out_channels=1,
                                lifting_channels=16, projection_channels=16, n_layers=2, norm=norm)


def _params64(model):
    return {k: v.detach().double().cpu().requires_grad_(True) for k, v in torch.nn.Module.state_dict(model).items()}


@pytest.mark.parametrize("norm", [None, "group_norm"])
def test_tfno2d_forward_and_gradients_match_oracle(norm, dev):
    model = _model(norm)
    d = model.flat_params.device
    rng = np.random.default_rng(42)
    x = rng.standard_normal((3, 3, 8, 8)).astype(np.float32)
    tgt = rng.standard_normal((3, 1, 8, 8)).astype(np.float32)
    out = model({"x": x})["y"]  # FNONet.forward runs the kernels' executor (the only implementation of the network)
    assert out.shape == (3, 1, 8, 8)
    import ppsci

    nat = model.native()
    y = nat.forward(torch.as_tensor(x).to(d))
    assert torch.equal(y, out)
    _, gy = ppsci.loss.MSELoss("mean").value_and_grad(y, torch.as_tensor(tgt).to(d), "y")
    model.flat_grad.fill_(float("nan"))
    nat.backward(gy)
    P = _params64(model)
    ref = R.fno_forward(torch.as_tensor(x).double(), P, 2, (4, 4), norm)
    assert rel(out.detach().cpu().numpy(), ref.detach().numpy()) < 2e-5
    lref = ((ref - torch.as_tensor(tgt).double()) ** 2).mean()
    names = [k for k, _ in torch.nn.Module.named_parameters(model)]
    gref = torch.autograd.grad(lref, [P[k] for k in names])
    flat_ref = np.concatenate([g.numpy().ravel() for g in gref])
    got = model.flat_grad.cpu().numpy()
    assert got.shape == flat_ref.shape  # the flat buffer follows named_parameters() order
    assert rel(got, flat_ref) < 2e-4, rel(got, flat_ref)
    # every tensor on its own, so that a small one cannot hide
    off = 0
    for g in gref:
        n = g.numel()
        assert rel(got[off:off + n], g.numpy().ravel()) < 1e-3
        off += n


def test_unbuilt_options_raise():
    import ppsci

    with pytest.raises(NotImplementedError):
        ppsci.arch.TFNO2dNet(("x",), ("y",), 4, 4, 8, use_mlp=True)
    with pytest.raises(NotImplementedError):
        ppsci.arch.TFNO1dNet(("x",), ("y",), 4, 8)
    with pytest.raises(NotImplementedError):
        ppsci.arch.FNONet(("x",), ("y",), (4, 4), 8, preactivation=True)
    # (domain_padding is built since round 3: tests/test_golden_fno.py pins it to reference-run fixtures)
    assert ppsci.arch.FNONet(("x",), ("y",), (4, 4), 8, domain_padding=0.1).padding_of(16, 20) == (2, 2, 0, 0)


def test_solver_trains_fno_like_oracle_adam(dev, tmp_path):
    """Three supervised steps through ppsci.solver.Solver (FunctionalLoss = mean squared error written by the
    user in torch, as in examples/neuraloperator/train_tfno.py) == three oracle Adam steps in fp64."""
    import ppsci

    model = _model("group_norm", seed=1)
    P = _params64(model)
    names = [k for k, _ in torch.nn.Module.named_parameters(model)]
    rng = np.random.default_rng(7)
    x = rng.standard_normal((4, 3, 8, 8)).astype(np.float32)
    y = rng.standard_normal((4, 1, 8, 8)).astype(np.float32)

    def mse(output_dict, label_dict, weight_dict=None):
        return {"l2": ((output_dict["y"] - label_dict["y"]) ** 2).mean()}

    cfg = {"dataset": {"name": "NamedArrayDataset", "input": {"x": x}, "label": {"y": y}}, "batch_size": 4,
           "sampler": {"name": "BatchSampler", "shuffle": False, "drop_last": True}}
    cst = ppsci.constraint.SupervisedConstraint(cfg, ppsci.loss.FunctionalLoss(mse), name="Sup")
    val = ppsci.validate.SupervisedValidator(cfg, ppsci.loss.FunctionalLoss(mse), metric={"MSE": ppsci.metric.MSE()},
                                             name="V")
    opt = ppsci.optimizer.Adam(learning_rate=1e-3)(model)
    solver = ppsci.solver.Solver(model, {"Sup": cst}, str(tmp_path), opt, epochs=3, iters_per_epoch=1, log_freq=1,
                                 validator={"V": val})
    solver.train()
    # oracle: Adam (paddle semantics) on the flat fp64 parameter vector
    shapes = [tuple(P[k].shape) for k in names]
    flat_ref = np.concatenate([P[k].detach().numpy().ravel() for k in names])
    adam = R.Adam(flat_ref.size, lr=1e-3)
    last = None
    for _ in range(3):
        Q, off = dict(P), 0
        for k, sh in zip(names, shapes):
            n = int(np.prod(sh))
            Q[k] = torch.tensor(flat_ref[off:off + n].reshape(sh), dtype=torch.float64, requires_grad=True)
            off += n
        out = R.fno_forward(torch.as_tensor(x).double(), Q, 2, (4, 4), "group_norm")
        loss = ((out - torch.as_tensor(y).double()) ** 2).mean()
        last = float(loss.detach())
        g = torch.autograd.grad(loss, [Q[k] for k in names])
        flat_ref = adam.step(flat_ref, np.concatenate([t.numpy().ravel() for t in g]))
    assert abs(solver.last_losses["loss"] - last) < 1e-4 * max(1.0, abs(last))
    np.testing.assert_allclose(model.flat_params.cpu().numpy(), flat_ref, rtol=0, atol=5e-5)
    metric, group = solver.eval()
    assert np.isfinite(metric) and "MSE.y" in group["V"]
    pred = solver.predict({"x": x}, batch_size=2, return_numpy=True)
    assert pred["y"].shape == (4, 1, 8, 8)


def test_training_survives_evaluation_at_another_shape(dev, tmp_path):
    """FNONet.forward (train, eval, predict) is ONE executor and the training step is a replayed HIP graph that holds the
    executor's buffer addresses (the reference's TFNO config trains at 16 x 16 and evaluates at 32 x 32 with
    eval_during_train; a ragged last eval batch changes the batch size): evaluating at another shape between training
    steps must leave those buffers alone -- the parameters after train / eval-elsewhere / train are bit for bit those of
    the same steps without the evaluation."""
    import ppsci

    rng = np.random.default_rng(3)
    x = rng.standard_normal((4, 3, 8, 8)).astype(np.float32)
    y = rng.standard_normal((4, 1, 8, 8)).astype(np.float32)
    x_big = rng.standard_normal((3, 3, 16, 16)).astype(np.float32)   # another resolution AND another batch size

    def run(evaluate_between):
        model = _model("group_norm", seed=5)
        cfg = {"dataset": {"name": "NamedArrayDataset", "input": {"x": x}, "label": {"y": y}}, "batch_size": 4,
               "sampler": {"name": "BatchSampler", "shuffle": False, "drop_last": True}}
        cst = ppsci.constraint.SupervisedConstraint(cfg, ppsci.loss.MSELoss("mean"), name="Sup")
        opt = ppsci.optimizer.Adam(learning_rate=1e-3)(model)
        solver = ppsci.solver.Solver(model, {"Sup": cst}, str(tmp_path), opt, epochs=4, iters_per_epoch=1, log_freq=10)
        solver.train()  # (the second step captures the graph, the later ones replay it)
        if evaluate_between:
            for _ in range(3):
                out = model({"x": x_big})["y"]
                assert out.shape == (3, 1, 16, 16) and torch.isfinite(out).all()
                # (what the caching allocator would hand the freed training buffers to)
                junk = [torch.full((4, 8, 64), float("nan"), device=model.flat_params.device) for _ in range(16)]
                del junk
            assert len(model.native()._sets) == 1 and model.native().shape == (3, 16, 16)
        solver.epochs = 8
        solver.train()
        return model.flat_params.detach().cpu().numpy().copy()

    a, b = run(False), run(True)
    assert np.isfinite(a).all() and np.array_equal(a, b)


def test_field_losses_combine_under_autograd(dev):
    """LpLoss / H1Loss values are differentiable functions of the field (loss/field.py scalar_loss): a user function that
    combines them (FunctionalLoss: l2 + h1, weights, a non-identity output expression) trains -- the operator engine
    differentiates that Python w.r.t. the network output with torch."""
    import ppsci
    from paddlescience_amd.loss.lp_h1 import H1Loss, LpLoss

    model = _model(None, seed=2)
    d = model.flat_params.device
    rng = np.random.default_rng(1)
    xt = torch.as_tensor(rng.standard_normal((2, 1, 8, 8)).astype(np.float32)).to(d).requires_grad_(True)
    yt = torch.as_tensor(rng.standard_normal((2, 1, 8, 8)).astype(np.float32)).to(d)
    l2, h1 = LpLoss(d=2, p=2), H1Loss(d=2)
    total = 0.7 * l2.rel(xt, yt) + 0.3 * h1.rel(xt, yt)
    (g,) = torch.autograd.grad(total, xt)
    _, g2 = l2.rel_and_grad(xt.detach(), yt, scale=0.7)
    _, g1 = h1.rel_and_grad(xt.detach(), yt, scale=0.3)
    assert rel(g.cpu().numpy(), (g2 + g1).cpu().numpy()) < 1e-6
    # values handed out earlier are not overwritten by later calls on the same plan
    v1 = l2.rel(xt.detach(), yt)
    keep = float(v1)
    l2.rel(2.0 * xt.detach(), yt)
    assert float(v1) == keep
