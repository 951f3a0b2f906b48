"""ppsci_reduce_rows_multi_adam: the row reductions that end a backward pass + the Adam update in one launch -- against the two
launches it replaces (same arithmetic, same summation order: bit-identical), at kernel level and through Solver.train for the
two engines that use it (SPINN, FNO)."""
import numpy as np
import pytest
import torch

import ppsci
from paddlescience_amd import _lib as L
from paddlescience_amd import device
from paddlescience_amd import hotpath as hp
from tests.common import make_dev_fixture

dev = make_dev_fixture()


def test_kernel_equals_reduce_then_adam(dev):
    d = device.get_device()
    rng = np.random.default_rng(0)
    n = 1000
    params0 = torch.tensor(rng.standard_normal(n).astype(np.float32), device=d)
    # gradient segments covering [100, 400) (16 rows) and [600, 840) (200 rows: the row-group path); a loss segment outside the
    # gradient buffer (600 rows x 2 columns: the narrow path); the rest of the gradient written by "another kernel"
    g_other = torch.tensor(rng.standard_normal(n).astype(np.float32), device=d)
    p1 = torch.tensor(rng.standard_normal((16, 300)).astype(np.float32), device=d)
    p2 = torch.tensor(rng.standard_normal((200, 240)).astype(np.float32), device=d)
    pl = torch.tensor(rng.standard_normal((600, 2)).astype(np.float32), device=d)

    def run(fused):
        params, grad = params0.clone(), g_other.clone()
        m, v = torch.zeros(n, device=d), torch.zeros(n, device=d)
        loss = torch.zeros(2, device=d)
        for t in (1, 2, 3):
            segs = [(p1.data_ptr(), grad[100:400].data_ptr(), 16, 300), (pl.data_ptr(), loss.data_ptr(), 600, 2),
                    (p2.data_ptr(), grad[600:840].data_ptr(), 200, 240)]
            if fused:
                hp.reduce_rows_multi_adam(segs, params, grad, m, v, 1e-2, t, grad_scale=0.5)
            else:
                arr = (L.ReduceSeg * 3)()
                for k, (src, dst, rows, cols) in enumerate(segs):
                    arr[k].partials, arr[k].out, arr[k].rows, arr[k].cols, arr[k].accumulate = src, dst, rows, cols, 0
                L.check(L.lib().ppsci_reduce_rows_multi(3, arr, hp._stream_ptr(grad)))
                hp.adam_step(params, grad, m, v, 1e-2, t, grad_scale=0.5)
        return [x.detach().cpu().numpy() for x in (params, grad, m, v, loss)]

    a, b = run(True), run(False)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert not np.array_equal(a[0], params0.cpu().numpy())
    # overlapping gradient segments are refused
    grad = g_other.clone()
    with pytest.raises(RuntimeError, match="same parameters"):
        hp.reduce_rows_multi_adam([(p1.data_ptr(), grad[100:400].data_ptr(), 16, 300), (p2.data_ptr(), grad[300:540].data_ptr(), 200, 240)],
                                  params0.clone(), grad, torch.zeros(n, device=d), torch.zeros(n, device=d), 1e-2, 1)


def _spinn_solver(tmp_path, tag):
    np.random.seed(111)
    model = ppsci.arch.SPINN(("x", "y", "z"), ("u",), r=16, num_layers=2, hidden_size=16, activation="tanh")
    eq = ppsci.equation.Helmholtz(3, 1.0)
    eq.model = model
    rng = np.random.default_rng(3)
    xs = [rng.uniform(-1, 1, (n, 1)).astype(np.float32) for n in (9, 9, 9)]
    uc = rng.standard_normal((9, 9, 9, 1)).astype(np.float32)
    data = {"x": xs[0], "y": xs[1], "z": xs[2], "uc": uc}
    pde = ppsci.constraint.SupervisedConstraint(
        {"dataset": {"name": "ContinuousNamedArrayDataset", "input": lambda: data, "label": lambda d: {"helmholtz": d["uc"]}}},
        output_expr=eq.equations, loss=ppsci.loss.MSELoss("mean"), name="PDE")
    opt = ppsci.optimizer.Adam(1e-3)(model)
    return ppsci.solver.Solver(model, {"PDE": pde}, str(tmp_path / tag), opt, epochs=4, iters_per_epoch=1, log_freq=1), model


def _count_fused(monkeypatch):
    calls = [0]
    orig = hp.reduce_rows_multi_adam

    def counted(*a, **k):
        calls[0] += 1
        return orig(*a, **k)

    monkeypatch.setattr(hp, "reduce_rows_multi_adam", counted)
    return calls


def test_spinn_training_is_unchanged_by_the_fused_tail(dev, tmp_path, monkeypatch):
    out, calls = [], _count_fused(monkeypatch)
    for fused in ("1", "0"):
        monkeypatch.setenv("PPSCI_FUSED_REDUCE_ADAM", fused)
        solver, model = _spinn_solver(tmp_path, fused)
        solver.train()
        out.append((model.flat_params.detach().cpu().numpy().copy(), solver.last_losses["loss"]))
        assert calls[0] == 4  # (four steps in the first run, none in the second)
    assert np.array_equal(out[0][0], out[1][0]) and out[0][1] == out[1][1]


def test_fno_training_is_unchanged_by_the_fused_tail(dev, tmp_path, monkeypatch):
    out, calls = [], _count_fused(monkeypatch)
    for fused in ("1", "0"):
        monkeypatch.setenv("PPSCI_FUSED_REDUCE_ADAM", fused)
        torch.manual_seed(5)
        model = ppsci.arch.TFNO2dNet(("x",), ("y",), 4, 4, hidden_channels=8, lifting_channels=16, projection_channels=16, n_layers=2,
                                     norm="group_norm")
        rng = np.random.default_rng(9)
        x = rng.standard_normal((4, 3, 8, 8)).astype(np.float32)
        y = rng.standard_normal((4, 1, 8, 8)).astype(np.float32)
        cfg = {"dataset": {"name": "NamedArrayDataset", "input": {"x": x}, "label": {"y": y}}, "batch_size": 4,
               "sampler": {"name": "BatchSampler", "shuffle": False, "drop_last": True}}
        cst = ppsci.constraint.SupervisedConstraint(cfg, ppsci.loss.MSELoss("mean"), name="Sup")
        opt = ppsci.optimizer.Adam(1e-3)(model)
        solver = ppsci.solver.Solver(model, {"Sup": cst}, str(tmp_path / fused), opt, epochs=4, iters_per_epoch=1, log_freq=1)
        solver.train()
        out.append((model.flat_params.detach().cpu().numpy().copy(), solver.last_losses["loss"]))
        assert calls[0] == 4
    assert np.array_equal(out[0][0], out[1][0]) and out[0][1] == out[1][1]
