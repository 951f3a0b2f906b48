"""SPINN / Helmholtz3D (BASELINE config 5) through the ppsci API: ppsci.arch.SPINN + ppsci.equation.Helmholtz +
SupervisedConstraint on a tensor-product grid + MSELoss("mean") + Adam, against the torch restatement of
ModifiedMLP / SPINN / Helmholtz (oracle/ref_torch.py) in fp64."""
import numpy as np
import pytest
import torch

import ppsci
from oracle import ref_torch as R
from tests.common import make_dev_fixture, rel

dev = make_dev_fixture()


def _params_of(model, b):
    names = [n for n, _ in model.spec.param_shapes()]
    vals = {}
    off = b * model.branch_params
    flat = model.flat_params.cpu().numpy().astype(np.float64)
    for (n, shp) in model.spec.param_shapes():
        k = int(np.prod(shp))
        vals[n] = flat[off:off + k].reshape(shp)
        off += k
    L = model.spec.L
    return dict(wu=vals["embed_u.0.weight"], bu=vals["embed_u.0.bias"], wv=vals["embed_v.0.weight"], bv=vals["embed_v.0.bias"],
                w=[vals[f"linears.{l}.weight"] for l in range(L)], b=[vals[f"linears.{l}.bias"] for l in range(L)],
                wl=vals["last_fc.weight"], bl=vals["last_fc.bias"])


def _build(tmp_path, shape=(7, 5, 6), act="tanh", r=4, hidden=16, layers=3):
    np.random.seed(111)
    model = ppsci.arch.SPINN(("x", "y", "z"), ("u",), r=r, num_layers=layers, hidden_size=hidden, activation=act)
    with torch.no_grad():  # non-zero biases so that every path is exercised
        model.flat_params.add_(torch.from_numpy(np.random.default_rng(5).uniform(-0.1, 0.1, model.flat_params.numel()).astype(np.float32)).to(model.flat_params.device))
    eq = ppsci.equation.Helmholtz(3, 1.0)
    eq.model = model
    rng = np.random.default_rng(3)
    xs = [rng.uniform(-1, 1, (n, 1)).astype(np.float32) for n in shape]
    uc = rng.standard_normal(shape + (1,)).astype(np.float32)

    def gen_in():
        return {"x": xs[0], "y": xs[1], "z": xs[2], "uc": uc}

    pde = ppsci.constraint.SupervisedConstraint(
        {"dataset": {"name": "ContinuousNamedArrayDataset", "input": gen_in, "label": lambda d: {"helmholtz": d["uc"]}}},
        output_expr=eq.equations, loss=ppsci.loss.MSELoss("mean"), name="PDE")
    face = {"x": np.asarray([[1.0]], np.float32), "y": xs[1], "z": xs[2]}
    bc = ppsci.constraint.SupervisedConstraint(
        {"dataset": {"name": "ContinuousNamedArrayDataset", "input": lambda: face,
                     "label": lambda d: {"u": np.zeros([1, shape[1], shape[2], 1], np.float32)}}},
        output_expr={"u": lambda out: out["u"]}, loss=ppsci.loss.MSELoss("mean"), name="BC0")
    opt = ppsci.optimizer.Adam(1e-3)(model)
    solver = ppsci.solver.Solver(model, {"PDE": pde, "BC0": bc}, str(tmp_path), opt, epochs=1, iters_per_epoch=1, log_freq=1,
                                 equation={"Helmholtz": eq})
    return solver, model, xs, uc, face


@pytest.mark.parametrize("act,shape,r,hidden,layers", [
    ("tanh", (7, 5, 6), 4, 16, 3), ("sin", (7, 5, 6), 4, 16, 3), ("tanh", (18, 35, 21), 20, 16, 3), ("tanh", (17, 9, 20), 40, 16, 3),
    ("tanh", (5, 6, 7), 70, 16, 3), ("tanh", (9, 9, 9), 8, 16, 3),
    # width and rank multiples of 16: the reverse sweep of the branch nets by 16-point tiles on the matrix cores
    # (modmlp_bwd_tile_kernel): ragged tiles, 1..4 feature blocks, 1..3 rank blocks, one hidden layer, a 1-point face
    ("tanh", (18, 35, 21), 32, 32, 3), ("sin", (7, 5, 6), 16, 16, 1), ("tanh", (17, 33, 20), 48, 64, 4),
    ("silu", (16, 16, 16), 32, 48, 2)])
def test_spinn_helmholtz_losses_and_grads(tmp_path, act, shape, r, hidden, layers):
    """ranks 4 / 20 / 40 take the MFMA grid kernels with 4 / 8 / 16 k-steps (ragged row, column and rank tails); rank 70
    the scalar ones; equal point counts on the three axes take the joint gradient-row layout (one reduce for all branches)."""
    solver, model, xs, uc, face = _build(tmp_path, shape=shape, act=act, r=r, hidden=hidden, layers=layers)
    nets = [R.ModifiedMLP1(_params_of(model, b), act) for b in range(3)]
    tx = [torch.tensor(x.astype(np.float64), requires_grad=True) for x in xs]
    u, res = R.spinn_helmholtz(nets, tx, 1.0)
    l_pde = ((res - torch.tensor(uc[..., 0].astype(np.float64))) ** 2).mean()
    tf = [torch.tensor(face[k].astype(np.float64), requires_grad=True) for k in ("x", "y", "z")]
    ub, _ = R.spinn_helmholtz(nets, tf, 1.0)
    l_bc = (ub**2).mean()
    params = [p for n in nets for p in n.parameters()]
    gref = torch.autograd.grad(l_pde + l_bc, params, allow_unused=True)
    gref = np.concatenate([(torch.zeros_like(p) if g is None else g).numpy().ravel() for g, p in zip(gref, params)])

    csts = list(solver._compiled.values())
    for name, cc in solver._compiled.items():
        inp, lab, _ = next(solver.constraint[name].data_iter)
        cc.bind(inp, lab)
    solver.engine.forward_backward(csts)
    assert solver._compiled["PDE"].loss() == pytest.approx(float(l_pde), rel=3e-5)
    assert solver._compiled["BC0"].loss() == pytest.approx(float(l_bc), rel=3e-5)
    assert rel(solver.engine.grad.cpu().numpy(), gref) < 1e-4
    # eager forward == predict on the grid
    pred = solver.predict({"x": xs[0], "y": xs[1], "z": xs[2]}, batch_size=None, return_numpy=True)["u"]
    assert pred.shape == tuple(shape) + (1,)
    assert rel(pred[..., 0], u.detach().numpy()) < 5e-6


def test_tile_and_point_kernels_of_the_reverse_sweep_agree(tmp_path):
    """The same step with the per-point reverse kernel (ppsci_set_modmlp_tile(0)) and with the tile kernel: equal up to the
    order of the sums."""
    from paddlescience_amd import _lib as L

    grads = []
    for tile in (1, 0, 2):
        L.lib().ppsci_set_modmlp_tile(tile)
        try:
            solver, model, xs, uc, face = _build(tmp_path / f"t{tile}", shape=(19, 16, 33), r=32, hidden=64, layers=3)
            csts = list(solver._compiled.values())
            rows = int(L.lib().ppsci_modmlp_bwd_rows(model.spec.desc, 19))
            assert rows == (2 if tile else 19)
            for name, cc in solver._compiled.items():
                inp, lab, _ = next(solver.constraint[name].data_iter)
                cc.bind(inp, lab)
            solver.engine.forward_backward(csts)
            grads.append(solver.engine.grad.cpu().numpy().copy())
        finally:
            L.lib().ppsci_set_modmlp_tile(1)
    assert rel(grads[0], grads[1]) < 2e-6 and rel(grads[2], grads[1]) < 2e-6


def test_group_partials_summed_by_the_branch_kernel_equal_the_sum_launch(tmp_path, monkeypatch):
    """ppsci_spinn_grid_bwd without its partial-sum launch + ppsci_modmlp_bwd_batch_parts (the tile kernel sums the grid kernel's
    per-group partials of dL/dF on load, in that launch's order): bit-identical gradients."""
    grads = []
    for parts in ("1", "0"):  # (off by default: slower at 3 x 128 points, spinn_engine.SpinnConstraint.backward)
        monkeypatch.setenv("PPSCI_SPINN_PARTS", parts)
        solver, model, xs, uc, face = _build(tmp_path / f"p{parts}", shape=(19, 16, 33), r=32, hidden=32, layers=2)
        csts = list(solver._compiled.values())
        for name, cc in solver._compiled.items():
            inp, lab, _ = next(solver.constraint[name].data_iter)
            cc.bind(inp, lab)
        solver.engine.forward_backward(csts)
        grads.append(solver.engine.grad.cpu().numpy().copy())
    assert np.array_equal(grads[0], grads[1]) and np.abs(grads[0]).sum() > 0


def test_spinn_training_step_runs(tmp_path):
    solver, model, *_ = _build(tmp_path)
    p0 = model.flat_params.clone()
    solver.train()
    assert torch.isfinite(model.flat_params).all() and not torch.equal(p0, model.flat_params)
    assert set(solver.last_losses) == {"loss", "PDE", "BC0"}
