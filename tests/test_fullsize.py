"""Full BASELINE sizes on the GPU, checked through size-independent properties (the fp64 oracle takes minutes at
these sizes): point independence (a random subset re-evaluated by the oracle), additivity of the parameter
gradient over shards of the batch, linearity of the reverse sweep in dL/dU, loss == mean of residual^2.
Plus the small edge cases of the tile machinery on the emulator and the GPU: single point, tile boundaries,
maximum input / output / stream counts, empty batch."""
import numpy as np
import pytest
import torch

from oracle import taylor_np as T
from tests import test_kernels as K
from tests.common import make_dev_fixture

dev = make_dev_fixture()


def _sync_device(dev):
    K.DEVICE = "cuda" if dev == "gpu" else "cpu"


CONFIGS = {
    # name: (hidden, d_out, dirs, n2, N)  -- BASELINE configs[1] and the per-GPU shard of configs[2]
    "allen_cahn_100k": ([64] * 4, 1, [[0.0, 1.0], [1.0, 0.0]], 1, 100_000),
    "navier_stokes_125k": ([128] * 5, 3, [[1.0, 0.0], [0.0, 1.0]], 2, 125_000),
}


@pytest.mark.parametrize("name", list(CONFIGS))
def test_full_size_properties(name, dev):
    if dev != "gpu":
        pytest.skip("full BASELINE sizes run on the GPU only")
    _sync_device(dev)
    hidden, dout, dirs, n2, N = CONFIGS[name]
    dirs = np.asarray(dirs, dtype=np.float64)
    S = 1 + len(dirs) + n2
    net = T.make_net(2, hidden, dout, seed=1234, bias_scale=0.05)
    rng = np.random.default_rng(42)
    X = rng.uniform(-1, 1, (N, 2)).astype(np.float32).astype(np.float64)
    # 1. point independence: a random subset against the fp64 oracle
    _, _, _, U, _ = K._run_fwd(net, X, dirs, n2)
    idx = np.sort(rng.choice(N, 512, replace=False))
    idx[:3] = [0, N - 1, N - (N % 16 or 16)]  # first / last point, first point of the last tile
    net32 = net.astype(np.float32).astype(np.float64)
    ref = T.taylor_forward(net32, X[idx], dirs, n2).reshape(-1, len(idx))
    got = U.cpu().numpy().astype(np.float64)[:, idx]
    tol = 2e-6 if max(hidden) <= 64 else 5e-6  # fp32 accumulation over 128-wide x 5 layers: 2.6e-6 measured
    for q in range(ref.shape[0]):
        assert K._rel(got[q], ref[q]) < tol, (q, K._rel(got[q], ref[q]))
    # 2. additivity over shards and linearity in dL/dU of the reverse sweep
    Ubar = (rng.standard_normal((dout, S, N)) / N).astype(np.float32).astype(np.float64)
    g_all = K._run_bwd(net, X, dirs, n2, Ubar)
    h = (N // 2 // 16) * 16 + 5  # ragged split point
    g_a = K._run_bwd(net, X[:h], dirs, n2, Ubar[:, :, :h])
    g_b = K._run_bwd(net, X[h:], dirs, n2, Ubar[:, :, h:])
    assert np.isfinite(g_all).all()
    assert K._rel(g_a + g_b, g_all) < 2e-5, K._rel(g_a + g_b, g_all)
    g_2 = K._run_bwd(net, X, dirs, n2, 2.0 * Ubar)
    assert K._rel(g_2, 2.0 * g_all) < 2e-6  # scaling by 2 is exact; only the atomic-add order may differ
    # 3. the subset's own gradient against the oracle (same weights, same points)
    _, cache = T.taylor_forward(net32, X[idx], dirs, n2, keep=True)
    gref = T.flat_grads(*T.taylor_backward(net32, cache, Ubar[:, :, idx]))
    g_sub = K._run_bwd(net, X[idx], dirs, n2, Ubar[:, :, idx])
    assert K._rel(g_sub, gref) < (5e-6 if max(hidden) <= 64 else 1e-5)


@pytest.mark.parametrize("N", [1, 15, 16, 17, 31, 33])
def test_tile_boundaries(N, dev):
    _sync_device(dev)
    net = T.make_net(2, [20, 20], 1, bias_scale=0.2)
    rng = np.random.default_rng(N)
    X = rng.uniform(-1, 1, (N, 2)).astype(np.float32).astype(np.float64)
    dirs = np.eye(2)
    Ubar = rng.standard_normal((1, 5, N)).astype(np.float32).astype(np.float64)
    _, _, _, U, _ = K._run_fwd(net, X, dirs, 2)
    net32 = net.astype(np.float32).astype(np.float64)
    Uref, cache = T.taylor_forward(net32, X, dirs, 2, keep=True)
    assert K._rel(U.cpu().numpy().astype(np.float64), Uref.reshape(-1, N)) < 2e-6
    got = K._run_bwd(net, X, dirs, 2, Ubar)
    assert K._rel(got, T.flat_grads(*T.taylor_backward(net32, cache, Ubar))) < 5e-6


def test_maximum_inputs_outputs_and_streams(dev):
    """PPSCI_MAX_IN = 8 raw inputs, PPSCI_MAX_OUT = 8 outputs, (n1, n2) = (3, 3): S = 7 streams."""
    _sync_device(dev)
    net = T.make_net(8, [24, 24], 8, bias_scale=0.1)
    rng = np.random.default_rng(1)
    N = 19
    X = rng.uniform(-1, 1, (N, 8)).astype(np.float32).astype(np.float64)
    dirs = np.zeros((3, 8))
    dirs[0, 0] = dirs[1, 3] = dirs[2, 7] = 1.0
    Ubar = rng.standard_normal((8, 7, N)).astype(np.float32).astype(np.float64)
    _, _, _, U, _ = K._run_fwd(net, X, dirs, 3)
    net32 = net.astype(np.float32).astype(np.float64)
    Uref, cache = T.taylor_forward(net32, X, dirs, 3, keep=True)
    assert K._rel(U.cpu().numpy().astype(np.float64), Uref.reshape(-1, N)) < 2e-6
    got = K._run_bwd(net, X, dirs, 3, Ubar)
    assert K._rel(got, T.flat_grads(*T.taylor_backward(net32, cache, Ubar))) < 5e-6


def test_empty_batch(dev):
    """N = 0: the forward entry point is a no-op that reports success; the sizing helpers return 0 and the
    reverse entry point refuses (there is nothing to size a partial buffer for)."""
    _sync_device(dev)
    import ctypes as C

    from paddlescience_amd import _lib as L
    from paddlescience_amd import hotpath as hp

    lay = hp.NetLayout(2, 2, 16, 1, "tanh")
    desc = lay.desc(hp.StreamSpec([[1.0, 0.0]], 1))
    params = torch.zeros(lay.n_params, device=K.DEVICE)
    ptrs = (C.c_void_p * 2)(None, None)
    rc = L.lib().ppsci_taylor_fwd(C.byref(desc), params.data_ptr(), 0, ptrs, params.data_ptr(), None, None)
    assert rc == 0
    assert hp.stash_bytes(desc, 0) == 0 and hp.bwd_partial_rows(desc, 0) == 0
    rc = L.lib().ppsci_taylor_bwd(C.byref(desc), params.data_ptr(), 0, ptrs, params.data_ptr(), params.data_ptr(),
                                  params.data_ptr(), params.data_ptr(), None)
    assert rc != 0 and b"invalid" in L.lib().ppsci_last_error()


def test_spinn_full_size_128(dev, tmp_path):
    """BASELINE configs[4] at its full size -- 128^3 tensor-product grid, three ModifiedMLP 1 -> 64 x 4 -> 32 branches --
    against the fp64 oracle (oracle/ref_torch.spinn_helmholtz, pinned by the reference-run fixture tests/golden/spinn.npz;
    separable, so the whole grid takes seconds): u on every grid point, the loss, and the gradient of every parameter."""
    if dev != "gpu":
        pytest.skip("full BASELINE sizes run on the GPU only")
    import ppsci
    from oracle import ref_torch as R

    nc = 128
    np.random.seed(111)
    model = ppsci.arch.SPINN(("x", "y", "z"), ("u",), 32, 4, 64, "tanh")
    eq = ppsci.equation.Helmholtz(3, 1.0)
    eq.model = model
    rng = np.random.default_rng(42)
    xs = [rng.uniform(-1, 1, (nc, 1)).astype(np.float32) for _ in range(3)]
    uc = rng.standard_normal((nc, nc, nc, 1)).astype(np.float32)
    data = {"x": xs[0], "y": xs[1], "z": xs[2], "uc": uc}
    lab = {"helmholtz": uc}
    pde = ppsci.constraint.SupervisedConstraint(
        {"dataset": {"name": "ContinuousNamedArrayDataset", "input": lambda: data, "label": lambda d: lab}},
        output_expr=eq.equations, loss=ppsci.loss.MSELoss("mean"), name="PDE")
    opt = ppsci.optimizer.Adam(1e-3)(model)
    solver = ppsci.solver.Solver(model, {"PDE": pde}, str(tmp_path), opt, epochs=1, iters_per_epoch=1)
    cc = solver._compiled["PDE"]
    cc.bind(data, lab)
    solver.engine.forward_backward([cc])
    grad = solver.engine.grad.detach().cpu().numpy().astype(np.float64)
    sd = {k: np.asarray(v.detach().cpu().numpy(), np.float64) for k, v in model.state_dict().items()}
    nets, names = [], []
    for b in range(3):
        P = {k.split(".", 2)[2]: v for k, v in sd.items() if k.startswith(f"branch_nets.{b}.")}
        nl = sum(1 for k in P if k.startswith("linears.") and k.endswith(".weight"))
        nets.append(R.ModifiedMLP1(dict(wu=P["embed_u.0.weight"], bu=P["embed_u.0.bias"], wv=P["embed_v.0.weight"],
                                        bv=P["embed_v.0.bias"], w=[P[f"linears.{l}.weight"] for l in range(nl)],
                                        b=[P[f"linears.{l}.bias"] for l in range(nl)], wl=P["last_fc.weight"],
                                        bl=P["last_fc.bias"]), "tanh"))
    xt = [torch.tensor(a.astype(np.float64), requires_grad=True) for a in xs]
    uo, ro = R.spinn_helmholtz(nets, xt, 1.0)
    loss = ((ro - torch.tensor(uc[..., 0].astype(np.float64))) ** 2).mean()
    pred = solver.predict({"x": xs[0], "y": xs[1], "z": xs[2]}, batch_size=None, return_numpy=True)["u"]
    assert K._rel(pred[..., 0].astype(np.float64), uo.detach().numpy()) < 5e-6
    assert abs(cc.loss() / float(loss.detach()) - 1.0) < 1e-5
    # the flat gradient follows model.parameters() order; the oracle's branch parameters are listed in the same order
    go = torch.autograd.grad(loss, [p for net in nets for p in net.parameters()])
    flat_ref = np.concatenate([g.numpy().ravel() for g in go])
    assert grad.shape == flat_ref.shape
    assert K._rel(grad, flat_ref) < 1e-4, K._rel(grad, flat_ref)


def test_fused_tile_step_full_size_100k(dev):
    """BASELINE configs[1] through the fused tile kernel (csrc/taylor_fused.inc) at its full size: the gradient of the 100 000
    point batch equals the separate launches' (which the test above holds to the oracle point by point), shard additivity,
    and the loss equals the mean of the squared residuals it writes out."""
    if dev != "gpu":
        pytest.skip("full BASELINE sizes run on the GPU only")
    from paddlescience_amd import hotpath as hp
    from paddlescience_amd.engine import Engine
    from tests.test_one_launch import _constraint, _weights

    d = torch.device("cuda")
    lay = hp.NetLayout(2, 4, 64, 1, "tanh")
    flat = _weights(lay, 7)
    N = 100_000
    out = {}
    for fused in (False, True):
        eng = Engine(lay, torch.tensor(flat, device=d))
        eng.one_launch = fused
        c = _constraint(d, "allen_cahn", lay, N, 100)
        eng.forward_backward([c])
        torch.cuda.synchronize()
        out[fused] = (eng.grad.cpu().numpy().astype(np.float64), c.loss_terms.cpu().numpy().copy(), c.resid.cpu().numpy().copy(), c)
    assert out[True][3]._step_kind == hp.STEP_FUSED_TILE
    assert K._rel(out[True][0], out[False][0]) < 3e-6
    np.testing.assert_allclose(out[True][1], out[False][1], rtol=2e-5)
    np.testing.assert_allclose(out[True][1][0], np.mean(out[True][2][0].astype(np.float64) ** 2), rtol=1e-5)
    # additivity over two ragged shards of the same points (each normalised by ITS size: rescale to the whole batch)
    c_all = out[True][3]
    h = (N // 2 // 16) * 16 + 5
    g = np.zeros_like(out[True][0])
    for lo, hi in ((0, h), (h, N)):
        eng = Engine(lay, torch.tensor(flat, device=d))
        cs = _constraint(d, "allen_cahn", lay, hi - lo, 100)
        for dst, src in zip(cs.inputs, c_all.inputs):
            dst.copy_(src[lo:hi])
        eng.forward_backward([cs])
        g += eng.grad.cpu().numpy().astype(np.float64) * (hi - lo) / N
    assert K._rel(g, out[True][0]) < 2e-5


def test_tfno_full_size_batch16_64x64(dev):
    """BASELINE configs[3] at its full size (batch 16, 64 x 64, hidden 32, lifting 256, projection 64, 4 blocks, GroupNorm)
    against the fp64 oracle (oracle/ref_torch.fno_forward, pinned by tests/golden/fno.npz): output, loss, every gradient."""
    if dev != "gpu":
        pytest.skip("full BASELINE sizes run on the GPU only")
    import ppsci
    from oracle import ref_torch as R

    torch.manual_seed(0)
    model = ppsci.arch.TFNO2dNet(("x",), ("y",), 12, 12, hidden_channels=32, in_channels=3, out_channels=1,
                                 lifting_channels=256, projection_channels=64, n_layers=4, norm="group_norm")
    x = torch.as_tensor(np.random.default_rng(42).standard_normal((16, 3, 64, 64)).astype(np.float32)).cuda()
    y = torch.as_tensor(np.random.default_rng(43).standard_normal((16, 1, 64, 64)).astype(np.float32)).cuda()
    P = {n: p.detach().cpu().double().requires_grad_(True) for n, p in torch.nn.Module.named_parameters(model)}
    yo = R.fno_forward(x.cpu().double(), P, 4, (12, 12), "group_norm")
    lo = ((yo - y.cpu().double()) ** 2).mean()
    names = sorted(P)
    go = dict(zip(names, torch.autograd.grad(lo, [P[n] for n in names])))
    nat = model.native()
    yh = nat.forward(x)
    losses, gy = ppsci.loss.MSELoss("mean").value_and_grad(yh, y, "y")
    assert K._rel(yh.cpu().numpy(), yo.detach().numpy()) < 2e-6
    assert abs(float(losses["y"]) / float(lo) - 1.0) < 1e-5
    model.flat_grad.fill_(float("nan"))
    nat.backward(gy)
    for n, p in torch.nn.Module.named_parameters(model):
        assert K._rel(p.grad.cpu().numpy(), go[n].numpy()) < 2e-5, n


@pytest.mark.parametrize("res,full_fft", [(16, False), (16, True), (32, False), (64, False)])
def test_uno_reference_config_full_size(dev, res, full_fft, monkeypatch):
    """The reference's UNO configuration (conf/uno_darcyflow_pretrain.yaml: hidden 64, lifting 256, projection 64, layers
    32-64-64-64-32, modes 16-8-8-8-16, scalings 1 / 0.5 / 1 / 2 / 1, GroupNorm, domain padding 0.2, batch 16) at its training
    resolution 16, its second evaluation resolution 32 and at 64, against the fp64 oracle (oracle/ref_torch.uno_forward, pinned by
    the reference-run tests/golden/uno.npz): output, loss, every gradient."""
    if dev != "gpu":
        pytest.skip("full sizes run on the GPU only")
    import ppsci
    from oracle import ref_torch as R

    monkeypatch.setenv("PPSCI_FNO_FULL_FFT", "1" if full_fft else "0")
    torch.manual_seed(0)
    outs, modes = [32, 64, 64, 64, 32], [[16, 16], [8, 8], [8, 8], [8, 8], [16, 16]]
    scal = [[1.0, 1.0], [0.5, 0.5], [1, 1], [2, 2], [1, 1]]
    model = ppsci.arch.UNONet(("x",), ("y",), 3, 1, 64, 256, 64, n_layers=5, uno_out_channels=outs, uno_n_modes=modes,
                              uno_scalings=scal, norm="group_norm", domain_padding=0.2, domain_padding_mode="one-sided")
    B = 16 if res <= 32 else 4
    x = torch.as_tensor(np.random.default_rng(42).standard_normal((B, 3, res, res)).astype(np.float32)).cuda()
    y = torch.as_tensor(np.random.default_rng(43).standard_normal((B, 1, res, res)).astype(np.float32)).cuda()
    P = {n: p.detach().cpu().double().requires_grad_(True) for n, p in torch.nn.Module.named_parameters(model)}
    yo = R.uno_forward(x.cpu().double(), P, outs, modes, scal, None, "group_norm", domain_padding=0.2)
    lo = ((yo - y.cpu().double()) ** 2).mean()
    names = sorted(P)
    go = dict(zip(names, torch.autograd.grad(lo, [P[n] for n in names])))
    nat = model.native()
    yh = nat.forward(x)
    assert all(e["kept"] != full_fft for e in nat.blk)
    losses, gy = ppsci.loss.MSELoss("mean").value_and_grad(yh, y, "y")
    assert K._rel(yh.cpu().numpy(), yo.detach().numpy()) < 5e-6
    assert abs(float(losses["y"]) / float(lo) - 1.0) < 1e-5
    model.flat_grad.fill_(float("nan"))
    nat.backward(gy)
    for n, p in torch.nn.Module.named_parameters(model):
        assert K._rel(p.grad.cpu().numpy(), go[n].numpy()) < 5e-5, n


@pytest.mark.parametrize("res", [(32, 64), (64, 128)])
def test_sfno_reference_config_full_size(dev, res):
    """The reference's SFNO configuration (conf/sfno_swe_pretrain.yaml: in 3, out 3, hidden 32, projection 64, 4 layers, n_modes
    (32, 32) = 32 degrees x 16 orders, GroupNorm, batch 4) at its training grid 32 x 64 and its second evaluation grid 64 x 128, against
    the fp64 oracle (oracle/ref_torch.sfno_forward, pinned by the reference-run tests/golden/sfno.npz): output, loss, every gradient."""
    if dev != "gpu":
        pytest.skip("full sizes run on the GPU only")
    import ppsci
    from oracle import ref_torch as R

    torch.manual_seed(0)
    model = ppsci.arch.SFNONet(("x",), ("y",), (32, 32), 32, in_channels=3, out_channels=3, lifting_channels=256, projection_channels=64,
                               n_layers=4, norm="group_norm")
    B = 4
    x = torch.as_tensor(np.random.default_rng(42).standard_normal((B, 3) + res).astype(np.float32)).cuda()
    y = torch.as_tensor(np.random.default_rng(43).standard_normal((B, 3) + res).astype(np.float32)).cuda()
    P = {n: p.detach().cpu().double().requires_grad_(True) for n, p in torch.nn.Module.named_parameters(model)}
    yo = R.sfno_forward(x.cpu().double(), P, 4, (32, 32), "group_norm")
    lo = ((yo - y.cpu().double()) ** 2).mean()
    names = sorted(P)
    go = dict(zip(names, torch.autograd.grad(lo, [P[n] for n in names])))
    nat = model.native()
    yh = nat.forward(x)
    losses, gy = ppsci.loss.MSELoss("mean").value_and_grad(yh, y, "y")
    assert K._rel(yh.cpu().numpy(), yo.detach().numpy()) < 5e-6
    assert abs(float(losses["y"]) / float(lo) - 1.0) < 1e-5
    model.flat_grad.fill_(float("nan"))
    nat.backward(gy)
    for n, p in torch.nn.Module.named_parameters(model):
        assert K._rel(p.grad.cpu().numpy(), go[n].numpy()) < 5e-5, n


@pytest.mark.parametrize("padding,full_fft", [(0.078125, False), (0.078125, True), (0.1, False)])
def test_tfno_64x64_with_the_yaml_domain_padding(dev, padding, full_fft, monkeypatch):
    """The padding the reference's TFNO yaml names (0.078125: 64 -> 69 x 69 planes, odd: element accesses, the double
    fftshift one row apart) and 0.1 (70 x 70: a multiple of 4 but not of 16) at the BASELINE plane size, batch 4, on the
    kept-mode transforms and on the library FFT: output, loss and every gradient against the fp64 oracle."""
    if dev != "gpu":
        pytest.skip("full plane sizes run on the GPU only")
    import ppsci
    from oracle import ref_torch as R

    monkeypatch.setenv("PPSCI_FNO_FULL_FFT", "1" if full_fft else "0")
    torch.manual_seed(1)
    model = ppsci.arch.TFNO2dNet(("x",), ("y",), 12, 12, hidden_channels=32, in_channels=3, out_channels=1,
                                 lifting_channels=256, projection_channels=64, n_layers=4, norm="group_norm",
                                 domain_padding=padding)
    x = torch.as_tensor(np.random.default_rng(2).standard_normal((4, 3, 64, 64)).astype(np.float32)).cuda()
    y = torch.as_tensor(np.random.default_rng(3).standard_normal((4, 1, 64, 64)).astype(np.float32)).cuda()
    P = {n: p.detach().cpu().double().requires_grad_(True) for n, p in torch.nn.Module.named_parameters(model)}
    yo = R.fno_forward(x.cpu().double(), P, 4, (12, 12), "group_norm", domain_padding=padding)
    lo = ((yo - y.cpu().double()) ** 2).mean()
    names = sorted(P)
    go = dict(zip(names, torch.autograd.grad(lo, [P[n] for n in names])))
    nat = model.native()
    yh = nat.forward(x)
    assert nat.kept == (not full_fft) and nat.hw == ((69, 69) if padding < 0.09 else (70, 70))
    losses, gy = ppsci.loss.MSELoss("mean").value_and_grad(yh, y, "y")
    assert K._rel(yh.cpu().numpy(), yo.detach().numpy()) < 2e-6
    assert abs(float(losses["y"]) / float(lo) - 1.0) < 1e-5
    model.flat_grad.fill_(float("nan"))
    nat.backward(gy)
    for n, p in torch.nn.Module.named_parameters(model):
        assert K._rel(p.grad.cpu().numpy(), go[n].numpy()) < 2e-5, n

