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
