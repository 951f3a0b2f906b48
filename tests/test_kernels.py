"""Kernel-level parity: the C ABI (include/ppsci_hip.h) against the fp64 oracle (oracle/taylor_np.py).

Every test runs twice:
  * dev="emu": the REAL kernel source (paddlescience_amd/csrc/*.hip) compiled for the CPU SIMT
    emulator (tests/emu/hip_emu.h) -- how MFMA layouts, LDS staging and the reverse sweep are
    validated in the GPU-less container (`-m "not gpu"`);
  * dev="gpu": the gfx950 build on the MI355X (`-m gpu`), through the same C ABI.
Tolerances are fp32 relative-L2 against the fp64 oracle evaluated on the fp32-rounded weights."""
import numpy as np
import pytest
import torch

from oracle import taylor_np as T
from tests.emu import build_emu


DEVICE = "cpu"


@pytest.fixture(autouse=True, params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def dev(request):
    global DEVICE
    from paddlescience_amd import _lib

    if request.param == "emu":
        build_emu.inject()
        DEVICE = "cpu"
    else:
        _lib._inject_for_tests(None)
        DEVICE = "cuda"
    yield request.param
    _lib._inject_for_tests(None)
    DEVICE = "cpu"


def _t(a, dtype=torch.float32):
    return torch.tensor(np.ascontiguousarray(a), dtype=dtype).to(DEVICE)


def _full(shape, val):
    return torch.full(shape, val, dtype=torch.float32, device=DEVICE)


def _run_fwd(net, X, dirs, n2, stash=False):
    from paddlescience_amd import hotpath as hp

    N = X.shape[0]
    emb = [1 if j in net.periods else 0 for j in range(net.d_raw)]
    om = [net.periods.get(j, 0.0) for j in range(net.d_raw)]
    lay = hp.NetLayout(net.d_raw, net.n_hidden, net.weights[1].shape[0], net.d_out, net.activation,
                       net.skip_connection, emb, om, net.fourier_half)
    spec = hp.StreamSpec([list(map(float, r)) for r in dirs], n2)
    desc = lay.desc(spec)
    params = _t(T.flat_params(net))
    assert params.numel() == lay.n_params
    inputs = [_t(X[:, j]) for j in range(net.d_raw)]
    U = _full((net.d_out * spec.S, N), float("nan"))
    st = None
    if stash:
        st = _full((hp.stash_bytes(desc, N) // 4,), float("nan"))
    hp.taylor_fwd(desc, params, inputs, U, st)
    return desc, params, inputs, U, st


def _rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


@pytest.mark.parametrize(
    "hidden,dout,dirs,n2,act,N",
    [
        ([20, 20, 20], 1, np.eye(2), 2, "tanh", 37),      # Laplace2D cfg1: H=20 padded to 32
        ([32, 32], 1, [[0, 1], [1, 0]], 1, "tanh", 16),   # Allen-Cahn stream set
        ([20, 20], 3, np.eye(2), 2, "silu", 70),          # NS stream set, 3 outputs, >1 block
        ([24, 24, 24], 2, np.eye(2), 0, "sin", 5),
        ([16], 1, np.zeros((0, 2)), 0, "tanh", 33),       # plain forward, single hidden layer
        ([20, 20], 1, np.eye(2), 2, "sigmoid", 21),
        ([20, 20], 1, np.eye(2), 2, "cos", 21),
        ([20, 20], 1, np.eye(2), 2, "gelu", 21),
        ([20, 20, 20], 1, np.eye(2), 2, "siren", 21),
    ],
)
def test_fwd_streams_match_oracle(hidden, dout, dirs, n2, act, N):
    dirs = np.asarray(dirs, dtype=np.float64).reshape(-1, 2)
    net = T.make_net(2, hidden, dout, activation=act, bias_scale=0.2)
    X = np.random.default_rng(7).uniform(-1, 1, (N, 2)).astype(np.float32).astype(np.float64)
    _, _, _, U, _ = _run_fwd(net, X, dirs, n2)
    net32 = net.astype(np.float32).astype(np.float64)
    ref = T.taylor_forward(net32, X, dirs, n2).reshape(-1, N)
    got = U.cpu().numpy().astype(np.float64)
    assert np.isfinite(got).all()
    # siren: sin(30 z) turns the fp32 rounding of z (1e-7 relative) into 30x that much phase error per layer
    tol = 4e-5 if act == "siren" else 2e-6
    for q in range(ref.shape[0]):
        assert _rel(got[q], ref[q]) < tol, (q, _rel(got[q], ref[q]))


@pytest.mark.parametrize("hidden,dout,n2,N,knob", [([100, 100, 100], 3, 2, 37, 8), ([100, 100, 100], 3, 2, 37, 16),
                                                   ([200, 200], 2, 1, 21, 8), ([256, 256, 256], 1, 2, 16, 8),
                                                   ([64, 64, 64, 64], 1, 1, 40, 4), ([50, 50, 50], 3, 2, 37, 4)])
def test_fwd_wide_nets(hidden, dout, n2, N, knob):
    """Feature-split forward kernel (one tile per workgroup, NB/4 waves): width 100 -> NB = 8 with the knob at 8
    (wide) and at 16 (single-wave kernel), widths 200 / 256 -> NB = 16 (wide only)."""
    from paddlescience_amd import _lib

    dirs = np.eye(2)
    net = T.make_net(2, hidden, dout, bias_scale=0.2)
    X = np.random.default_rng(17).uniform(-1, 1, (N, 2)).astype(np.float32).astype(np.float64)
    _lib.lib().ppsci_set_wide_min_nb(knob)
    try:
        _, _, _, U, st = _run_fwd(net, X, dirs, n2, stash=True)
    finally:
        _lib.lib().ppsci_set_wide_min_nb(8)
    ref = T.taylor_forward(net.astype(np.float32).astype(np.float64), X, dirs, n2).reshape(-1, N)
    got = U.cpu().numpy().astype(np.float64)
    assert np.isfinite(got).all() and np.isfinite(st.cpu().numpy()).all()
    for q in range(ref.shape[0]):
        assert _rel(got[q], ref[q]) < 8e-6, (q, _rel(got[q], ref[q]))  # fp32 sums over up to 256 features


def test_fwd_period_embedding_and_skip():
    w = 2 * np.pi / 2.0
    net = T.make_net(2, [20, 20, 20], 1, periods={1: float(np.float32(w))}, skip_connection=True, bias_scale=0.2)
    X = np.random.default_rng(3).uniform(-1, 1, (21, 2)).astype(np.float32).astype(np.float64)
    dirs = np.array([[0.0, 1.0], [1.0, 0.0]])
    _, _, _, U, _ = _run_fwd(net, X, dirs, 1)
    ref = T.taylor_forward(net.astype(np.float32).astype(np.float64), X, dirs, 1).reshape(-1, 21)
    for q in range(ref.shape[0]):
        assert _rel(U.cpu().numpy()[q].astype(np.float64), ref[q]) < 3e-6


def _run_bwd(net, X, dirs, n2, Ubar):
    from paddlescience_amd import hotpath as hp

    desc, params, inputs, U, st = _run_fwd(net, X, dirs, n2, stash=True)
    N = X.shape[0]
    rows = hp.bwd_partial_rows(desc, N)
    P = params.numel()
    partials = _full((rows, P), float("nan"))
    ub = _t(Ubar.reshape(-1, N))
    ws = _full((max(4, hp.bwd_workspace_bytes(desc, N) // 4),), float("nan"))
    hp.taylor_bwd(desc, params, inputs, ub, st, ws, partials)
    grad = _full((P,), float("nan"))
    hp.reduce_rows(partials, rows, P, grad, False)
    return grad.cpu().numpy().astype(np.float64)


@pytest.mark.parametrize(
    "hidden,dout,dirs,n2,act,N",
    [
        ([20, 20, 20], 1, np.eye(2), 2, "tanh", 37),
        ([32, 32], 1, [[0, 1], [1, 0]], 1, "tanh", 16),
        ([20, 20], 3, np.eye(2), 2, "silu", 70),
        ([24, 24, 24], 2, np.eye(2), 0, "sin", 5),
        ([16], 1, np.zeros((0, 2)), 0, "tanh", 33),
        ([20, 20], 1, np.eye(2), 2, "sigmoid", 21),
        ([20, 20], 1, np.eye(2), 2, "cos", 21),
        ([20, 20, 20], 1, np.eye(2), 2, "gelu", 21),
        ([20, 20, 20], 1, np.eye(2), 2, "siren", 21),
        ([40, 40, 40], 1, np.eye(2), 2, "tanh", 19),     # NB = 4 (H=40 padded to 64)
        ([64, 64, 64, 64], 1, [[0, 1], [1, 0]], 1, "tanh", 40),  # bench shape: register-accumulator path
        ([20, 20, 20, 20, 20], 1, np.eye(2), 2, "tanh", 25),     # reference laplace2d.yaml depth (5x20)
    ],
)
def test_bwd_param_grads_match_oracle(hidden, dout, dirs, n2, act, N):
    dirs = np.asarray(dirs, dtype=np.float64).reshape(-1, 2)
    net = T.make_net(2, hidden, dout, activation=act, bias_scale=0.2)
    rng = np.random.default_rng(11)
    X = rng.uniform(-1, 1, (N, 2)).astype(np.float32).astype(np.float64)
    S = 1 + dirs.shape[0] + n2
    Ubar = rng.standard_normal((dout, S, N)).astype(np.float32).astype(np.float64)
    got = _run_bwd(net, X, dirs, n2, Ubar)
    net32 = net.astype(np.float32).astype(np.float64)
    _, cache = T.taylor_forward(net32, X, dirs, n2, keep=True)
    gW, gb = T.taylor_backward(net32, cache, Ubar)
    ref = T.flat_grads(gW, gb)
    assert np.isfinite(got).all()
    amp = 20.0 if act == "siren" else 1.0  # sin(30 z): 30x the phase error per layer (see the forward test)
    assert _rel(got, ref) < 5e-6 * amp, _rel(got, ref)
    # per-tensor check so that a small tensor cannot hide behind a big one
    off = 0
    for w, b in zip(gW, gb):
        for t in (w, b):
            n = t.size
            if np.linalg.norm(t) > 0:
                assert _rel(got[off:off + n], t.ravel()) < 2e-5 * amp
            else:
                assert np.abs(got[off:off + n]).max() < 1e-6
            off += n


def test_bwd_period_embedding_and_skip():
    w = float(np.float32(2 * np.pi / 2.0))
    net = T.make_net(2, [20, 20, 20], 1, periods={1: w}, skip_connection=True, bias_scale=0.2)
    rng = np.random.default_rng(5)
    X = rng.uniform(-1, 1, (21, 2)).astype(np.float32).astype(np.float64)
    dirs = np.array([[0.0, 1.0], [1.0, 0.0]])
    Ubar = rng.standard_normal((1, 4, 21)).astype(np.float32).astype(np.float64)
    got = _run_bwd(net, X, dirs, 1, Ubar)
    net32 = net.astype(np.float32).astype(np.float64)
    _, cache = T.taylor_forward(net32, X, dirs, 1, keep=True)
    ref = T.flat_grads(*T.taylor_backward(net32, cache, Ubar))
    assert _rel(got, ref) < 5e-6


def test_nonresident_wide_net_and_multi_iteration():
    """H=100 -> NB=8: hidden-layer fragments do not fit LDS together (per-layer lock-step staging and
    per-layer flush of the weight-gradient accumulator); grid capped to 1 block so every wave loops
    over several tiles."""
    from paddlescience_amd import _lib

    net = T.make_net(2, [100, 100, 100], 3, bias_scale=0.2)
    rng = np.random.default_rng(2)
    N = 150  # 10 tiles -> 3 block-iterations on a single 4-wave block
    X = rng.uniform(-1, 1, (N, 2)).astype(np.float32).astype(np.float64)
    dirs = np.eye(2)
    Ubar = rng.standard_normal((3, 5, N)).astype(np.float32).astype(np.float64)
    _lib.lib().ppsci_set_max_grid(1)
    try:
        got = _run_bwd(net, X, dirs, 2, Ubar)
        _, _, _, U, _ = _run_fwd(net, X, dirs, 2)
    finally:
        _lib.lib().ppsci_set_max_grid(0)
    net32 = net.astype(np.float32).astype(np.float64)
    Uref, cache = T.taylor_forward(net32, X, dirs, 2, keep=True)
    assert _rel(U.cpu().numpy().astype(np.float64), Uref.reshape(-1, N)) < 2e-6
    ref = T.flat_grads(*T.taylor_backward(net32, cache, Ubar))
    assert _rel(got, ref) < 5e-6


def test_ns_config_shape_5x128_three_outputs():
    """BASELINE config 3 shape (2 -> 128 x 5 -> 3, S = 5): LDS budget of the lock-step mode and NB = 8 kernels."""
    net = T.make_net(2, [128] * 5, 3, bias_scale=0.1)
    rng = np.random.default_rng(8)
    N = 40
    X = rng.uniform(-0.05, 0.05, (N, 2)).astype(np.float32).astype(np.float64)
    Ubar = rng.standard_normal((3, 5, N)).astype(np.float32).astype(np.float64)
    got = _run_bwd(net, X, np.eye(2), 2, Ubar)
    net32 = net.astype(np.float32).astype(np.float64)
    _, cache = T.taylor_forward(net32, X, np.eye(2), 2, keep=True)
    ref = T.flat_grads(*T.taylor_backward(net32, cache, Ubar))
    assert _rel(got, ref) < 5e-6


@pytest.mark.parametrize("hidden,dout,dirs,n2,N,knob", [
    ([100, 100, 100], 3, np.eye(2), 2, 37, 8),        # NB = 8, two waves per tile
    ([128] * 5, 3, np.eye(2), 2, 40, 8),              # BASELINE config 3 shape
    ([200, 200], 2, [[0, 1], [1, 0]], 1, 21, 8),      # NB = 16 (padded 256), four waves per tile
    ([256, 256, 256, 256], 1, [[0, 1], [1, 0]], 1, 33, 8),  # reference allen_cahn.yaml width
    ([72], 1, np.eye(2), 2, 19, 8),                   # single hidden layer: no hidden-to-hidden weights
    ([64, 64, 64, 64], 1, [[0, 1], [1, 0]], 1, 70, 4),  # width <= 64 on the feature-split XDL kernels (knob 4): bench net
    ([50, 50, 50], 3, np.eye(2), 2, 37, 4),
])
def test_bwd_wide_nets(hidden, dout, dirs, n2, N, knob):
    """Feature-split reverse kernel against the fp64 oracle; one block only so that several tiles share it."""
    from paddlescience_amd import _lib

    dirs = np.asarray(dirs, dtype=np.float64).reshape(-1, 2)
    net = T.make_net(2, hidden, dout, bias_scale=0.1)
    rng = np.random.default_rng(23)
    X = rng.uniform(-1, 1, (N, 2)).astype(np.float32).astype(np.float64)
    S = 1 + dirs.shape[0] + n2
    Ubar = rng.standard_normal((dout, S, N)).astype(np.float32).astype(np.float64)
    _lib.lib().ppsci_set_wide_min_nb(knob)
    _lib.lib().ppsci_set_max_grid(2)
    try:
        got = _run_bwd(net, X, dirs, n2, Ubar)
    finally:
        _lib.lib().ppsci_set_wide_min_nb(8)
        _lib.lib().ppsci_set_max_grid(0)
    net32 = net.astype(np.float32).astype(np.float64)
    _, cache = T.taylor_forward(net32, X, dirs, n2, keep=True)
    gW, gb = T.taylor_backward(net32, cache, Ubar)
    ref = T.flat_grads(gW, gb)
    assert np.isfinite(got).all()
    assert _rel(got, ref) < 1e-5, _rel(got, ref)
    off = 0
    for w, b in zip(gW, gb):
        for t in (w, b):
            n = t.size
            if np.linalg.norm(t) > 0:
                assert _rel(got[off:off + n], t.ravel()) < 3e-5
            off += n


@pytest.mark.parametrize("hidden,dout,dirs,n2,act,N", [
    ([256, 256, 256, 256], 1, [[0, 1], [1, 0]], 1, "tanh", 70),   # the reference allen_cahn.yaml's width, S = 4: three launches
    ([200, 200], 2, [[0, 1], [1, 0]], 1, "silu", 37),               # one hidden matrix: the single launch is top AND bottom
    ([256, 256, 256], 1, [[1, 0]], 1, "sin", 21),                   # S = 3: the odd stream's K = 16 step
])
def test_bwd_layerwise_width_256(hidden, dout, dirs, n2, act, N):
    """Padded width 256: the layer-by-layer XDL reverse kernel (csrc/taylor_bwd_lw.inc: one launch per hidden matrix, the adjoint
    handed on through the workspace) against the fp64 oracle and against the round-2 fp32-MFMA kernel it replaces; several
    tiles per workgroup (max_grid 2)."""
    from paddlescience_amd import _lib
    from paddlescience_amd import hotpath as hp

    dirs = np.asarray(dirs, dtype=np.float64).reshape(-1, 2)
    net = T.make_net(2, hidden, dout, activation=act, bias_scale=0.1)
    rng = np.random.default_rng(29)
    X = rng.uniform(-1, 1, (N, 2)).astype(np.float32).astype(np.float64)
    S = 1 + dirs.shape[0] + n2
    Ubar = rng.standard_normal((dout, S, N)).astype(np.float32).astype(np.float64)
    lib = _lib.lib()
    lib.ppsci_set_max_grid(2)
    try:
        desc = _run_fwd(net, X, dirs, n2, stash=True)[0]
        lib.ppsci_set_bwd_layerwise(1)
        ws_lw = hp.bwd_workspace_bytes(desc, N)
        got = _run_bwd(net, X, dirs, n2, Ubar)
        lib.ppsci_set_bwd_layerwise(0)
        ws_old = hp.bwd_workspace_bytes(desc, N)
        old = _run_bwd(net, X, dirs, n2, Ubar)
    finally:
        lib.ppsci_set_bwd_layerwise(1)
        lib.ppsci_set_max_grid(0)
    assert ws_lw != ws_old  # (the two kernels lay their workspace out differently: really two code paths)
    net32 = net.astype(np.float32).astype(np.float64)
    _, cache = T.taylor_forward(net32, X, dirs, n2, keep=True)
    gW, gb = T.taylor_backward(net32, cache, Ubar)
    ref = T.flat_grads(gW, gb)
    assert np.isfinite(got).all()
    assert _rel(got, ref) < 1e-5 and _rel(old, ref) < 1e-5, (_rel(got, ref), _rel(old, ref))
    off = 0
    for w, b in zip(gW, gb):
        for t in (w, b):
            n = t.size
            if np.linalg.norm(t) > 0:
                assert _rel(got[off:off + n], t.ravel()) < 3e-5
            off += n


def test_resident_multi_block_multi_iteration():
    from paddlescience_amd import _lib

    net = T.make_net(2, [20, 20, 20], 1, bias_scale=0.2)
    rng = np.random.default_rng(4)
    N = 300  # 19 tiles, 2 blocks -> 3 iterations
    X = rng.uniform(-1, 1, (N, 2)).astype(np.float32).astype(np.float64)
    Ubar = rng.standard_normal((1, 5, N)).astype(np.float32).astype(np.float64)
    _lib.lib().ppsci_set_max_grid(2)
    try:
        got = _run_bwd(net, X, np.eye(2), 2, Ubar)
    finally:
        _lib.lib().ppsci_set_max_grid(0)
    net32 = net.astype(np.float32).astype(np.float64)
    _, cache = T.taylor_forward(net32, X, np.eye(2), 2, keep=True)
    ref = T.flat_grads(*T.taylor_backward(net32, cache, Ubar))
    assert _rel(got, ref) < 5e-6


def test_epilogue_allen_cahn_program_and_adjoint():
    from paddlescience_amd import _lib as L
    from paddlescience_amd import hotpath as hp

    rng = np.random.default_rng(9)
    N = 700  # 3 blocks
    U = rng.standard_normal((4, N)).astype(np.float32)
    w = rng.uniform(0.5, 1.5, N).astype(np.float32)
    lab = rng.standard_normal(N).astype(np.float32) * 0.1
    # streams (u, u_x, u_t, u_xx); residual = u_t - eps^2*u_xx + 5*u*u*u - 5*u  (allen_cahn.py:62)
    pr = hp.Program(4, 2)
    u, ut, uxx = pr.ld_u(0), pr.ld_u(2), pr.ld_u(3)
    t1 = pr.op(L.OP_MUL, pr.const(0.01**2), uxx)
    t2 = pr.op(L.OP_SUB, ut, t1)
    five = pr.const(5.0)
    t3 = pr.op(L.OP_MUL, pr.op(L.OP_MUL, pr.op(L.OP_MUL, five, u), u), u)
    t4 = pr.op(L.OP_ADD, t2, t3)
    r = pr.op(L.OP_SUB, t4, pr.op(L.OP_MUL, five, u))
    scale = 1.0 / N
    pr.residual(r, label=0, weight=1, scale=scale)
    e = pr.build()
    rows = hp.epilogue_partial_rows(N)
    tU = _t(U)
    aux = [_t(lab), _t(w)]
    resid = _full((1, N), 0.0)
    Ubar = _full((4, N), 0.0)
    part = _full((rows, 1), 0.0)
    xs = [_full((N,), 0.0), _full((N,), 0.0)]
    hp.epilogue(e, N, xs, tU, aux, resid, Ubar, part)
    U64 = U.astype(np.float64)
    eps2 = float(np.float32(0.01**2))
    rr = U64[2] - eps2 * U64[3] + 5 * U64[0] ** 3 - 5 * U64[0]
    np.testing.assert_allclose(resid.cpu().numpy()[0], rr, rtol=2e-6, atol=2e-6)
    loss = (scale * w * (rr - lab) ** 2).sum()
    out = _full((1,), 0.0)
    hp.reduce_rows(part, rows, 1, out, False)
    assert float(out[0]) == pytest.approx(loss, rel=1e-5)
    seed = 2 * scale * w * (rr - lab)
    ref = np.zeros((4, N))
    ref[0] = seed * (15 * U64[0] ** 2 - 5)
    ref[2] = seed
    ref[3] = -eps2 * seed
    np.testing.assert_allclose(Ubar.cpu().numpy(), ref, rtol=2e-5, atol=1e-7)


def test_epilogue_detach_blocks_adjoint_and_misc_ops():
    from paddlescience_amd import _lib as L
    from paddlescience_amd import hotpath as hp

    rng = np.random.default_rng(1)
    N = 65
    U = rng.uniform(0.5, 1.5, (2, N)).astype(np.float32)
    x = rng.uniform(0.1, 1.0, N).astype(np.float32)
    pr = hp.Program(2, 1)
    a, b, xi = pr.ld_u(0), pr.ld_u(1), pr.ld_in(0)
    # r = detach(a) * b + sin(x) * a**2 / b
    t = pr.op(L.OP_MUL, pr.op(L.OP_DETACH, a), b)
    q = pr.op(L.OP_DIV, pr.op(L.OP_MUL, pr.op(L.OP_SIN, xi), pr.op(L.OP_POW, a, pr.const(2.0))), b)
    r = pr.op(L.OP_ADD, t, q)
    pr.residual(r, scale=1.0)
    e = pr.build()
    rows = hp.epilogue_partial_rows(N)
    Ubar = _full((2, N), 0.0)
    part = _full((rows, 1), 0.0)
    hp.epilogue(e, N, [_t(x)], _t(U), [], None, Ubar, part)
    a64, b64, x64 = U[0].astype(np.float64), U[1].astype(np.float64), x.astype(np.float64)
    rr = a64 * b64 + np.sin(x64) * a64**2 / b64
    seed = 2 * rr
    np.testing.assert_allclose(Ubar.cpu().numpy()[0], seed * (np.sin(x64) * 2 * a64 / b64), rtol=3e-5)
    np.testing.assert_allclose(Ubar.cpu().numpy()[1], seed * (a64 - np.sin(x64) * a64**2 / b64**2), rtol=3e-5, atol=1e-6)


def test_epilogue_remaining_sympy_map_ops():
    """asin .. floor of SYMPY_TO_PADDLE (symbolic.py:79-108): values and adjoints against torch fp64 autograd."""
    import torch

    from paddlescience_amd import _lib as L
    from paddlescience_amd import hotpath as hp

    rng = np.random.default_rng(4)
    N = 300
    U = np.stack([rng.uniform(-0.8, 0.8, N), rng.uniform(1.2, 3.0, N)]).astype(np.float32)
    pr = hp.Program(2, 0)
    a, b = pr.ld_u(0), pr.ld_u(1)  # a in (-0.8, 0.8), b in (1.2, 3)
    terms = [pr.op(L.OP_ASIN, a), pr.op(L.OP_ACOS, a), pr.op(L.OP_ATAN, b), pr.op(L.OP_ATAN2, a, b),
             pr.op(L.OP_ASINH, b), pr.op(L.OP_ACOSH, b), pr.op(L.OP_ATANH, a), pr.op(L.OP_ERF, a),
             pr.op(L.OP_LGAMMA, b), pr.op(L.OP_MUL, pr.op(L.OP_CEIL, b), a), pr.op(L.OP_MUL, pr.op(L.OP_FLOOR, b), a),
             pr.op(L.OP_LGAMMA, pr.op(L.OP_SUB, a, pr.const(1.5)))]  # negative argument: reflection branch
    coef = [1.0, 0.5, -0.7, 1.3, 0.9, -1.1, 0.6, 1.7, 0.8, 0.25, -0.35, 0.45]
    r = None
    for c, t in zip(coef, terms):
        ct = pr.op(L.OP_MUL, pr.const(c), t)
        r = ct if r is None else pr.op(L.OP_ADD, r, ct)
    pr.residual(r, scale=1.0)
    e = pr.build()
    rows = hp.epilogue_partial_rows(N)
    resid = _full((1, N), 0.0)
    Ubar = _full((2, N), 0.0)
    part = _full((rows, 1), 0.0)
    hp.epilogue(e, N, [], _t(U), [], resid, Ubar, part)
    ta = torch.tensor(U[0].astype(np.float64), requires_grad=True)
    tb = torch.tensor(U[1].astype(np.float64), requires_grad=True)
    fs = [torch.asin(ta), torch.acos(ta), torch.atan(tb), torch.atan2(ta, tb), torch.asinh(tb), torch.acosh(tb),
          torch.atanh(ta), torch.erf(ta), torch.lgamma(tb), torch.ceil(tb) * ta, torch.floor(tb) * ta,
          torch.lgamma(ta - 1.5)]
    rr = sum(float(np.float32(c)) * f for c, f in zip(coef, fs))
    (rr**2).sum().backward()
    np.testing.assert_allclose(resid.cpu().numpy()[0], rr.detach().numpy(), rtol=5e-6, atol=5e-6)
    np.testing.assert_allclose(Ubar.cpu().numpy()[0], ta.grad.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(Ubar.cpu().numpy()[1], tb.grad.numpy(), rtol=1e-4, atol=1e-4)


def _fourier_net(width, n_linears, dout, periods=None, skip=False, seed=5):
    """Kernel-layout net behind a FourierEmbedding of dimension `width` (mlp.py:117-136, :233-237): layer 0 is
    [B, B] with zero bias, B ~ N(0, 1)."""
    net = T.make_net(2, [width] * (n_linears + 1), dout, periods=periods, skip_connection=skip, bias_scale=0.2)
    d0 = net.weights[0].shape[0]
    B = np.random.default_rng(seed).normal(0.0, 1.0, (d0, width // 2))
    net.weights[0] = np.concatenate([B, B], axis=1)
    net.biases[0] = np.zeros(width)
    net.fourier_half = width // 2
    return net


@pytest.mark.parametrize("width,nl,dout,n2,N,periods,skip", [(32, 2, 1, 2, 37, None, False), (64, 3, 2, 1, 21, {1: 3.0}, True),
                                                             (128, 2, 1, 2, 19, None, False), (256, 2, 1, 1, 16, {1: 3.0}, False)])
def test_fourier_embedding_layer_fwd_and_bwd(width, nl, dout, n2, N, periods, skip):
    """FourierEmbedding run as hidden layer 0 (cos | sin by feature index) in the single-wave (NB = 2, 4) and
    the feature-split (NB = 8, 16) kernels; the skip quirk counts self.linears, not the embedding layer."""
    net = _fourier_net(width, nl, dout, periods, skip)
    rng = np.random.default_rng(23)
    X = rng.uniform(-1, 1, (N, 2)).astype(np.float32).astype(np.float64)
    dirs = np.eye(2)
    net32 = net.astype(np.float32).astype(np.float64)
    _, _, _, U, _ = _run_fwd(net, X, dirs, n2)
    ref, cache = T.taylor_forward(net32, X, dirs, n2, keep=True)
    got = U.cpu().numpy().astype(np.float64)
    for q in range(got.shape[0]):
        assert _rel(got[q], ref.reshape(-1, N)[q]) < 8e-6, (q, _rel(got[q], ref.reshape(-1, N)[q]))
    S = 1 + 2 + n2
    Ubar = rng.standard_normal((dout, S, N)).astype(np.float32).astype(np.float64)
    g = _run_bwd(net, X, dirs, n2, Ubar)
    gW, gb = T.taylor_backward(net32, cache, Ubar)
    assert _rel(g, T.flat_grads(gW, gb)) < 1e-5


def test_linear_materialize_and_pullback_kinds():
    """csrc/reparam.hip against torch autograd of the reference formulas (mlp.py:50-54, :91-92, :128-136)."""
    import torch

    from paddlescience_amd import _lib as L
    from paddlescience_amd import hotpath as hp

    rng = np.random.default_rng(2)
    fin, fout = 37, 48
    v = rng.standard_normal((fin, fout)).astype(np.float32)
    g = rng.uniform(0.5, 2.0, fout).astype(np.float32)
    b = rng.standard_normal(fout).astype(np.float32)
    gW = rng.standard_normal((fin, fout)).astype(np.float32)
    gb = rng.standard_normal(fout).astype(np.float32)
    for kind in (L.LINEAR_PLAIN, L.LINEAR_WEIGHT_NORM, L.LINEAR_RWF, L.LINEAR_FOURIER):
        vv = v[:, : fout // 2].copy() if kind == L.LINEAR_FOURIER else v
        tv = torch.tensor(vv.astype(np.float64), requires_grad=True)
        tg = torch.tensor(g.astype(np.float64), requires_grad=True)
        if kind == L.LINEAR_PLAIN:
            Wref = tv
        elif kind == L.LINEAR_WEIGHT_NORM:
            Wref = tg * tv / torch.linalg.vector_norm(tv, ord=2, dim=0, keepdim=True)
        elif kind == L.LINEAR_RWF:
            Wref = tg * tv
        else:
            Wref = torch.cat([tv, tv], dim=1)
        (Wref * torch.tensor(gW.astype(np.float64))).sum().backward()
        has_g = kind in (L.LINEAR_WEIGHT_NORM, L.LINEAR_RWF)
        W, bo = _full((fin, fout), float("nan")), _full((fout,), float("nan"))
        dv, dg = _t(vv), _t(g) if has_g else None
        hp.linear_materialize(kind, fin, fout, dv, dg, None if kind == L.LINEAR_FOURIER else _t(b), W, bo)
        np.testing.assert_allclose(W.cpu().numpy(), Wref.detach().numpy(), rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(bo.cpu().numpy(), np.zeros(fout) if kind == L.LINEAR_FOURIER else b)
        gv = _full(vv.shape, float("nan"))
        gg = _full((fout,), float("nan")) if has_g else None
        gbo = _full((fout,), float("nan")) if kind != L.LINEAR_FOURIER else None
        hp.linear_pullback(kind, fin, fout, dv, dg, _t(gW), _t(gb), gv, gg, gbo)
        np.testing.assert_allclose(gv.cpu().numpy(), tv.grad.numpy(), rtol=2e-5, atol=2e-6)
        if has_g:
            np.testing.assert_allclose(gg.cpu().numpy(), tg.grad.numpy(), rtol=2e-5, atol=2e-6)
        if gbo is not None:
            np.testing.assert_allclose(gbo.cpu().numpy(), gb)


@pytest.mark.parametrize("act,hidden,dout,n2,N", [("swish", [20, 20, 20], 1, 2, 37), ("stan", [24, 24], 2, 1, 21),
                                                  ("swish", [40, 40], 1, 2, 19), ("stan", [100, 100], 1, 1, 16)])
def test_learnable_activations_fwd_and_bwd(act, hidden, dout, n2, N):
    """Swish x*sigmoid(p x) / Stan tanh(x)(1 + p x) with a trainable per-feature vector p per hidden layer (kernel
    layout: behind the last bias): streams, weight / bias gradients and dL/dp against the numpy oracle.  Width 100 runs
    on the single-wave NB = 8 kernels (the feature-split ones do not carry the parameter)."""
    net = T.make_net(2, hidden, dout, activation=act, bias_scale=0.2)
    rng = np.random.default_rng(29)
    net.act_params = [rng.uniform(0.6, 1.4, hidden[0]) for _ in hidden]
    X = rng.uniform(-1, 1, (N, 2)).astype(np.float32).astype(np.float64)
    dirs = np.eye(2)
    net32 = net.astype(np.float32).astype(np.float64)
    _, _, _, U, _ = _run_fwd(net, X, dirs, n2)
    ref, cache = T.taylor_forward(net32, X, dirs, n2, keep=True)
    got = U.cpu().numpy().astype(np.float64)
    for q in range(got.shape[0]):
        assert _rel(got[q], ref.reshape(-1, N)[q]) < 8e-6, (q, _rel(got[q], ref.reshape(-1, N)[q]))
    S = 1 + 2 + n2
    Ubar = rng.standard_normal((dout, S, N)).astype(np.float32).astype(np.float64)
    g = _run_bwd(net, X, dirs, n2, Ubar)
    gW, gb = T.taylor_backward(net32, cache, Ubar)
    gref = T.flat_grads(gW, gb, cache["gP"])
    assert g.size == gref.size
    nP = sum(p.size for p in net.act_params)
    assert _rel(g[:-nP], gref[:-nP]) < 1e-5
    assert _rel(g[-nP:], gref[-nP:]) < 1e-5, _rel(g[-nP:], gref[-nP:])


@pytest.mark.parametrize("width,N", [(128, 21), (256, 16)])
def test_wide_nets_with_three_second_order_streams(width, N):
    """(3, 3) stream set (unsteady 2-D Navier-Stokes, 3-D Laplace) on the feature-split kernels: S = 7."""
    dirs = np.eye(3)
    net = T.make_net(3, [width, width], 2, bias_scale=0.2)
    rng = np.random.default_rng(41)
    X = rng.uniform(-1, 1, (N, 3)).astype(np.float32).astype(np.float64)
    emb = [0, 0, 0]
    from paddlescience_amd import hotpath as hp

    lay = hp.NetLayout(3, 2, width, 2, "tanh", False, emb, [0.0] * 3)
    spec = hp.StreamSpec([list(map(float, r)) for r in dirs], 3)
    desc = lay.desc(spec)
    params = _t(T.flat_params(net))
    inputs = [_t(X[:, j]) for j in range(3)]
    U = _full((2 * spec.S, N), float("nan"))
    st = _full((hp.stash_bytes(desc, N) // 4,), float("nan"))
    hp.taylor_fwd(desc, params, inputs, U, st)
    net32 = net.astype(np.float32).astype(np.float64)
    ref, cache = T.taylor_forward(net32, X, dirs, 3, keep=True)
    got = U.cpu().numpy().astype(np.float64)
    for q in range(got.shape[0]):
        assert _rel(got[q], ref.reshape(-1, N)[q]) < 8e-6
    Ubar = rng.standard_normal((2, spec.S, N)).astype(np.float32).astype(np.float64)
    rows = hp.bwd_partial_rows(desc, N)
    P = params.numel()
    partials = _full((rows, P), float("nan"))
    ws = _full((max(4, hp.bwd_workspace_bytes(desc, N) // 4),), float("nan"))
    hp.taylor_bwd(desc, params, inputs, _t(Ubar.reshape(-1, N)), st, ws, partials)
    grad = _full((P,), float("nan"))
    hp.reduce_rows(partials, rows, P, grad, False)
    gW, gb = T.taylor_backward(net32, cache, Ubar)
    assert _rel(grad.cpu().numpy().astype(np.float64), T.flat_grads(gW, gb)) < 1e-5


def test_adam_step_matches_oracle():
    from oracle import ref_torch as R
    from paddlescience_amd import hotpath as hp

    rng = np.random.default_rng(0)
    n = 777
    p0 = rng.standard_normal(n).astype(np.float32)
    p = _t(p0)
    m = _full((n,), 0.0)
    v = _full((n,), 0.0)
    ref = R.Adam(n, 1e-3)
    q = p0.astype(np.float64)
    for t in range(1, 6):
        g = rng.standard_normal(n).astype(np.float32)
        hp.adam_step(p, _t(g), m, v, 1e-3, t)
        q = ref.step(q, g.astype(np.float64))
    np.testing.assert_allclose(p.cpu().numpy(), q, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("hidden,dout,n2,N,grid", [
    ([64, 64, 64, 64], 1, 1, 16 * 11 + 3, 2),   # NB = 4: 12 tiles on 2 workgroups -> 2 rounds, the second one ragged
    ([20, 20, 20], 1, 2, 16 * 9 + 1, 1),        # NB = 2: two accumulator copies (waves 0-1 / 2-3), 3 rounds
    ([40, 40], 3, 2, 70, 0),                    # NB = 4, one hidden matrix, three outputs, automatic grid
])
def test_bwd_workgroup_accumulation_matches_streaming_and_oracle(hidden, dout, n2, N, grid):
    """ppsci_taylor_bwd accumulates the hidden-weight gradient per workgroup in LDS (rotated feature blocks, slot
    barriers) where the net allows it; `ppsci_set_bwd_accum(0)` forces the per-tile streaming path of wider nets.
    Both must reproduce the oracle, on rounds in which some waves of a workgroup have no tile as well."""
    from paddlescience_amd import _lib

    dirs = np.eye(2) if n2 == 2 else np.array([[0.0, 1.0], [1.0, 0.0]])
    net = T.make_net(2, hidden, dout, bias_scale=0.2)
    rng = np.random.default_rng(11)
    X = rng.uniform(-1, 1, (N, 2)).astype(np.float32).astype(np.float64)
    Ubar = rng.standard_normal((dout, 1 + 2 + n2, N)).astype(np.float32).astype(np.float64)
    net32 = net.astype(np.float32).astype(np.float64)
    _, cache = T.taylor_forward(net32, X, dirs, n2, keep=True)
    ref = T.flat_grads(*T.taylor_backward(net32, cache, Ubar))
    got = {}
    _lib.lib().ppsci_set_max_grid(grid)
    _lib.lib().ppsci_set_wide_min_nb(16)  # the single-wave kernels (by default width 33..64 runs feature-split)
    try:
        for mode in (1, 0):
            _lib.lib().ppsci_set_bwd_accum(mode)
            got[mode] = _run_bwd(net, X, dirs, n2, Ubar)
    finally:
        _lib.lib().ppsci_set_bwd_accum(1)
        _lib.lib().ppsci_set_max_grid(0)
        _lib.lib().ppsci_set_wide_min_nb(8)
    assert _rel(got[1], ref) < 5e-6 and _rel(got[0], ref) < 5e-6
    assert _rel(got[1], got[0]) < 2e-6


@pytest.mark.parametrize("hidden,dout,n2,N", [([64, 64, 64, 64], 1, 1, 20_000), ([20, 20, 20], 1, 2, 10_000),
                                              ([128] * 5, 3, 2, 4_000), ([256] * 4, 1, 1, 24), ([256] * 4, 1, 1, 5_000)])
def test_bwd_is_bitwise_reproducible(hidden, dout, n2, N, dev):
    """No float atomics across waves: two launches on the same inputs give bit-identical gradients (on the GPU this
    is also the race detector for the slot-barrier scheme)."""
    if dev != "gpu":
        pytest.skip("run-to-run reproducibility is a property of the hardware scheduling: GPU only")
    dirs = np.eye(2) if n2 == 2 else np.array([[0.0, 1.0], [1.0, 0.0]])
    net = T.make_net(2, hidden, dout, bias_scale=0.2)
    rng = np.random.default_rng(5)
    X = rng.uniform(-1, 1, (N, 2)).astype(np.float32).astype(np.float64)
    Ubar = (rng.standard_normal((dout, 1 + 2 + n2, N)) / N).astype(np.float32).astype(np.float64)
    runs = [_run_bwd(net, X, dirs, n2, Ubar) for _ in range(6)]
    for r in runs[1:]:
        assert np.array_equal(r, runs[0])
    net32 = net.astype(np.float32).astype(np.float64)
    if N <= 5_000:  # and it is the right gradient (the oracle finishes in seconds at this size)
        _, cache = T.taylor_forward(net32, X, dirs, n2, keep=True)
        assert _rel(runs[0], T.flat_grads(*T.taylor_backward(net32, cache, Ubar))) < 2e-5
