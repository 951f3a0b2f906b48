"""One-launch training step (ppsci_taylor_step, csrc/taylor_step.inc) against the separate launches
(ppsci_taylor_fwd -> ppsci_epilogue -> ppsci_taylor_bwd -> reductions -> ppsci_adam_step) on the same buffers: the two
compute the same sums in different orders (tree of workgroup rows vs chunked), so values agree to fp32 rounding.  The
separate path itself is pinned to the reference by the golden tests; with the engine's default (one launch for small
batches) those run through this kernel as well."""
import numpy as np
import pytest
import torch

from paddlescience_amd import _lib as L
from paddlescience_amd import device
from paddlescience_amd import hotpath as hp
from paddlescience_amd.engine import Engine, FusedConstraint
from tests.common import make_dev_fixture, rel

dev = make_dev_fixture()


def _weights(lay, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _, shp in lay.param_shapes():
        fan = shp[0] if len(shp) == 2 else 1
        out.append((rng.standard_normal(int(np.prod(shp))) * (0.6 / np.sqrt(fan) if len(shp) == 2 else 0.1)).astype(np.float32))
    return np.concatenate(out)


def _program(kind, n):
    if kind == "allen_cahn":  # streams u, u_x, u_t, u_xx (allen_cahn.py:56-64)
        pr = hp.Program(4, 2)
        u, ut, uxx = pr.ld_u(0), pr.ld_u(2), pr.ld_u(3)
        five = pr.const(5.0)
        r = pr.op(L.OP_SUB, pr.op(L.OP_ADD, pr.op(L.OP_SUB, ut, pr.op(L.OP_MUL, pr.const(1e-4), uxx)),
                                  pr.op(L.OP_MUL, pr.op(L.OP_MUL, pr.op(L.OP_MUL, five, u), u), u)), pr.op(L.OP_MUL, five, u))
        pr.residual(r, scale=1.0 / n)
        return pr.build(), hp.StreamSpec([[0.0, 1.0], [1.0, 0.0]], 1), 0
    if kind == "laplace":  # u, u_x, u_y, u_xx, u_yy; label + weight arrays (laplace.py:40-55, mse.py:82-105)
        pr = hp.Program(5, 2)
        r = pr.op(L.OP_ADD, pr.ld_u(3), pr.ld_u(4))
        pr.residual(r, label=0, weight=1, scale=1.0)
        pr.residual(pr.op(L.OP_MUL, pr.ld_u(0), pr.ld_in(1)), scale=0.5)  # a second term that reads an input
        return pr.build(), hp.StreamSpec([[1.0, 0.0], [0.0, 1.0]], 2), 2
    if kind == "value":  # boundary-condition style: S = 1
        pr = hp.Program(1, 2)
        pr.residual(pr.ld_u(0), label=0, scale=1.0 / n)
        return pr.build(), hp.StreamSpec([], 0), 1
    if kind.startswith("streams:"):  # every stream of a given set, weighted: "streams:n1,n2[,n3,n4]"
        nn = [int(v) for v in kind.split(":")[1].split(",")] + [0, 0]
        n1, n2, n3, n4 = nn[:4]
        S = 1 + n1 + n2 + n3 + n4
        pr = hp.Program(S, 2)
        acc = pr.op(L.OP_MUL, pr.ld_u(0), pr.ld_u(0))
        for q in range(1, S):
            acc = pr.op(L.OP_ADD, acc, pr.op(L.OP_MUL, pr.const(0.3 + 0.1 * q), pr.ld_u(q)))
        pr.residual(acc, label=0, scale=1.0 / n)
        dirs = [[1.0, 0.0], [0.0, 1.0], [0.6, 0.8]][:n1]
        return pr.build(), hp.StreamSpec(dirs, n2, n3, n4), 1
    raise KeyError(kind)


def _constraint(d, kind, lay, n, seed):
    rng = np.random.default_rng(seed)
    ed, streams, n_aux = _program(kind, n)
    xs = [torch.tensor(rng.random(n, dtype=np.float32) * 2 - 1, device=d) for _ in range(2)]
    aux = [torch.tensor(rng.random(n, dtype=np.float32) + 0.5, device=d) for _ in range(n_aux)]
    keys = [f"k{i}" for i in range(ed.n_res)]
    return FusedConstraint(kind, lay, streams, ed, xs, aux, keys, want_residual=True)


def _run(d, lay, specs, flat, one_launch, steps, max_grid=0, max_constraints=4):
    L.lib().ppsci_set_max_grid(max_grid)
    try:
        params = torch.tensor(flat, device=d)
        eng = Engine(lay, params)
        eng.one_launch = one_launch
        eng.one_launch_max_constraints = max_constraints  # (default 1: several small constraints run as parallel branches)
        csts = [_constraint(d, kind, lay, n, 100 + i) for i, (kind, n) in enumerate(specs)]
        assert eng.one_launch_ready(csts) == (one_launch and len(csts) <= max_constraints)
        grads, losses = [], []
        for _ in range(steps):
            eng.train_step(csts, 1e-2)
            grads.append(eng.grad.detach().cpu().numpy().copy())
            losses.append([c.loss_terms.detach().cpu().numpy().copy() for c in csts])
        resid = [c.resid.detach().cpu().numpy().copy() for c in csts]
        return params.detach().cpu().numpy(), grads, losses, resid
    finally:
        L.lib().ppsci_set_max_grid(0)


CASES = [
    # (activation, hidden layers, width, constraints [(program, points)], max_grid)
    ("tanh", 3, 20, [("laplace", 1000)], 0),                     # one tile short of a full last workgroup
    ("tanh", 3, 20, [("laplace", 700), ("value", 90)], 0),       # two constraints: the second accumulates, then Adam
    ("tanh", 4, 32, [("allen_cahn", 300)], 0),                   # the full padded width
    ("tanh", 2, 32, [("allen_cahn", 1500)], 3),                  # several tile rounds per workgroup
    ("silu", 3, 24, [("allen_cahn", 400)], 0),
    ("sin", 3, 16, [("laplace", 200)], 1),                       # a single workgroup: no tree at all
    # every instantiated stream set (csrc/taylor_step.inc PPSCI_STEP_CASE list)
    ("tanh", 3, 20, [("streams:0,0", 500)], 0),
    ("tanh", 3, 20, [("streams:1,1", 500)], 0),
    ("silu", 3, 20, [("streams:2,0", 500)], 0),
    ("sin", 3, 20, [("streams:2,1", 500)], 0),
    ("silu", 2, 30, [("streams:2,2", 500)], 0),
    ("tanh", 3, 20, [("streams:3,3", 330)], 0),
    ("tanh", 3, 20, [("streams:1,1,1,1", 104)], 0),              # euler_beam: u_xxxx
]


@pytest.mark.parametrize("act,depth,width,specs,max_grid", CASES)
def test_one_launch_matches_separate_launches(dev, act, depth, width, specs, max_grid):
    d = device.get_device()
    lay = hp.NetLayout(2, depth, width, 1, act)
    flat = _weights(lay, 7)
    steps = 3
    if dev != "gpu":  # the emulator runs ~1 k points per second: a quarter of the points (still several workgroups:
        specs = [(k, max(72, n // 4 + 3)) for k, n in specs]  # its 4 "CUs" and fan-in 3 give a multi-level tree), two steps
        steps = 2
    p_sep, g_sep, l_sep, r_sep = _run(d, lay, specs, flat, False, steps, max_grid)
    p_one, g_one, l_one, r_one = _run(d, lay, specs, flat, True, steps, max_grid)
    for s in range(steps):
        # (later steps start from parameters that already differ by rounding)
        assert rel(g_one[s], g_sep[s]) < (3e-6 if s == 0 else 1e-4), (s, rel(g_one[s], g_sep[s]))
        for a, b in zip(l_one[s], l_sep[s]):
            np.testing.assert_allclose(a, b, rtol=2e-5 if s == 0 else 1e-3)
    for a, b in zip(r_one, r_sep):
        assert rel(a, b) < 1e-3
    assert rel(p_one, p_sep) < 1e-4
    # the update is not a no-op, and the reduction tree left its counters at zero (steps 2 and 3 would be wrong otherwise)
    assert rel(p_one, flat) > 1e-4


def test_one_launch_is_deterministic(dev):
    d = device.get_device()
    lay = hp.NetLayout(2, 3, 20, 1, "tanh")
    flat = _weights(lay, 3)
    steps = 2 if dev != "gpu" else 300  # on the GPU also the race detector of the reduction tree (10 201 points: two levels)
    n = 300 if dev != "gpu" else 10_201
    a = _run(d, lay, [("laplace", n)], flat, True, steps)
    b = _run(d, lay, [("laplace", n)], flat, True, steps)
    assert np.array_equal(a[0], b[0]) and all(np.array_equal(x, y) for x, y in zip(a[1], b[1]))
    assert np.isfinite(a[0]).all()


def test_loss_terms_from_the_epilogue_launch_are_reproducible(dev):
    """Separate launches: the loss terms are summed by whichever epilogue workgroup finishes last (ppsci_epilogue_losses),
    in a fixed order -- bit-identical run to run (on the GPU also the race detector of its ticket protocol)."""
    d = device.get_device()
    lay = hp.NetLayout(2, 3, 20, 1, "tanh")
    flat = _weights(lay, 5)
    steps, n = (2, 700) if dev != "gpu" else (200, 40_000)  # 157 epilogue workgroups
    a = _run(d, lay, [("laplace", n)], flat, False, steps)
    b = _run(d, lay, [("laplace", n)], flat, False, steps)
    assert all(np.array_equal(x[0], y[0]) for x, y in zip(a[2], b[2])) and np.array_equal(a[0], b[0])
    assert np.isfinite(a[2][-1][0]).all()


def test_one_launch_unsupported_falls_to_separate_launches(dev):
    """No one-launch kernel (activation without an instantiation, learnable equation parameters): workspace query
    says 0 and the engine keeps the separate launches -- which are the same HIP kernels, not a fallback off the GPU."""
    d = device.get_device()
    ed, streams, _ = _program("allen_cahn", 64)
    assert hp.taylor_step_kind(hp.NetLayout(2, 4, 64, 1, "tanh").desc(streams), ed, 64) == hp.STEP_FUSED_TILE  # padded width 64
    assert hp.taylor_step_workspace_bytes(hp.NetLayout(2, 4, 128, 1, "tanh").desc(streams), ed, 64) == 0  # padded width 128
    lay = hp.NetLayout(2, 3, 20, 1, "gelu")
    assert hp.taylor_step_workspace_bytes(lay.desc(streams), ed, 64) == 0
    eng = Engine(lay, torch.tensor(_weights(lay, 1), device=d))
    cst = _constraint(d, "allen_cahn", lay, 64, 5)
    assert not eng.one_launch_ready([cst])
    eng.train_step([cst], 1e-3)
    assert np.isfinite(cst.loss_terms.cpu().numpy()).all()


def test_one_launch_kernels_as_parallel_branches(dev):
    """Default engine, a step of several small constraints: their one-launch kernels (gradient only) run as parallel
    branches of the captured graph, each into its own gradient row; rows summed in constraint order, Adam behind."""
    if dev != "gpu":
        pytest.skip("streams and graph capture: GPU only")
    d = device.get_device()
    lay = hp.NetLayout(2, 3, 20, 1, "tanh")
    flat = _weights(lay, 11)
    specs = [("laplace", 3000), ("value", 400), ("allen_cahn", 1000)]
    p_sep, g_sep, l_sep, _ = _run(d, lay, specs, flat, False, 6, max_constraints=1)
    p_one, g_one, l_one, _ = _run(d, lay, specs, flat, True, 6, max_constraints=1)  # steps 1 / 2 / 3+: eager, capture, replay
    assert rel(g_one[0], g_sep[0]) < 3e-6 and rel(p_one, p_sep) < 1e-4
    for a, b in zip(l_one[0], l_sep[0]):
        np.testing.assert_allclose(a, b, rtol=2e-5)
    again = _run(d, lay, specs, flat, True, 6, max_constraints=1)
    assert np.array_equal(again[0], p_one)
