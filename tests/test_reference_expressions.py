"""the reference's examples that call `ppsci.autodiff.jacobian / hessian` DIRECTLY inside output expressions,
equation functions or model transforms (15 example scripts), restated here expression by expression -- same operations, own
words, file:line of the original next to each -- and pushed through this framework's tracer and lowering
(`compile.trace_exprs` -> `graph.lower`: what a constraint does at construction).  Every case either LOWERS to the stream set +
residual program of the fused kernels or raises with a reason; the table below is the inventory (nothing here runs a kernel).

What lowers: arithmetic / elementwise functions of the inputs, the network outputs and their derivatives of orders 0..4 along
the inputs (mixed: u_ab, u_aabb), `paddle.where` on input columns, one-row slices, several networks side by side (ModelList).
What does not (raises NotImplementedError / TypeError with the reason): derivatives of PRODUCTS of network outputs taken as a whole
(`jacobian(rho * u, x)` -- lowers: the product rule is applied symbolically); a network fed by another network's output or
derivatives (deephpms: `model_pde` takes u, u_x, u_xx as inputs).  Since round 6 also lowered: batch-wide reductions
(`u.mean()`, tests/test_batch_reductions.py) and Volterra's quadrature over other points of the batch (graph.couple).
"""
import numpy as np
import pytest

import ppsci
from paddlescience_amd import compile as cp
from paddlescience_amd import device, graph
from paddlescience_amd.compile import LABEL_PREFIX
from ppsci.autodiff import hessian, jacobian


@pytest.fixture(autouse=True)
def _cpu():
    device.set_device("cpu")
    yield
    device.set_device(None)


def _mlp(inputs, outputs):
    return ppsci.arch.MLP(tuple(inputs), tuple(outputs), 2, 16, "tanh")


def _lower(model, exprs, input_keys, label_keys=None, batch=None):
    loss = ppsci.loss.MSELoss("mean")
    outputs = cp.trace_exprs(model, tuple(input_keys), exprs, (), batch, [])
    label_keys = list(label_keys if label_keys is not None else exprs.keys())
    for k in label_keys:
        if k not in outputs:
            outputs[k] = graph.Sym.net(model, model.output_keys.index(k))
    # (a batch-coupled output carries its row mask as the weight column: compile.CompiledConstraint adds it)
    losses = [dict(key=k, label=LABEL_PREFIX + k, weight=(cp.WEIGHT_PREFIX + k) if outputs[k].kind == "couple" else None, area=None,
                   scale=loss.term_scale(k, 100), kind=0, causal=None, periodic=False) for k in label_keys]
    low = graph.lower(outputs, losses, ())
    return low.program.build(), low.streams


F = ppsci.functional if hasattr(ppsci, "functional") else None


def case_quick_start():  # examples/quick_start/case2.py:36
    m = _mlp(("x",), ("u",))
    return m, {"du_dx": lambda out: jacobian(out["u"], out["x"])}, ("x",)


def case_euler_beam():  # examples/euler_beam/euler_beam.py:49-54 (whole-batch forms; the row slices are compile.py's business)
    m = _mlp(("x",), ("u",))
    return m, {"u__x": lambda d: jacobian(d["u"], d["x"]), "u__x__x": lambda d: hessian(d["u"], d["x"]),
               "u__x__x__x": lambda d: jacobian(hessian(d["u"], d["x"]), d["x"])}, ("x",)


def case_darcy():  # examples/darcy/darcy2d.py:113-140: derivatives next to analytic references built from the inputs
    import paddlescience_amd.functional as Fn

    m = _mlp(("x", "y"), ("p",))
    two_pi = float(2 * np.pi)
    return m, {"ux": lambda d: jacobian(d["p"], d["x"]),
               "ux_diff": lambda d: jacobian(d["p"], d["x"]) - two_pi * Fn.cos(two_pi * d["x"]) * Fn.cos(two_pi * d["y"]),
               "uy_diff": lambda d: jacobian(d["p"], d["y"]) - (-two_pi * Fn.sin(two_pi * d["x"]) * Fn.sin(two_pi * d["y"])),
               "p_diff": lambda d: Fn.sin(two_pi * d["x"]) * Fn.cos(two_pi * d["y"]) - d["p"]}, ("x", "y")


def case_chip_heat_interior():  # examples/chip_heat/chip_heat.py:376-378: Laplacian + a source read from the inputs
    m = _mlp(("x", "y", "u_one"), ("T",))
    return m, {"chip": lambda out: hessian(out["T"], out["x"]) + hessian(out["T"], out["y"]) + 100 * out["u_one"]}, ("x", "y", "u_one")


def case_chip_heat_boundary():  # chip_heat.py:217-232: a condition picked per point by an input column (`paddle.where` on bc)
    import paddlescience_amd.functional as Fn

    m = _mlp(("x", "y", "u_one", "bc"), ("T",))

    def chip(out):
        tx = jacobian(out["T"], out["x"])
        robin = tx + out["u_one"] * (out["T"] ** 2 - 1) * (out["T"] ** 2 + 1) * 5.6 / 50000
        return Fn.where(out["bc"] == 1, tx - out["u_one"],
                        Fn.where(out["bc"] == 0, out["T"] - out["u_one"],
                                 Fn.where(out["bc"] == 2, tx + out["u_one"] * (out["T"] - 1), robin)))

    return m, {"chip": chip}, ("x", "y", "u_one", "bc")


def case_biharmonic_bc():  # examples/biharmonic2d/biharmonic2d.py:150-196: moment conditions nu u_xx + u_yy
    m = _mlp(("x", "y"), ("u",))
    nu = 0.3
    return m, {"bc_mx": lambda d: nu * hessian(d["u"], d["x"]) + hessian(d["u"], d["y"]),
               "bc_my": lambda d: hessian(d["u"], d["x"]) + nu * hessian(d["u"], d["y"])}, ("x", "y")


def case_biharmonic_moments():  # biharmonic2d.py:325-336: mixed derivative and third derivatives of the Laplacian
    m = _mlp(("x", "y"), ("u",))
    D, nu = 1.0, 0.3

    def q_x(d):
        w, x, y = d["u"], d["x"], d["y"]
        return -jacobian(hessian(w, x) + hessian(w, y), x) * D

    def m_xy(d):
        return jacobian(jacobian(d["u"], d["x"]), d["y"]) * D * (1 - nu)

    return m, {"Q_x": q_x, "M_xy": m_xy}, ("x", "y")


def case_gpinn():  # examples/gpinn/poisson_1d.py:66-75, :199: output transform x + tanh(x) tanh(pi - x) u, then du/dx
    import paddlescience_amd.functional as Fn

    m = _mlp(("x",), ("u",))
    m.register_output_transform(lambda in_, out: {"u": in_["x"] + Fn.tanh(in_["x"]) * Fn.tanh(float(np.pi) - in_["x"]) * out["u"]})
    return m, {"dudx": lambda out: jacobian(out["u"], out["x"])}, ("x",)


def case_hpinns():  # examples/hpinns/holography.py:79-93: first and second derivatives of two of three networks' outputs
    re, im, eps = _mlp(("x", "y"), ("e_real",)), _mlp(("x", "y"), ("e_imaginary",)), _mlp(("x", "y"), ("epsilon",))
    ml = ppsci.arch.ModelList((re, im, eps))
    return ml, {"de_re_x": lambda out: jacobian(out["e_real"], out["x"]), "de_re_yy": lambda out: hessian(out["e_real"], out["y"]),
                "de_im_y": lambda out: jacobian(out["e_imaginary"], out["y"]), "de_im_xx": lambda out: hessian(out["e_imaginary"], out["x"]),
                "epsilon": lambda out: out["epsilon"]}, ("x", "y")


def case_bubble_poisson():  # examples/bubble/bubble.py:122-127: pressure Poisson residual (the p network of a ModelList)
    psi, p, phil = _mlp(("t", "x", "y"), ("psi",)), _mlp(("t", "x", "y"), ("p",)), _mlp(("t", "x", "y"), ("phil",))
    ml = ppsci.arch.ModelList((psi, p, phil))
    return ml, {"pressure_Poisson": lambda out: hessian(out["p"], out["x"]) + hessian(out["p"], out["y"])}, ("t", "x", "y")


def case_bubble_streamfunction():  # bubble.py:91-101: velocities as derivatives of the stream function, in an OUTPUT TRANSFORM
    psi = _mlp(("t", "x", "y"), ("psi",))
    psi.register_output_transform(lambda in_, out: {"u": jacobian(out["psi"], in_["y"]), "v": -jacobian(out["psi"], in_["x"])})
    return psi, {"u": lambda out: out["u"], "v": lambda out: out["v"]}, ("t", "x", "y")


def case_shock_wave():  # examples/shock_wave/shock_wave.py:45-59: derivatives of PRODUCTS of outputs, |.|, a division
    import paddlescience_amd.functional as Fn

    m = _mlp(("t", "x", "y"), ("u", "v", "p", "rho"))

    def continuity(out):
        t, x, y = out["t"], out["x"], out["y"]
        u, v, rho = out["u"], out["v"], out["rho"]
        div = jacobian(u, x) + jacobian(v, y)
        lam = (0.1 * (Fn.abs(div) - div)) * 0.5 + 1
        return (jacobian(rho, t) + jacobian(rho * u, x) + jacobian(rho * v, y)) / lam

    return m, {"continuity": continuity}, ("t", "x", "y")


def case_volterra():  # examples/ide/volterra_ide.py:48-94: u' + u against a quadrature over OTHER points of the (fixed) batch
    m = _mlp(("x",), ("u",))
    eq = ppsci.equation.Volterra(0.0, 12, 20, lambda x, s: np.exp(s - x), lambda out: jacobian(out["u"], out["x"]) + out["u"])
    x = np.linspace(0.0, 5.0, 12, dtype=np.float32).reshape(-1, 1)  # [points | their quadrature points]: the dataset transform
    batch = {"x": np.concatenate([x, eq.get_quad_points(x).reshape(-1, 1)], axis=0)}
    return m, eq.equations, ("x",), batch


def case_deephpms():  # examples/deephpms/burgers.py:84-99: a second network takes (u, u_x, u_xx) of the first as its INPUTS
    idn = _mlp(("t", "x"), ("u_idn",))
    pde = _mlp(("u_x", "du_x", "du_xx"), ("f_pde",))

    def transform_f(_in):
        u = idn({"t": _in["t"], "x": _in["x"]})["u_idn"]
        return {"u_x": u, "du_x": jacobian(u, _in["x"]), "du_xx": hessian(u, _in["x"])}

    pde.register_input_transform(transform_f)
    ml = ppsci.arch.ModelList((idn, pde))
    return ml, {"du_t": lambda out: jacobian(out["u_idn"], out["t"]), "f_pde": lambda out: out["f_pde"]}, ("t", "x")


CASES = [
    # (example of the reference, builder, expected: "lowers" or the exception type that explains why not)
    ("quick_start/case2.py", case_quick_start, "lowers"),
    ("euler_beam/euler_beam.py", case_euler_beam, "lowers"),
    ("darcy/darcy2d.py", case_darcy, "lowers"),
    ("chip_heat/chip_heat.py (interior)", case_chip_heat_interior, "lowers"),
    ("chip_heat/chip_heat.py (boundary)", case_chip_heat_boundary, "lowers"),
    ("biharmonic2d/biharmonic2d.py (moments)", case_biharmonic_bc, "lowers"),
    ("biharmonic2d/biharmonic2d.py (shear)", case_biharmonic_moments, "lowers"),
    ("gpinn/poisson_1d.py", case_gpinn, "lowers"),
    ("hpinns/holography.py", case_hpinns, "lowers"),
    ("bubble/bubble.py (pressure)", case_bubble_poisson, "lowers"),
    ("bubble/bubble.py (stream function)", case_bubble_streamfunction, "lowers"),
    ("shock_wave/shock_wave.py", case_shock_wave, "lowers"),
    ("ide/volterra_ide.py", case_volterra, "lowers"),  # (a batch-coupled residual: per-point programs around two matrix-vector launches)
    ("deephpms/burgers.py (+ korteweg_de_vries, kuramoto_sivashinsky, navier_stokes, schrodinger: the same structure)", case_deephpms,
     "raises"),
]


def _attempt(builder):
    try:
        model, exprs, keys, *batch = builder()
        ed, streams = _lower(model, exprs, keys, batch=batch[0] if batch else None)
        return "lowers", f"{ed.n_instr} instructions, {ed.n_res} terms, streams n1={len(streams.dirs)} n2={streams.n2}"
    except (NotImplementedError, TypeError, ValueError, KeyError, AttributeError, AssertionError) as e:
        return "raises", f"{type(e).__name__}: {str(e)[:160]}"


@pytest.mark.parametrize("example,builder,expected", CASES, ids=[c[0].split(" ")[0] + ("" if " " not in c[0] else c[0].split(" ")[1]) for c in CASES])
def test_reference_example_expression(example, builder, expected):
    status, detail = _attempt(builder)
    print(f"{example:60s} {status:8s} {detail}")
    assert status == expected, (example, status, detail)


def test_inventory_has_no_unexplained_raises(capsys):
    """The sweep as one table (pytest -s shows it): every example either lowers or raises one of the documented reasons."""
    rows = [(ex,) + _attempt(b) for ex, b, _ in CASES]
    with capsys.disabled():
        for ex, st, det in rows:
            print(f"  f1 sweep | {ex[:58]:58s} | {st:6s} | {det}")
    lowered = sum(1 for _, st, _ in rows if st == "lowers")
    assert lowered == 13
    for ex, st, det in rows:
        if st == "raises":  # the one structural case of the module docstring: nothing else may fail
            assert "input transform" in det, (ex, det)


def test_third_order_mixed_derivative_values(tmp_path):
    """u_xxy and u_xyy by polarisation of third derivatives along x + y, x - y (graph.lower) -- what the shear forces of
    biharmonic2d.py:325-336 need -- and `where` on an input column, through the kernels (CPU SIMT emulator), against torch's
    reverse-over-reverse autograd in float64 on the same weights."""
    import torch

    import paddlescience_amd.functional as Fn
    from paddlescience_amd import _lib
    from tests.emu import build_emu

    build_emu.inject()
    try:
        torch.manual_seed(3)
        m = ppsci.arch.MLP(("x", "y", "bc"), ("u",), 2, 16, "tanh")
        n = 40
        rng = np.random.default_rng(5)
        inp = {"x": rng.uniform(-1, 1, (n, 1)).astype(np.float32), "y": rng.uniform(-1, 1, (n, 1)).astype(np.float32),
               "bc": rng.integers(0, 3, (n, 1)).astype(np.float32)}
        exprs = {"u_xxy": lambda d: jacobian(hessian(d["u"], d["x"]), d["y"]),
                 "u_xyy": lambda d: jacobian(hessian(d["u"], d["y"]), d["x"]),
                 "sel": lambda d: Fn.where(d["bc"] == 1, jacobian(d["u"], d["x"]), Fn.where(d["bc"] > 1.5, d["u"] * d["u"], d["y"]))}
        solver = ppsci.solver.Solver(m, None, str(tmp_path))
        got = solver.predict(inp, exprs, batch_size=None, return_numpy=True)
        # float64 restatement: y = tanh(.. tanh(x W0 + b0) ..) W_last + b_last on the model's own parameters
        ps = [p.detach().double().cpu() for p in m.parameters()]
        X = {k: torch.tensor(v.astype(np.float64), requires_grad=True) for k, v in inp.items()}
        h = torch.cat([X["x"], X["y"], X["bc"]], 1)
        for i in range(0, len(ps) - 2, 2):
            h = torch.tanh(h @ ps[i] + ps[i + 1])
        u = h @ ps[-2] + ps[-1]
        g = lambda f, v: torch.autograd.grad(f.sum(), v, create_graph=True)[0]  # noqa: E731
        ux, uy = g(u, X["x"]), g(u, X["y"])
        uxx, uyy = g(ux, X["x"]), g(uy, X["y"])
        ref = {"u_xxy": g(uxx, X["y"]), "u_xyy": g(uyy, X["x"]),
               "sel": torch.where(X["bc"] == 1, ux, torch.where(X["bc"] > 1.5, u * u, X["y"]))}
        for k in exprs:
            a, b = got[k][:, 0].astype(np.float64), ref[k].detach().numpy()[:, 0]
            assert np.linalg.norm(a - b) / np.linalg.norm(b) < 2e-5, (k, np.linalg.norm(a - b) / np.linalg.norm(b))
    finally:
        _lib._inject_for_tests(None)


def test_where_on_a_static_batch_and_on_two_traced_values():
    """with the values of a FIXED batch behind the trace (what the Solver gives a full, unshuffled constraint)
    `d["bc"] == 1` is a bool array for Python -- and must still be a traced condition for `functional.where`; and
    `where(symA == symB, ...)` compares per point instead of silently taking the `y` branch (identity of the two nodes)."""
    import paddlescience_amd.functional as Fn

    m = _mlp(("x", "bc"), ("u",))
    bc = np.array([0, 1, 1, 2, 0, 1], np.float32).reshape(-1, 1)
    batch = {"x": np.linspace(0, 1, 6, dtype=np.float32).reshape(-1, 1), "bc": bc}
    exprs = {"sel": lambda d: Fn.where(d["bc"] == 1, jacobian(d["u"], d["x"]), d["u"]),
             "same": lambda d: Fn.where(d["x"] == d["bc"], d["u"], d["x"])}
    free = cp.trace_exprs(m, ("x", "bc"), exprs, (), None, [])
    fixed = cp.trace_exprs(m, ("x", "bc"), exprs, (), batch, [])
    for k in exprs:
        assert repr(fixed[k]) == repr(free[k]), k  # the same per-point program with or without the batch behind it
        assert "heaviside" in repr(fixed[k])
    # ... and nothing was specialised to the batch by handing the mask to where()
    with graph.batch_values(batch) as tr:
        d = {"x": graph.Sym.input("x"), "bc": graph.Sym.input("bc")}
        mask = d["bc"] == 1
        Fn.where(mask, d["x"], d["bc"])
        assert tr.concretized == []
        assert mask.tolist() == [[False], [True], [True], [False], [False], [True]] or list(np.asarray(mask).ravel()) == [False, True, True, False, False, True]
        assert tr.concretized  # LOOKING at the values does specialise the trace (and is recorded for the rank check)
    # truthiness of sym == sym stays node identity (hash-consing, `in` on lists of nodes)
    a = graph.Sym.input("x")
    assert (a == a) and not (a == graph.Sym.input("bc")) and (a != graph.Sym.input("bc"))
    assert a in [graph.Sym.input("bc"), a]
