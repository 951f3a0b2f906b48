"""The equation classes against tests/golden/equations.json: `sympy.srepr` of every residual expression of the REFERENCE's own
classes (tests/golden/make_equations_golden.py) -- same names, same order, structurally identical expressions (sympy's canonical
argument order makes the lowered program's operation order the reference's too) -- and the new classes of round 6 through the
lowering (stream sets, a per-point program) and, for one of them, through the kernels."""
import json
import os

import numpy as np
import pytest
import sympy as sp

import ppsci
from tests.golden.equations_cases import CASES

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "equations.json")))


@pytest.mark.parametrize("case", sorted(CASES))
def test_equation_expressions_are_the_references(case):
    cls, _, kwargs = CASES[case]
    eq = getattr(ppsci.equation, cls)(**kwargs)
    mine = {name: sp.srepr(expr) for name, expr in eq.equations.items() if isinstance(expr, sp.Basic)}
    assert set(mine) == set(GOLD[case])  # (the json is written with sorted keys: registration order is checked below)
    for name in mine:
        assert mine[name] == GOLD[case][name], (case, name)


def test_registration_order_matches_the_reference_for_elasticity():
    eq = ppsci.equation.LinearElasticity(E=None, nu=None, lambda_=1e4, mu=100, dim=3)
    assert list(eq.equations) == ["stress_disp_xx", "stress_disp_yy", "stress_disp_xy", "stress_disp_zz", "stress_disp_xz",
                                  "stress_disp_yz", "equilibrium_x", "equilibrium_y", "equilibrium_z", "traction_x", "traction_y",
                                  "traction_z"]  # linear_elasticity.py:158-180
    with pytest.raises(ValueError):
        ppsci.equation.NormalDotVec(())


def test_new_equation_classes_lower_to_per_point_programs():
    from paddlescience_amd import compile as cp
    from paddlescience_amd import device, graph
    from paddlescience_amd.compile import LABEL_PREFIX

    device.set_device("cpu")
    try:
        def lower(model, eq, inputs):
            exprs = {k: ppsci.lambdify(v, model) if isinstance(v, sp.Basic) else v for k, v in eq.equations.items()}
            outs = cp.trace_exprs(model, tuple(inputs), exprs, (), None, [])
            loss = ppsci.loss.MSELoss("mean")
            losses = [dict(key=k, label=LABEL_PREFIX + k, weight=None, area=None, scale=loss.term_scale(k, 64), kind=0, causal=None,
                           periodic=False) for k in exprs]
            return graph.lower(outs, losses, ())

        low = lower(ppsci.arch.MLP(("t", "x"), ("Eu", "Ev", "pu", "pv", "eta"), 2, 16), ppsci.equation.NLSMB(0.5, -1, -1, True), ("t", "x"))
        assert low.program.build().n_res == 5 and low.streams.n2 >= 1  # E_tt: a second-order stream along t
        disp = ppsci.arch.MLP(("x", "y", "z"), ("u", "v", "w"), 2, 16)
        stress = ppsci.arch.MLP(("x", "y", "z"), ("sigma_xx", "sigma_yy", "sigma_zz", "sigma_xy", "sigma_xz", "sigma_yz"), 2, 16)
        model = ppsci.arch.ModelList((disp, stress))
        el = ppsci.equation.LinearElasticity(E=None, nu=None, lambda_=1e4, mu=100, dim=3)
        keys = [k for k in el.equations if not k.startswith("traction")]
        el.equations = {k: el.equations[k] for k in keys[:8]}  # (at most eight loss terms per constraint: the interior ones)
        low = lower(model, el, ("x", "y", "z"))
        assert low.program.build().n_res == 8 and len(low.nets) == 2
        hx = ppsci.arch.ModelList((ppsci.arch.MLP(("x", "t", "qm_h"), ("T_h",), 2, 16), ppsci.arch.MLP(("x", "t", "qm_c"), ("T_c",), 2, 16),
                                   ppsci.arch.MLP(("x", "t"), ("T_w",), 2, 16)))
        low = lower(hx, ppsci.equation.HeatExchanger(1.0, 1.0, 1.0, 1.0, 1.0, 1.0), ("x", "t", "qm_h", "qm_c"))
        assert low.program.build().n_res == 3 and len(low.nets) == 3
    finally:
        device.set_device(None)
