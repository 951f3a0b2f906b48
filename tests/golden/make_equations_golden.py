"""Generates tests/golden/equations.json by INSTANTIATING the reference's own equation classes
(/root/reference/ppsci/equation/pde/*.py, executed here under the torch-backed paddle shim of tests/golden/_paddle_shim.py)
and storing, per case and equation name, `sympy.srepr` of the residual expression -- after `_apply_detach`, i.e. exactly what
the reference's `lambdify` would be handed.  tests/test_equations.py holds this package's classes to these strings.

    python tests/golden/make_equations_golden.py"""
import importlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import _paddle_shim as S  # noqa: E402

from tests.golden.equations_cases import CASES  # noqa: E402  (class name, module, kwargs) -- shared with the test


def main():
    import sympy as sp

    S.import_hotpath()
    out = {}
    for case, (cls, module, kwargs) in CASES.items():
        mod = importlib.import_module(f"ppsci.equation.pde.{module}")
        eq = getattr(mod, cls)(**kwargs)
        out[case] = {name: sp.srepr(expr) for name, expr in eq.equations.items() if isinstance(expr, sp.Basic)}
        assert out[case], case
    path = os.path.join(HERE, "equations.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(path, {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
