"""Step time of small-batch configs with one launch per constraint (csrc/taylor_step.inc) against the separate
launches replayed as a HIP graph:  python tools/one_launch_bench.py"""
import json
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PPSCI_BENCH_PURE_STEPS"] = "1"
import bench  # noqa: E402
import numpy as np  # noqa: E402


def laplace(tmp, hidden, n, one):
    import ppsci

    X = np.random.default_rng(42).random((n, 2), dtype=np.float32)
    solver, opt, cc, _ = bench.api_pinn(f"lap{len(hidden)}x{hidden[0]}_{n}_{one}", ("x", "y"), ("u",), hidden,
                                        ppsci.equation.Laplace(2), X, "sum", None, tmp)
    solver.engine.one_launch = one
    e = bench.pinn_entry("", solver, opt, cc, n, 1, 5, 200, 20, "<>")

    def step():
        if not solver._step_in_one_launch([cc.fused], 1.0):
            solver.engine.forward_backward([cc.fused])
            opt.step(solver.engine.grad)

    return round(e["ms_per_step"] * 1e3, 2), round(bench.time_events(step, 50) * 1e6, 2)


if __name__ == "__main__":
    with tempfile.TemporaryDirectory() as tmp:
        for hidden, n in (([20] * 3, 10_000), ([20] * 5, 10_201), ([32] * 4, 4096), ([20] * 3, 1024), ([20] * 3, 16384),
                          ([32] * 4, 16384)):
            r = {"net": f"{len(hidden)}x{hidden[0]}", "points": n, "separate_us (wall, hip events)": laplace(tmp, hidden, n, False),
                 "one_launch_us (wall, hip events)": laplace(tmp, hidden, n, True)}
            print(json.dumps(r), flush=True)


def two_constraints(one, serial=False):
    """Laplace2D example shape: 10 201 interior points (S = 5 streams) + 400 boundary points (S = 1), engine level."""
    import torch
    from paddlescience_amd import hotpath as hp
    from paddlescience_amd.engine import Engine
    from tests.test_one_launch import _constraint, _weights

    lay = hp.NetLayout(2, 5, 20, 1, "tanh")
    eng = Engine(lay, torch.tensor(_weights(lay, 1), device="cuda"))
    eng.one_launch = one
    eng.one_launch_max_constraints = 4 if serial else 1  # 1 (default): the constraints' launches as parallel graph branches
    csts = [_constraint("cuda", "laplace", lay, 10_201, 1), _constraint("cuda", "value", lay, 400, 2)]
    return round(bench.time_wall(lambda: eng.train_step(csts, 1e-3), 300, 30) * 1e6, 2)


if __name__ == "__main__":
    print(json.dumps({"laplace2d example shape (5x20, 10201 + 400 points)": {"separate_us": two_constraints(False),
                                                                            "one_launch_parallel_branches_us": two_constraints(True),
                                                                            "one_launch_serial_with_adam_us": two_constraints(True, True)}}), flush=True)
