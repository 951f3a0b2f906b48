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
        if not solver._step_in_one_launch([cc.fused], [], 1.0):
            solver.engine.forward_backward([cc.fused])
            opt.step(solver.engine.grad)

    return round(e["ms_per_step"] * 1e3, 2), round(bench.time_events(step, 50) * 1e3, 2)


if __name__ == "__main__":
    with tempfile.TemporaryDirectory() as tmp:
        for hidden, n in (([20] * 3, 10_000), ([20] * 5, 10_201), ([32] * 4, 4096), ([20] * 3, 1024), ([20] * 3, 16384),
                          ([32] * 4, 16384)):
            r = {"net": f"{len(hidden)}x{hidden[0]}", "points": n, "separate_us (wall, hip events)": laplace(tmp, hidden, n, False),
                 "one_launch_us (wall, hip events)": laplace(tmp, hidden, n, True)}
            print(json.dumps(r), flush=True)
