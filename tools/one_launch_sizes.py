"""Where the one-launch step stops paying off with the batch size (narrow net, one constraint):  python tools/one_launch_sizes.py"""
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
    solver.engine.one_launch_max_points = 1 << 30
    return round(bench.pinn_entry("", solver, opt, cc, n, 1, 5, 100, 10, "<>")["ms_per_step"] * 1e3, 1)


if __name__ == "__main__":
    with tempfile.TemporaryDirectory() as tmp:
        for hidden in ([20] * 3, [32] * 4):
            for n in (16384, 32768, 65536, 131072, 262144):
                print(json.dumps({"net": f"{len(hidden)}x{hidden[0]}", "points": n, "separate_us": laplace(tmp, hidden, n, False),
                                  "one_launch_us": laplace(tmp, hidden, n, True)}), flush=True)
