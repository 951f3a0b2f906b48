"""bench.py's launch contract on the CPU SIMT emulator (gloo, tiny sizes; PPSCI_BENCH_EMU=1 is a test-only mode):
`python bench.py --gpus 2` WITHOUT a launcher must start two ranks itself and print one JSON line that says so."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, args):
    env = dict(os.environ, PPSCI_BENCH_EMU="1", OMP_NUM_THREADS="1", **extra_env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_gpus2_self_launches_two_ranks():
    from tests.emu import build_emu

    build_emu.build()  # once, before two ranks race for the build directory
    r = _run({}, ["--gpus", "2", "--steps", "1", "--warmup", "1"])
    assert r["n_gpus"] == 2 and r["config"]["parallelism"] == "dp2" and r["steps"] == 1
    assert "EMULATOR TEST MODE" in r["data"]
    s = r["strong_scaling"]
    assert s["n_gpus"] == 2 and s["comm_world_size"] == 2 and s["comm_backend"] == "gloo"
    assert s["points_per_rank"] * 2 == s["points_total"] and s["scaling"] == "strong"
    assert r["value"] > 0 and s["value"] > 0


def test_bench_gpus8_as_the_driver_launches_it():
    """The driver's 8-GPU run, first time right: eight ranks (here: emulator + gloo, tiny sizes), every rank a comm rank,
    the 1 M-point strong-scaling entry sharded 8 ways with its all-reduce timed on its own."""
    from tests.emu import build_emu

    build_emu.build()
    r = _run({}, ["--gpus", "8", "--steps", "1", "--warmup", "0"])
    assert r["n_gpus"] == 8 and r["config"]["parallelism"] == "dp8" and r["scaling"] == "weak"
    s = r["strong_scaling"]
    assert s["n_gpus"] == 8 and s["comm_world_size"] == 8 and s["comm_backend"] == "gloo" and s["scaling"] == "strong"
    assert s["points_per_rank"] * 8 == s["points_total"]
    assert s["allreduce_ms"] is not None and s["allreduce_ms"] > 0 and s["allreduce_bytes"] == 4 * 66819
    assert r["value"] > 0 and s["value"] > 0
    # the strong-scaling numbers also sit in `config` (the part of the line every consumer keeps), next to the communicator
    c = r["config"]
    assert c["strong_points_per_s"] == s["value"] and c["strong_ms_per_step"] == s["ms_per_step"]
    assert c["strong_points_per_rank"] * 8 == 1_000_000 or "EMULATOR" in r["data"]
    assert c["strong_allreduce_ms"] == s["allreduce_ms"] and c["strong_allreduce_bytes"] == 4 * 66819
    assert c["comm_world_size"] == 8 and c["comm_backend"] == "gloo"
    # the primary step under data parallelism: the fused tile kernel stays, Adam + fragments in one launch behind the all-reduce
    assert "all-reduce" in c["dp_step"] and "Adam + fragments" in c["dp_step"]


def test_bench_refuses_a_launcher_world_that_differs_from_gpus():
    env = dict(os.environ, PPSCI_BENCH_EMU="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr


def test_bench_secondary_pinn_entry_runs_the_solver_step(tmp_path, monkeypatch):
    """The secondary entries (cfg 1 / cfg 3) time `Solver`'s own iteration body through `bench.pinn_entry`; they only run
    on a GPU, so a signature drift between bench.py and the Solver is caught here on the emulator at a tiny size."""
    monkeypatch.setenv("PPSCI_BENCH_EMU", "1")
    monkeypatch.setenv("PPSCI_BENCH_PURE_STEPS", "1")
    import importlib

    import numpy as np

    from tests.emu import build_emu

    build_emu.inject()
    sys.path.insert(0, ROOT)
    bench = importlib.reload(importlib.import_module("bench"))
    import ppsci

    X = np.random.default_rng(0).uniform(0, 1, (48, 2)).astype(np.float32)
    solver, opt, cc, _ = bench.api_pinn("t", ("x", "y"), ("u",), [20, 20, 20], ppsci.equation.Laplace(2), X, "mean", None,
                                        str(tmp_path))
    e = bench.pinn_entry("tiny", solver, opt, cc, 48, 1.0, 5, 1, 0, "k")
    assert e["value"] > 0 and e["steps"] == 1

