"""bench.py -- collocation-points/sec of the PINN hot path (PDE residual + gradient + Adam) on MI355X.

PRIMARY LINE (`metric`, `value`, `roofline`, `cpu_baseline`): BASELINE.json configs[1] -- Allen-Cahn 1D+t, MLP
2->64x4->1 tanh, 100 000 collocation points per GPU, residual u_t - eps^2 u_xx + 5u^3 - 5u (eps = 0.01), MSE-mean,
Adam.  Synthetic points `default_rng(42+rank).uniform([0,-1],[1,1])`, Xavier-uniform weights `default_rng(1234)`
(SURVEY.md 8d).  A "step" = taylor_fwd + epilogue + taylor_bwd + gradient reduce (+ RCCL all-reduce of the flat
gradient when N > 1) + fused Adam, inputs resident in HBM.  One process per GPU
(`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`); per-GPU work is fixed => weak scaling.

The same JSON line also carries
  * `parity`     -- the TIMED net (initial weights) on the first 2 048 points of the timed batch against
                    tests/golden/bench_nets.npz, i.e. values produced by running the reference's own Python in fp64
                    (tests/golden/make_bench_nets_golden.py): residual / gradient rel-L2, loss rel;
  * `secondary`  -- (N = 1 only) the other BASELINE configs, each with points/s, its dominant kernel's roofline
                    fraction and its own parity: cfg 1 Laplace2D 3x20 10 k points (+ its cpu_baseline), cfg 3 shard
                    NavierStokes 5x128 125 k points, cfg 4 TFNO-2D 64x64 batch 16, cfg 5 SPINN Helmholtz3D 128^3;
  * `strong_scaling` -- BASELINE configs[2] as the north star states it: 1 000 000 NavierStokes points sharded
                    rank-strided over the N ranks (N = 1: all of them on one GPU), one SUM all-reduce of the flat
                    gradient (66 819 floats = 267 KB) per step; `value` = 1e6 / step time, "scaling": "strong".
Rank 0 prints ONE JSON line.  The oracle (oracle/) is used by the `cpu_baseline` legs and, as the CHECKER of the
cfg 4 / cfg 5 parity entries, never inside a timed region.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HIDDEN, WIDTH, N_PER_GPU = 4, 64, 100_000
# TEST MODE (tests/test_bench_launch.py only): the launch / rank plumbing of this file on the CPU SIMT emulator with the
# gloo backend and tiny sizes; the JSON line says so in `data`.  Never set on a GPU box.
EMU = os.environ.get("PPSCI_BENCH_EMU") == "1"
# profiling runs (tools/*_step.py under rocprofv3): launch nothing but the training steps, so that launches / steps in the
# counter files are launches per step
PURE = os.environ.get("PPSCI_BENCH_PURE_STEPS") == "1"
if EMU:
    N_PER_GPU = 128
PEAK_FP32_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 MFMA (f32 in) peak == fp32 vector peak
PEAK_HBM_TBPS = 8.0       # MI355X_MICROARCH.md: HBM3E
PEAK_BF16_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA
# The GEMMs run on the bf16 (XDL) pipe as six bf16 products per fp32 product: the pipe-true ceiling for ALGORITHMIC fp32 flops
PEAK_PIPE_TFLOPS = PEAK_BF16_TFLOPS / 6.0
CLOCK_GHZ, N_SIMD = 2.4, 1024
EPS = 0.01
N_FIX = 2048              # points of the reference-run fixtures (tests/golden/bench_nets.npz)
NS_TOTAL = 64 if EMU else 1_000_000  # BASELINE configs[2]
CPU_THREADS = 8  # measured on the GPU box (tools/cpu_threads.py): 8 threads is the fastest setting for this
                 # graph of small ops; 32+ threads are slower, 256 threads 100x slower


def bench_weights(d_in, hidden, d_out, seed=1234):
    """W ~ U(+-sqrt(6/(in+out))), b = 0, default_rng(seed), layer by layer, W then b; flat fp32
    vector in `parameters()` order.  (Same draw as oracle.taylor_np.make_net / tests/golden/make_bench_nets_golden.)"""
    rng = np.random.default_rng(seed)
    sizes = [d_in] + list(hidden) + [d_out]
    out = []
    for fi, fo in zip(sizes[:-1], sizes[1:]):
        lim = np.sqrt(6.0 / (fi + fo))
        out.append(rng.uniform(-lim, lim, size=(fi, fo)).astype(np.float32).ravel())
        out.append(np.zeros(fo, np.float32))
    return np.concatenate(out)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "bench_nets.npz"))


def _sync():
    if not EMU:
        torch.cuda.synchronize()


class quiet_host:
    """The timed regions run with the cyclic garbage collector paused (as `timeit` does) after a full collection: a gen-2
    collection of an earlier config's solver / graph objects landing inside a 25-step window of a ~1 ms host-launched step was
    measured as a 60 ms hiccup (TFNO entry 0.98 -> 3.5 ms per step, identical device time)."""

    def __enter__(self):
        import gc

        gc.collect()
        self._was = gc.isenabled()
        gc.disable()

    def __exit__(self, *exc):
        import gc

        if self._was:
            gc.enable()


def time_wall(fn, steps, warmup, barrier=None):
    sync = barrier or _sync
    with quiet_host():  # (the collection first, the warm-up steps directly in front of the timed ones: no idle gap between them)
        for _ in range(warmup):
            fn()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        sync()
        dt = time.perf_counter() - t0
    return dt / steps


def profile_info(stem, kernel_substr=None, ms_per_step=None, once_per_step=False):
    """What the committed rocprofv3 summaries of config `stem` say (profiles/<round>_<stem>_{kernel_stats.csv,
    pmc_summary.json}, newest round; tools/profile_all.sh + tools/summarize_profile.py): the DOMINANT kernel of a step by
    total time, the HBM bytes of one step (sum over all kernels of (2 x FETCH_SIZE + WRITE_SIZE) per launch x launches
    per step; a step = one launch of the kernel that applies Adam) and, for `kernel_substr`, that kernel's own bytes per launch.  PMC
    counters can only be read by rocprofv3 around a process, so these figures are QUOTED from the named files, not
    measured by this run."""
    import csv
    import glob

    out = {}
    try:
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{stem}_pmc_summary.json")))
        if not files:
            return out
        pmc = json.load(open(files[-1]))
        out["source"] = os.path.relpath(files[-1], ROOT)
        stats = files[-1].replace("_pmc_summary.json", "_kernel_stats.csv")
        if os.path.exists(stats):
            rows = [r for r in csv.DictReader(open(stats)) if "at::native" not in r["Name"] and "rocclr" not in r["Name"]]
            tot = sum(float(r["TotalDurationNs"]) for r in rows) or 1.0
            top = max(rows, key=lambda r: float(r["TotalDurationNs"]))
            out["dominant_kernel"] = {"name": top["Name"].split("(")[0][-70:], "share_of_kernel_time": float(top["TotalDurationNs"]) / tot,
                                      "avg_us": float(top["AverageNs"]) / 1e3}
        # (a step = one launch of the kernel that applies the optimizer: adam_kernel, or the row reductions + Adam in one launch)
        nstep = next((v.get("launches_pmc_fetch") for k, v in pmc.items()
                      if k.startswith("adam_kernel") or k.startswith("reduce_rows_multi_adam_kernel")), None)
        if once_per_step:  # one-constraint PINN step: every kernel type runs once (forward, epilogue, reductions, reverse, Adam)
            nstep = 1
        if nstep:
            byts = sum(v.get("hbm_bytes_per_launch", 0.0) * (1 if once_per_step else v.get("launches_pmc_fetch", 0))
                       for v in pmc.values()) / nstep
            out["hbm_bytes_per_step"] = byts
            if ms_per_step:
                out["hbm_frac_of_8TBps_over_step"] = byts / (ms_per_step * 1e-3) / (PEAK_HBM_TBPS * 1e12)
        if kernel_substr:
            for k, v in pmc.items():
                if kernel_substr in k:
                    out["kernel_hbm_bytes_per_launch"] = v.get("hbm_bytes_per_launch")
    except Exception as e:  # noqa: BLE001 -- quoted context must never cost the measured line
        out["error"] = f"{type(e).__name__}: {e}"[:200]
    return out


def pipe_roofline(ach_tflops, stem, kernel_substr):
    """What VERDICT r03 asked for next to `frac`: the fraction of the ceiling of the pipe the GEMMs actually run on
    (2 500 TF bf16 / 6 products), and -- QUOTED from the committed PMC summary of config `stem` -- how busy that pipe was:
    SQ_VALU_MFMA_BUSY_CYCLES / (kernel duration x 2.4 GHz x 1 024 SIMDs) for the kernel whose name contains `kernel_substr`."""
    import csv
    import glob

    out = {"pipe_peak": PEAK_PIPE_TFLOPS, "pipe_frac": ach_tflops / PEAK_PIPE_TFLOPS if ach_tflops else None,
           "pipe": "v_mfma_f32_16x16x32_bf16, six products per fp32 product: 2 500 / 6 TFLOP/s of algorithmic fp32 flops"}
    try:
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{stem}_pmc_summary.json")))
        if files:
            pmc = json.load(open(files[-1]))
            stats = files[-1].replace("_pmc_summary.json", "_kernel_stats.csv")
            busy = next((v.get("SQ_VALU_MFMA_BUSY_CYCLES") for k, v in pmc.items() if kernel_substr in k), None)
            ns = None
            if os.path.exists(stats):
                ns = next((float(r["AverageNs"]) for r in csv.DictReader(open(stats)) if kernel_substr in r["Name"]), None)
            if busy and ns:
                out["mfma_busy"] = busy / (ns * CLOCK_GHZ * N_SIMD)
                out["mfma_busy_source"] = os.path.relpath(files[-1], ROOT)
    except Exception as e:  # noqa: BLE001 -- quoted context must never cost the measured line
        out["error"] = f"{type(e).__name__}: {e}"[:200]
    return out


def step_hbm_roofline(stem, t_step):
    """Step-level HBM roofline of a config whose kernels are all HBM / latency-bound: achieved = (HBM bytes of one step,
    summed over all its kernels from the committed PMC summary) / (the step time measured by THIS run); the entry names the
    dominant kernel of the step from the committed rocprofv3 kernel statistics."""
    prof = profile_info(stem, None, t_step * 1e3)
    byts = prof.get("hbm_bytes_per_step")
    dom = prof.get("dominant_kernel", {})
    ach = byts / t_step / 1e12 if byts else None
    return {"bound": "hbm", "scope": "whole step", "kernel": dom.get("name"), "dominant_kernel_share": dom.get("share_of_kernel_time"),
            "dominant_kernel_avg_us": dom.get("avg_us"), "achieved": ach, "peak": PEAK_HBM_TBPS, "unit": "TB/s",
            "frac": ach / PEAK_HBM_TBPS if ach else None, "traffic": byts, "traffic_source": prof.get("source")}


def time_events(fn, reps=20, median=False):
    """Average (or median) duration of `fn`'s launches with HIP events on the launch stream (torch's current stream)."""
    if EMU:
        return time_wall(fn, 1, 0)
    fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    dts = [ev[i].elapsed_time(ev[i + 1]) for i in range(reps)]
    return float(np.median(dts) if median else np.mean(dts)) * 1e-3


def time_events_list(fn, reps):
    """Durations (seconds) of `reps` back-to-back launches of `fn`, HIP events on the launch stream, no warm-up launch."""
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return [ev[i].elapsed_time(ev[i + 1]) * 1e-3 for i in range(reps)]


# ------------------------------------------------------------------------------------------ cfg 2 (primary)
def allen_cahn_program(n_scale):
    from paddlescience_amd import _lib as L
    from paddlescience_amd import hotpath as hp

    pr = hp.Program(4, 2)
    u, ut, uxx = pr.ld_u(0), pr.ld_u(2), pr.ld_u(3)
    five = pr.const(5.0)
    r = pr.op(L.OP_SUB,
              pr.op(L.OP_ADD, pr.op(L.OP_SUB, ut, pr.op(L.OP_MUL, pr.const(EPS**2), uxx)),
                    pr.op(L.OP_MUL, pr.op(L.OP_MUL, pr.op(L.OP_MUL, five, u), u), u)),
              pr.op(L.OP_MUL, five, u))
    pr.residual(r, scale=1.0 / n_scale)
    return pr.build()


def allen_cahn_constraint(dev, X, n_scale, want_residual=False):
    from paddlescience_amd import hotpath as hp
    from paddlescience_amd.engine import FusedConstraint

    lay = hp.NetLayout(2, HIDDEN, WIDTH, 1, "tanh")
    xs = [torch.tensor(X[:, j].copy(), device=dev) for j in range(2)]  # (t, x)
    streams = hp.StreamSpec([[0.0, 1.0], [1.0, 0.0]], 1)  # streams: u, u_x, u_t, u_xx
    return lay, FusedConstraint("EQ", lay, streams, allen_cahn_program(n_scale), xs, [], ["allen_cahn"],
                                want_residual=want_residual)


def parity_allen_cahn(dev, flat, X):
    """The timed net at its initial weights on the first N_FIX points of the timed batch vs the reference-run values."""
    from paddlescience_amd.engine import Engine

    G = gold()
    assert np.array_equal(X[:N_FIX], G["allen_cahn_4x64/X"]), "bench batch no longer starts with the fixture's points"
    lay, cst = allen_cahn_constraint(dev, X[:N_FIX], N_FIX, want_residual=True)
    eng = Engine(lay, torch.tensor(flat, device=dev))
    eng.forward_backward([cst])
    torch.cuda.synchronize()
    return {"reference": "tests/golden/bench_nets.npz (reference's own Python, fp64, torch-backed paddle shim)",
            "points": N_FIX,
            "residual_rel_l2": rel(cst.resid[0].cpu().numpy(), G["allen_cahn_4x64/res/allen_cahn"]),
            "grad_rel_l2": rel(eng.grad.cpu().numpy(), G["allen_cahn_4x64/grad"]),
            "loss_rel": abs(cst.losses()["allen_cahn"] / float(G["allen_cahn_4x64/total"]) - 1.0)}


def cpu_steps(model, cst, n_params, steps, threads):
    """Median wall time of `steps` full training steps (residual + MSE + backward + Adam) of the oracle's
    restatement of the reference algorithm (reverse-over-reverse autodiff, fp32, torch-CPU)."""
    from oracle import ref_torch as R

    torch.set_num_threads(threads)
    opt = R.Adam(n_params, 1e-3, dtype=np.float32)
    flat = np.concatenate([p.detach().numpy().ravel() for p in model.parameters()])
    times = []
    first = None
    for i in range(steps + 1):
        t0 = time.perf_counter()
        total, _, g, _ = R.loss_and_grads(model, [cst])
        if first is None:
            first = (total, g.copy())  # at the initial weights: what `parity.full_batch` compares the GPU step with
        flat = opt.step(flat, g).astype(np.float32)
        off = 0
        with torch.no_grad():
            for p in model.parameters():
                k = p.numel()
                p.copy_(torch.from_numpy(flat[off:off + k].reshape(p.shape)))
                off += k
        if i > 0:
            times.append(time.perf_counter() - t0)
    cpu_steps.first = first
    return float(np.median(times))


def oracle_net(d_in, hidden, d_out, flat):
    from oracle import taylor_np as T

    net = T.make_net(d_in, hidden, d_out)
    off = 0
    for i in range(len(net.weights)):
        n = net.weights[i].size
        net.weights[i] = flat[off:off + n].reshape(net.weights[i].shape).astype(np.float64)
        off += n
        n = net.biases[i].size
        net.biases[i] = flat[off:off + n].astype(np.float64)
        off += n
    return net


def cpu_baseline(kind, flat, X, steps=20):
    """`kind` in {"allen_cahn", "laplace"}: the same batch the GPU was timed on; best-thread and 1-thread figures."""
    from oracle import ref_torch as R

    n = X.shape[0]
    if kind == "allen_cahn":
        net = oracle_net(2, [WIDTH] * HIDDEN, 1, flat)
        model = R.MLP(("t", "x"), ("u",), net, dtype=torch.float32)
        cst = dict(name="EQ", input={"t": X[:, :1], "x": X[:, 1:]}, exprs={"allen_cahn": R.allen_cahn_fn(EPS)},
                   label={"allen_cahn": np.zeros((n, 1), np.float32)}, reduction="mean")
    else:
        net = oracle_net(2, [20] * 3, 1, flat)
        model = R.MLP(("x", "y"), ("u",), net, dtype=torch.float32)
        cst = dict(name="EQ", input={"x": X[:, :1], "y": X[:, 1:]},
                   exprs={k: R.lambdify(e, model, dtype=torch.float32) for k, e in R.laplace_exprs(2).items()},
                   label={"laplace": np.zeros((n, 1), np.float32)}, reduction="sum")
    thr = min(CPU_THREADS, os.cpu_count())
    t_best = cpu_steps(model, cst, flat.size, steps, thr)
    first = cpu_steps.first
    t_one = cpu_steps(model, cst, flat.size, steps, 1)
    # `cores` (the contract's name) = the threads actually used, as `threads` says; the machine's size is `host_cores`
    return {"value": n / t_best, "unit": "points/s", "cores": thr, "threads": thr, "host_cores": os.cpu_count(), "kind": "port",
            "value_1_thread": n / t_one, "_first_step": first,
            "sample": f"the same {n}-point batch, median of {steps} full training steps (residual + MSE + backward + "
                      "Adam) of the torch-CPU fp32 reverse-over-reverse restatement of the reference algorithm "
                      f"(oracle/ref_torch.py); {thr} threads (the fastest setting) and 1 thread of {os.cpu_count()} "
                      f"host cores; torch {torch.__version__}"}


# ------------------------------------------------------------------------------------------ API-level configs
def api_pinn(tag, inputs, outputs, hidden, eq, X, reduction, weight, tmp, batch=None, periods=None):
    import ppsci

    n = X.shape[0]
    model = ppsci.arch.MLP(inputs, outputs, len(hidden), hidden[0], "tanh", periods=periods)
    flat = bench_weights(len(inputs) + len(periods or {}), hidden, len(outputs))
    model.flat_params.copy_(torch.tensor(flat).to(model.flat_params.device))
    keys = list(eq.equations.keys())
    cfg = {"dataset": {"name": "NamedArrayDataset", "input": {k: X[:, j:j + 1] for j, k in enumerate(inputs)},
                       "label": {k: np.zeros((n, 1), np.float32) for k in keys},
                       "weight": None if weight is None else {k: np.full((n, 1), weight, np.float32) for k in keys}},
           "batch_size": batch or n, "sampler": {"name": "BatchSampler", "shuffle": False, "drop_last": True}}
    pde = ppsci.constraint.SupervisedConstraint(cfg, ppsci.loss.MSELoss(reduction), eq.equations, name="EQ")
    opt = ppsci.optimizer.Adam(1e-3)(model)
    solver = ppsci.solver.Solver(model, {"EQ": pde}, os.path.join(tmp, tag), opt, epochs=1, iters_per_epoch=1)
    return solver, opt, solver._compiled["EQ"], flat


def api_parity(name, inputs, outputs, hidden, make_eq, reduction, weight, tmp, periods=None):
    """The config's net (same seeded weights) on the fixture's N_FIX points through the ppsci API vs reference-run values."""
    G = gold()
    X = G[f"{name}/X"]
    eq = make_eq()
    solver, _, cc, _ = api_pinn("par_" + name, inputs, outputs, hidden, eq, X, reduction, weight, tmp, periods=periods)
    solver.engine.forward_backward([cc.fused])
    losses = cc.fused.losses()
    res = solver.predict({k: X[:, j:j + 1] for j, k in enumerate(inputs)}, eq.equations, batch_size=None, return_numpy=True)
    keys = list(eq.equations.keys())
    return {"points": N_FIX, "reference": "tests/golden/bench_nets.npz",
            "residual_rel_l2": max(rel(res[k][:, 0], G[f"{name}/res/{k}"]) for k in keys),
            "grad_rel_l2": rel(solver.engine.grad.cpu().numpy(), G[f"{name}/grad"]),
            "loss_rel": max(abs(losses[k] / float(G[f"{name}/loss/{k}"]) - 1.0) for k in keys)}


def pinn_entry(label, solver, opt, cc, n, p_mat, S, steps, warmup, kernel_name, stem=None):
    def step():  # == the body of Solver.train()'s iteration for one constraint
        if not solver._step_in_one_launch([cc.fused], 1.0):
            solver.engine.forward_backward([cc.fused])
            opt.step(solver.engine.grad)

    t = time_wall(step, steps, warmup)
    if PURE:
        return {"config": label, "value": n / t, "unit": "points/s", "ms_per_step": t * 1e3, "steps": steps}
    if solver.engine.one_launch_ready([cc.fused]):
        # the whole step is ONE kernel (csrc/taylor_step.inc): forward (2 P S flops per point) + reverse (4 P S)
        t_k = time_events(step)
        ach = 6.0 * p_mat * S * n / t_k / 1e12
        name = "taylor_step_kernel" + kernel_name[kernel_name.index("<"):]
        return {"config": label, "value": n / t, "unit": "points/s", "ms_per_step": t * 1e3, "steps": steps,
                "launches_per_step": 1, "matrix_tflops_step": 6.0 * p_mat * S * n / t / 1e12,
                "roofline": {"bound": "mfma", "kernel": name, "achieved": ach, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                             "frac": ach / PEAK_FP32_TFLOPS, **pipe_roofline(ach, stem, name.split("<")[0]), "kernel_ms": t_k * 1e3,
                             "note": "forward + epilogue + reverse + reduction tree + Adam in one launch; latency-bound at "
                                     "this size (one 16-point tile per wave, one round)"}}
    t_bwd = time_events(lambda: cc.fused.backward(solver.engine.params))  # incl. the two small reduction kernels
    t_fwd = time_events(lambda: cc.fused.forward(solver.engine.params, True))
    ach = 4.0 * p_mat * S * n / t_bwd / 1e12
    return {"config": label, "value": n / t, "unit": "points/s", "ms_per_step": t * 1e3, "steps": steps,
            "matrix_tflops_step": 6.0 * p_mat * S * n / t / 1e12,
            "roofline": {"bound": "mfma", "kernel": kernel_name, "achieved": ach, "peak": PEAK_FP32_TFLOPS,
                         "unit": "TFLOP/s", "frac": ach / PEAK_FP32_TFLOPS,
                         **pipe_roofline(ach, stem, kernel_name.split(",")[0]), "kernel_ms": t_bwd * 1e3,
                         "fwd_plus_epilogue_ms": t_fwd * 1e3}}


def secondary_laplace(tmp, steps, warmup, with_cpu):
    import ppsci

    X = np.random.default_rng(42).random((10_000, 2), dtype=np.float32)
    solver, opt, cc, flat = api_pinn("lap", ("x", "y"), ("u",), [20] * 3, ppsci.equation.Laplace(2), X, "sum", None, tmp)
    e = pinn_entry("cfg1 Laplace2D, MLP 2->20x3->1 tanh, 10 000 interior points, u_xx+u_yy, MSE-sum, Adam "
                   "(BASELINE.json configs[0]); launch/latency-bound: 625 tiles on 1 024 wave slots",
                   solver, opt, cc, 10_000, 2 * 20 + 2 * 400 + 20, 5, steps, warmup, "taylor_bwd_kernel<2, 2, 2, 0>", "laplace")
    e["step_profile"] = profile_info("laplace", None, e["ms_per_step"], once_per_step=True)
    if PURE:
        return e
    e["parity"] = api_parity("laplace2d_3x20", ("x", "y"), ("u",), [20] * 3, lambda: ppsci.equation.Laplace(2), "sum",
                             None, tmp)
    if with_cpu:
        e["cpu_baseline"] = cpu_baseline("laplace", flat, X)
        e["cpu_baseline"].pop("_first_step", None)
        e["speedup_vs_cpu_best_thread"] = e["value"] / e["cpu_baseline"]["value"]
    return e


def ns_setup(tmp, X, tag, batch=None):
    import ppsci

    eq = ppsci.equation.NavierStokes(0.01, 1.0, 2, False)
    return api_pinn(tag, ("x", "y"), ("u", "v", "p"), [128] * 5, eq, X, "sum", 1e-4, tmp, batch)


NS_PMAT = 2 * 128 + 4 * 128 * 128 + 128 * 3


def secondary_ns(tmp, steps, warmup):
    import ppsci

    X = np.random.default_rng(42).uniform(-0.05, 0.05, (NS_TOTAL, 2)).astype(np.float32)[0::8]  # rank 0 of 8, strided
    solver, opt, cc, _ = ns_setup(tmp, X, "ns")
    e = pinn_entry("cfg3 shard: LDC NavierStokes 2-D steady, MLP 2->128x5->3 tanh, 125 000 points (rank 0 of 8 of the "
                   "1 M-point cloud), continuity + momentum_x + momentum_y, weights 1e-4, MSE-sum, Adam",
                   solver, opt, cc, X.shape[0], NS_PMAT, 5, steps, warmup, "taylor_bwd_wx_kernel<8, 1, 5, 2, 2, 0>", "ns")
    e["step_profile"] = profile_info("ns", None, e["ms_per_step"], once_per_step=True)
    if PURE:
        return e
    e["parity"] = api_parity("ns2d_5x128", ("x", "y"), ("u", "v", "p"), [128] * 5,
                             lambda: ppsci.equation.NavierStokes(0.01, 1.0, 2, False), "sum", 1e-4, tmp)
    return e


def secondary_ac256(tmp, steps, warmup):
    """BASELINE configs[1] at the shape of the reference's OWN yaml (examples/allen_cahn/conf/allen_cahn.yaml:38-42): MLP (t, x) -> u,
    4 x 256 tanh, period embedding of x (d0 = 3), 100 000 collocation points, MSE-mean, Adam.  Padded width 256: forward on the
    feature-split XDL kernel, reverse on the layer-by-layer XDL kernel (csrc/taylor_bwd_lw.inc, round 5: one launch per hidden
    matrix, its gradient in registers, the adjoint handed on through the workspace)."""
    import ppsci
    from paddlescience_amd import _lib as L

    periods = {"x": [2.0, False]}
    X = np.random.default_rng(42).uniform([0, -1], [1, 1], (N_PER_GPU, 2)).astype(np.float32)
    solver, opt, cc, _ = api_pinn("ac256", ("t", "x"), ("u",), [256] * 4, ppsci.equation.AllenCahn(0.01), X, "mean", None, tmp,
                                  periods=periods)
    p_mat = 3 * 256 + 3 * 256 * 256 + 256
    e = pinn_entry("cfg2 at the reference yaml's shape: Allen-Cahn 1D+t, MLP 2->256x4->1 tanh with periods {x: 2.0}, 100 000 points, "
                   "MSE-mean, Adam (examples/allen_cahn/conf/allen_cahn.yaml)", solver, opt, cc, N_PER_GPU, p_mat, 4, steps, warmup,
                   "taylor_bwd_lw_kernel<16, 2, 2, 1, 0>", None)
    if PURE:
        return e
    # the round-2 fp32-MFMA reverse kernel this replaces (per-tile gradient blocks through HBM), same buffers
    L.lib().ppsci_set_bwd_layerwise(0)
    try:
        s2, o2, c2, _ = api_pinn("ac256_old", ("t", "x"), ("u",), [256] * 4, ppsci.equation.AllenCahn(0.01), X, "mean", None, tmp,
                                 periods=periods)
        s2.engine.forward_backward([c2.fused])
        e["reverse_ms_round2_kernel"] = time_events(lambda: c2.fused.backward(s2.engine.params)) * 1e3
        del s2, o2, c2
        torch.cuda.empty_cache()
    finally:
        L.lib().ppsci_set_bwd_layerwise(1)
    e["parity"] = api_parity("allen_cahn_4x256_period", ("t", "x"), ("u",), [256] * 4, lambda: ppsci.equation.AllenCahn(0.01), "mean",
                             None, tmp, periods=periods)
    return e


def strong_ns(tmp, world, rank, steps, warmup, barrier):
    """BASELINE configs[2]: 1 M NavierStokes points, rank-strided shards, one SUM all-reduce of the flat gradient."""
    # the whole cloud goes into the dataset; the (Distributed)BatchSampler hands this rank its strided shard
    # X[rank::world] (/root/reference/ppsci/data/__init__.py:76-99), bound once and resident in HBM
    X = np.random.default_rng(42).uniform(-0.05, 0.05, (NS_TOTAL, 2)).astype(np.float32)
    n_local = NS_TOTAL // world
    solver, opt, cc, _ = ns_setup(tmp, X, f"ns_strong_r{rank}", n_local)
    eng = solver.engine
    assert eng.world == world and cc.fused.n == n_local, (eng.world, world, cc.fused.n)

    def step():
        eng.forward_backward([cc.fused])
        eng.allreduce()
        opt.step(eng.grad)

    t = time_wall(step, steps, warmup, barrier)
    # the collective alone (HIP events on the launch stream around eng.allreduce(); wall clock in the emulator test mode),
    # after a barrier so that it measures the all-reduce, not the wait for the slowest rank's kernels
    t_ar = None
    if world > 1:
        ts = []
        for _ in range(5):
            barrier()
            ts.append(time_events(eng.allreduce, 1) if not EMU else time_wall(eng.allreduce, 1, 0))
        t_ar = float(np.median(ts))
    tt = torch.tensor([t, t_ar or 0.0], device="cpu" if EMU else "cuda", dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
    t = float(tt[0])
    t_ar = float(tt[1]) if world > 1 else None
    return {"workload": "LDC NavierStokes 2-D steady, MLP 2->128x5->3 tanh, 1 000 000 collocation points sharded "
                        "rank-strided over the ranks, SUM all-reduce of the flat gradient, Adam (BASELINE.json configs[2])",
            "value": NS_TOTAL / t, "unit": "points/s", "ms_per_step": t * 1e3, "steps": steps, "n_gpus": world,
            "scaling": "strong", "points_total": NS_TOTAL, "points_per_rank": n_local,
            "allreduce_bytes": int(eng.grad.numel()) * 4, "allreduce_ms": None if t_ar is None else t_ar * 1e3,
            "comm_world_size": eng.world,
            "comm_backend": torch.distributed.get_backend() if world > 1 else None,
            "matrix_tflops_per_gpu": 6.0 * NS_PMAT * 5 * n_local / t / 1e12}


def extra_uno(steps, warmup, B=16, R=16):
    """The reference's UNO configuration (examples/neuraloperator/conf/uno_darcyflow_pretrain.yaml: in 3, hidden 64, lifting 256,
    projection 64, five Fourier layers 32-64-64-64-32 with modes 16-8-8-8-16 and scalings 1, 0.5, 1, 2, 1, group_norm, domain padding
    0.2, batch 16 at 16 x 16); one training step = forward + MSE + backward + fused Adam (uno_engine.UnoNative)."""
    import ppsci

    torch.manual_seed(0)
    outs, modes = [32, 64, 64, 64, 32], [[16, 16], [8, 8], [8, 8], [8, 8], [16, 16]]
    scal = [[1.0, 1.0], [0.5, 0.5], [1, 1], [2, 2], [1, 1]]
    model = ppsci.arch.UNONet(("x",), ("y",), 3, 1, 64, 256, 64, n_layers=5, uno_out_channels=outs, uno_n_modes=modes,
                              uno_scalings=scal, norm="group_norm", domain_padding=0.2, domain_padding_mode="one-sided")
    x = torch.as_tensor(np.random.default_rng(42).standard_normal((B, 3, R, R)).astype(np.float32)).cuda()
    y = torch.as_tensor(np.random.default_rng(43).standard_normal((B, 1, R, R)).astype(np.float32)).cuda()
    opt = ppsci.optimizer.Adam(1e-3)(model)
    # parity (checker: the oracle's fp64 restatement of unonet.py, pinned by the reference-run tests/golden/uno.npz)
    from oracle import ref_torch as Rf

    P = {n: p.detach().cpu().double().requires_grad_(True) for n, p in torch.nn.Module.named_parameters(model)}
    yo = Rf.uno_forward(x.cpu().double(), P, outs, modes, scal, None, "group_norm", domain_padding=0.2)
    lo = ((yo - y.cpu().double()) ** 2).mean()
    names = sorted(P)
    go = dict(zip(names, torch.autograd.grad(lo, [P[n] for n in names])))
    nat = model.native()
    mse_loss = ppsci.loss.MSELoss("mean")
    yh = nat.forward(x.contiguous())
    lh, gy = mse_loss.value_and_grad(yh, y, "y")
    yh = yh.clone()
    model.flat_grad.fill_(float("nan"))
    nat.backward(gy)
    gh = {n: p.grad.detach().cpu().numpy().copy() for n, p in torch.nn.Module.named_parameters(model)}
    model.flat_grad.zero_()
    parity = {"checker": "oracle/ref_torch.uno_forward fp64 (pinned by reference-run tests/golden/uno.npz), the whole timed batch, "
                         "the timed model's weights",
              "output_rel_l2": rel(yh.detach().cpu().numpy(), yo.detach().numpy()),
              "grad_rel_l2": max(rel(gh[n], go[n].numpy()) for n in names),
              "loss_rel": abs(float(lh["y"].detach()) / float(lo.detach()) - 1.0)}
    from paddlescience_amd.engine import step_with_adam
    from paddlescience_amd.operator_engine import OperatorConstraint, OperatorEngine

    cst = OperatorConstraint("Sup", model, {"y": lambda d: d["y"]}, mse_loss, x.device, ["y"], B)
    cst.bind({"x": x}, {"y": y})
    eng = OperatorEngine(model)

    def step():
        step_with_adam(eng, [cst], opt, model.flat_params)

    t = time_wall(step, steps, warmup)
    t_ev = time_events(step, reps=steps)
    return {"config": "extra: UNO Darcy (reference config): 16x16 grid padded to 19x19, batch 16, in 3, hidden 64, lifting 256, "
                      "projection 64, layers 32-64-64-64-32 on grids 19/10/10/20/19, group_norm; forward + MSE + backward + Adam",
            "value": B * R * R / t, "unit": "grid-points/s", "samples_per_s": B / t, "ms_per_step": t * 1e3, "steps": steps,
            "ms_per_step_hip_events": t_ev * 1e3, "params": int(model.flat_params.numel()), "roofline": step_hbm_roofline("uno", t),
            "native_forward_backward": type(eng.native).__name__, "parity": parity}


def extra_sfno(steps, warmup, B=4, H=32, W=64):
    """The reference's SFNO configuration (examples/neuraloperator/conf/sfno_swe_pretrain.yaml: in 3, out 3, hidden 32, lifting 256,
    projection 64, 4 layers, n_modes (32, 32) = 32 degrees x 16 orders, group_norm, batch 4 on the 32 x 64 grid); one training step =
    forward + MSE + backward + fused Adam (fno_engine.FnoNative with the transform pair of csrc/sht.hip)."""
    import ppsci

    torch.manual_seed(0)
    model = ppsci.arch.SFNONet(("x",), ("y",), (32, 32), 32, in_channels=3, out_channels=3, lifting_channels=256,
                               projection_channels=64, n_layers=4, norm="group_norm")
    x = torch.as_tensor(np.random.default_rng(42).standard_normal((B, 3, H, W)).astype(np.float32)).cuda()
    y = torch.as_tensor(np.random.default_rng(43).standard_normal((B, 3, H, W)).astype(np.float32)).cuda()
    opt = ppsci.optimizer.Adam(1e-3)(model)
    # parity (checker: the oracle's fp64 restatement of sfnonet.py / paddle_harmonics, pinned by the reference-run tests/golden/sfno.npz)
    from oracle import ref_torch as Rf

    P = {n: p.detach().cpu().double().requires_grad_(True) for n, p in torch.nn.Module.named_parameters(model)}
    yo = Rf.sfno_forward(x.cpu().double(), P, 4, (32, 32), "group_norm")
    lo = ((yo - y.cpu().double()) ** 2).mean()
    names = sorted(P)
    go = dict(zip(names, torch.autograd.grad(lo, [P[n] for n in names])))
    nat = model.native()
    mse_loss = ppsci.loss.MSELoss("mean")
    yh = nat.forward(x.contiguous())
    lh, gy = mse_loss.value_and_grad(yh, y, "y")
    yh = yh.clone()
    model.flat_grad.fill_(float("nan"))
    nat.backward(gy)
    gh = {n: p.grad.detach().cpu().numpy().copy() for n, p in torch.nn.Module.named_parameters(model)}
    model.flat_grad.zero_()
    parity = {"checker": "oracle/ref_torch.sfno_forward fp64 (pinned by reference-run tests/golden/sfno.npz), the whole timed batch, "
                         "the timed model's weights",
              "output_rel_l2": rel(yh.detach().cpu().numpy(), yo.detach().numpy()),
              "grad_rel_l2": max(rel(gh[n], go[n].numpy()) for n in names),
              "loss_rel": abs(float(lh["y"].detach()) / float(lo.detach()) - 1.0)}
    from paddlescience_amd.engine import step_with_adam
    from paddlescience_amd.operator_engine import OperatorConstraint, OperatorEngine

    cst = OperatorConstraint("Sup", model, {"y": lambda d: d["y"]}, mse_loss, x.device, ["y"], B)
    cst.bind({"x": x}, {"y": y})
    eng = OperatorEngine(model)

    def step():
        step_with_adam(eng, [cst], opt, model.flat_params)

    t = time_wall(step, steps, warmup)
    t_ev = time_events(step, reps=steps)
    return {"config": "extra: SFNO shallow-water shape (reference config): 32x64 lat-lon grid, batch 4, in 3, out 3, hidden 32, lifting "
                      "256, projection 64, 4 layers, 32 degrees x 16 orders, group_norm; forward + MSE + backward + Adam",
            "value": B * H * W / t, "unit": "grid-points/s", "samples_per_s": B / t, "ms_per_step": t * 1e3, "steps": steps,
            "ms_per_step_hip_events": t_ev * 1e3, "params": int(model.flat_params.numel()), "roofline": step_hbm_roofline("sfno", t),
            "native_forward_backward": type(eng.native).__name__, "parity": parity}


def secondary_tfno(steps, warmup, B=16, H=64, W=64):
    """BASELINE configs[3] / SURVEY 8(d): TFNO2dNet in 3, hidden 32, lifting 256, projection 64, 4 layers, n_modes
    (12, 12), group_norm, fft_norm forward; one training step = forward + MSE + backward + fused Adam."""
    import ppsci
    from paddlescience_amd import hotpath as hp

    torch.manual_seed(0)
    model = ppsci.arch.TFNO2dNet(("x",), ("y",), 12, 12, hidden_channels=32, in_channels=3, out_channels=1,
                                 lifting_channels=256, projection_channels=64, n_layers=4, norm="group_norm")
    x = torch.as_tensor(np.random.default_rng(42).standard_normal((B, 3, H, W)).astype(np.float32)).cuda()
    y = torch.as_tensor(np.random.default_rng(43).standard_normal((B, 1, H, W)).astype(np.float32)).cuda()
    opt = ppsci.optimizer.Adam(1e-3)(model)

    # parity (checker: the oracle's fp64 restatement of fno_block.py / tfnonet.py, pinned by tests/golden/fno.npz)
    from oracle import ref_torch as R

    P = {n: p.detach().cpu().double().requires_grad_(True) for n, p in torch.nn.Module.named_parameters(model)}
    xs, ys = x.cpu().double(), y.cpu().double()  # the FULL timed batch (16 x 64 x 64)
    yo = R.fno_forward(xs, P, 4, (12, 12), "group_norm")
    lo = ((yo - ys) ** 2).mean()
    names = sorted(P)
    go = dict(zip(names, torch.autograd.grad(lo, [P[n] for n in names])))
    nat = model.native()  # the path that is timed below
    mse_loss = ppsci.loss.MSELoss("mean")
    yh = nat.forward(x.contiguous())
    lh, gy = mse_loss.value_and_grad(yh, y, "y")  # csrc/field_loss.hip: value and dL/dy
    lh = lh["y"]
    yh = yh.clone()
    model.flat_grad.fill_(float("nan"))
    nat.backward(gy)
    gh = {n: p.grad.detach().cpu().numpy().copy() for n, p in torch.nn.Module.named_parameters(model)}
    model.flat_grad.zero_()
    parity = {"checker": "oracle/ref_torch.fno_forward fp64 (pinned by reference-run tests/golden/fno.npz), the whole "
                         "timed batch (16 x 64 x 64), the timed model's weights",
              "output_rel_l2": rel(yh.detach().cpu().numpy(), yo.detach().numpy()),
              "grad_rel_l2": max(rel(gh[n], go[n].numpy()) for n in names),
              "loss_rel": abs(float(lh.detach()) / float(lo.detach()) - 1.0)}

    # the training step as the Solver runs it: OperatorEngine -> native forward + hand-written backward
    # (paddlescience_amd/fno_engine.py), captured once into a HIP graph and replayed, then the fused Adam kernel
    from paddlescience_amd.operator_engine import OperatorConstraint, OperatorEngine

    cst = OperatorConstraint("Sup", model, {"y": lambda d: d["y"]}, mse_loss, x.device, ["y"], B)
    cst.bind({"x": x}, {"y": y})
    eng = OperatorEngine(model)

    from paddlescience_amd.engine import step_with_adam

    def step():  # (what Solver.train runs for this engine: the sums over the weight-gradient partials + Adam in one launch)
        step_with_adam(eng, [cst], opt, model.flat_params)

    t = time_wall(step, steps, warmup)
    # the same step timed with HIP events on the launch stream (device-side duration of the replayed graph + Adam): the
    # wall-clock figure of this ~60-node graph has been seen 3.6x higher on some boxes with identical kernel times
    t_ev = time_events(step, reps=steps)
    # the per-mode complex contraction alone (the kernel north_star reserves MFMA for), through the C ABI
    import ctypes as C

    from paddlescience_amd import _lib as L

    conv = model.fno_blocks.convs[0]
    d = L.SpectralDesc()
    d.batch, d.c_in, d.c_out, d.h, d.wf, d.modes_x, d.modes_y = B, 32, 32, H, W // 2 + 1, *conv.n_modes
    mx_, my_ = conv.n_modes
    x_ft = torch.randn(B, 32, mx_, my_, 2, device="cuda")  # kept-mode spectra, as the step holds them
    o_ft = torch.zeros_like(x_ft)
    st = hp._stream_ptr(x_ft)
    t_k = time_events(lambda: L.check(L.lib().ppsci_spectral_conv2d_fwd_kept(
        C.byref(d), hp._p(x_ft), hp._p(conv.weight_real), hp._p(conv.weight_imag), hp._p(o_ft), 1.0, st)))
    byts = 4.0 * (2 * 32 * 32 * 84 + 2 * 2 * B * 32 * 84)  # weights re+im, x_ft slice in, out slice out
    ach = byts / t_k / 1e12
    return {"config": "cfg4 TFNO-2D Darcy shape: 64x64 grid, batch 16, in 3, hidden 32, lifting 256, projection 64, "
                      "4 layers, n_modes (12,12), group_norm; forward + MSE + backward + Adam",
            "value": B * H * W / t, "unit": "grid-points/s", "samples_per_s": B / t, "ms_per_step": t * 1e3, "steps": steps,
            "ms_per_step_hip_events": t_ev * 1e3,
            "roofline": step_hbm_roofline("tfno", t),
            "spectral_contract_kernel": {"kernel_ms": t_k * 1e3, "operand_TBps": ach,
                                         "note": "84 modes x [16 x 64]x[64 x 64] real GEMM, 1.4 MB of operands, 11 MFLOP, one "
                                                 "workgroup per mode with its operands staged in LDS, on kept-mode spectra "
                                                 "(ppsci_dft2_kept_*): launch / latency-bound at this size"},
            "native_forward_backward": eng.native is not None, "parity": parity}


def secondary_spinn(tmp, steps, warmup, nc=128):
    import ppsci

    np.random.seed(111)
    model = ppsci.arch.SPINN(("x", "y", "z"), ("u",), 32, 4, 64, "tanh")
    eq = ppsci.equation.Helmholtz(3, 1.0)
    eq.model = model
    rng = np.random.default_rng(42)

    def run(n, timed):
        xs = [rng.uniform(-1, 1, (n, 1)).astype(np.float32) for _ in range(3)]
        uc = rng.standard_normal((n, n, n, 1)).astype(np.float32)
        data = {"x": xs[0], "y": xs[1], "z": xs[2], "uc": uc}
        lab = {"helmholtz": uc}
        pde = ppsci.constraint.SupervisedConstraint(
            {"dataset": {"name": "ContinuousNamedArrayDataset", "input": lambda: data, "label": lambda d: lab}},
            output_expr=eq.equations, loss=ppsci.loss.MSELoss("mean"), name="PDE")
        opt = ppsci.optimizer.Adam(1e-3)(model)
        solver = ppsci.solver.Solver(model, {"PDE": pde}, os.path.join(tmp, f"spinn{n}"), opt, epochs=1, iters_per_epoch=1)
        cc = solver._compiled["PDE"]
        cc.bind(data, lab)
        return solver, opt, cc, xs, uc

    if PURE:  # profiling run: nothing but the timed steps
        solver, opt, cc, xs, uc = run(nc, True)

        from paddlescience_amd.engine import step_with_adam

        def pure_step():
            step_with_adam(solver.engine, [cc], opt, model.flat_params)

        t = time_wall(pure_step, steps, warmup)
        return {"value": nc ** 3 / t, "ms_per_step": t * 1e3, "steps": steps}
    # parity on the FULL timed grid (checker: oracle fp64 restatement of spinn.py / helmholtz.py, pinned by
    # tests/golden/spinn.npz; the separable form makes 128^3 a few seconds of host time)
    from oracle import ref_torch as R

    solver, opt, cc, xs, uc = run(nc, False)
    solver.engine.forward_backward([cc])
    sd = {k: np.asarray(v.detach().cpu().numpy() if hasattr(v, "detach") else v, np.float64) for k, v in model.state_dict().items()}
    nets = []
    for b in range(3):
        P = {k.split(".", 2)[2]: v for k, v in sd.items() if k.startswith(f"branch_nets.{b}.")}
        nl = sum(1 for k in P if k.startswith("linears.") and k.endswith(".weight"))
        nets.append(R.ModifiedMLP1(dict(wu=P["embed_u.0.weight"], bu=P["embed_u.0.bias"], wv=P["embed_v.0.weight"],
                                        bv=P["embed_v.0.bias"], w=[P[f"linears.{l}.weight"] for l in range(nl)],
                                        b=[P[f"linears.{l}.bias"] for l in range(nl)], wl=P["last_fc.weight"],
                                        bl=P["last_fc.bias"]), "tanh"))
    xt = [torch.tensor(a.astype(np.float64), requires_grad=True) for a in xs]
    uo, ro = R.spinn_helmholtz(nets, xt, 1.0)
    lo = float(((ro - torch.tensor(uc[..., 0].astype(np.float64))) ** 2).mean().detach())
    pred = solver.predict({"x": xs[0], "y": xs[1], "z": xs[2]}, batch_size=None, return_numpy=True)["u"]
    parity = {"checker": f"oracle/ref_torch.spinn_helmholtz fp64 (pinned by reference-run tests/golden/spinn.npz), the whole "
                         f"{nc}^3 grid, the timed model's weights",
              "u_rel_l2": rel(pred[..., 0], uo.detach().numpy()), "loss_rel": abs(cc.loss() / lo - 1.0)}

    from paddlescience_amd.engine import step_with_adam

    def step():  # (what Solver.train runs for this engine: gradient rows + loss rows summed and Adam applied in one launch)
        step_with_adam(solver.engine, [cc], opt, model.flat_params)

    t = time_wall(step, steps, warmup)
    t_g = time_events(lambda: cc.forward(True))
    pts = nc ** 3
    ach = pts * 4 / t_g / 1e12
    return {"config": f"cfg5 SPINN Helmholtz3D: 3 x ModifiedMLP 1->64x4->32 tanh, {nc}^3 tensor-product grid, "
                      "k^2 u + u_xx + u_yy + u_zz - f, MSE-mean, Adam",
            "value": pts / t, "unit": "grid-points/s", "ms_per_step": t * 1e3, "steps": steps,
            "roofline": step_hbm_roofline("spinn", t),
            "grid_forward": {"kernel_ms": t_g * 1e3, "label_read_TBps": ach,
                             "note": "3 x modmlp_fwd + spinn_grid_fwd_mfma_kernel: 8.4 MB algorithmic read (the grid itself is "
                                     "never materialised): latency-bound at this size"},
            "parity": parity}


def extra_piratenet(tmp, steps, warmup, n=8192):
    """Not a BASELINE config: ppsci.arch.PirateNet at the configuration of examples/allen_cahn/conf/allen_cahn_piratenet.yaml
    (3 blocks x 256, periodic x, Fourier 256 scale 2, RWF), Allen-Cahn residual on 8192 points -- the layer-by-layer HIP path
    (csrc/pirate.hip + the MFMA 1x1-conv GEMMs), with parity of that path against the reference-run fixture."""
    import ppsci
    from tests.golden.make_piratenet_golden import CASES, equations

    gold = np.load(os.path.join(ROOT, "tests", "golden", "piratenet.npz"))
    name = "allen_cahn_rwf"
    c = CASES[name]
    small = ppsci.arch.PirateNet(c["inputs"], c["outputs"], c["blocks"], c["hidden"], c["act"], periods=c["periods"],
                                 fourier=c["fourier"], random_weight=c["rwf"])
    small.set_state_dict({k.split("/", 2)[2]: gold[k] for k in gold.files if k.startswith(f"{name}/param/")})
    X = gold[f"{name}/X"].astype(np.float32)
    inp = {k: X[:, j:j + 1] for j, k in enumerate(c["inputs"])}
    cfg = {"dataset": {"name": "IterableNamedArrayDataset", "input": inp,
                       "label": {"allen_cahn": gold[f"{name}/label/allen_cahn"][:, None].astype(np.float32)}}}
    cst = ppsci.constraint.SupervisedConstraint(cfg, ppsci.loss.MSELoss(c["reduction"]), equations(c), name="EQ")
    solver = ppsci.solver.Solver(small, {"EQ": cst}, os.path.join(tmp, "pirate_parity"), ppsci.optimizer.Adam(1e-3)(small),
                                 epochs=1, iters_per_epoch=1)
    solver.engine.forward_backward([solver._compiled["EQ"].fused])
    g = solver.engine.grad.cpu().numpy()
    gref = np.concatenate([gold[f"{name}/grad/{n_}"].ravel() for n_, _ in small.named_parameters()])
    res = solver.predict(inp, equations(c), batch_size=None, return_numpy=True)
    parity = {"reference": "tests/golden/piratenet.npz (the reference's own PirateNet, fp64, torch-backed paddle shim), case "
                           "allen_cahn_rwf: 2 blocks x 32, 40 points",
              "residual_rel_l2": rel(res["allen_cahn"][:, 0], gold[f"{name}/res/allen_cahn"]), "grad_rel_l2": rel(g, gref)}

    np.random.seed(1)
    model = ppsci.arch.PirateNet(("t", "x"), ("u",), 3, 256, "tanh", periods={"x": [2.0, False]},
                                 fourier={"dim": 256, "scale": 2.0}, random_weight={"mean": 1.0, "std": 0.1})
    eq = ppsci.equation.AllenCahn(eps=0.01)
    tx = np.random.default_rng(0).uniform([0, -1], [1, 1], (n, 2)).astype(np.float32)
    big = {"t": tx[:, 0:1], "x": tx[:, 1:2]}
    cfg = {"dataset": {"name": "IterableNamedArrayDataset", "input": big, "label": {"allen_cahn": np.zeros((n, 1), np.float32)}}}
    cst = ppsci.constraint.SupervisedConstraint(cfg, ppsci.loss.MSELoss("mean"), eq.equations, name="PDE")
    opt = ppsci.optimizer.Adam(1e-3)(model)
    # the TIMED model (3 x 256: two output-channel slabs per GEMM) on the first 256 points against the fp64 oracle
    # (oracle/ref_torch.PirateNet, pinned by the fixture above)
    from oracle import ref_torch as R

    state = {nm: v.detach().cpu().numpy().astype(np.float64) for nm, v in model.named_parameters()}
    om = R.PirateNet(("t", "x"), ("u",), state, "tanh", {"x": (2.0, False)})
    sub = {k: v[:256] for k, v in big.items()}
    ocst = dict(name="EQ", input={k: v.astype(np.float64) for k, v in sub.items()},
                exprs={"allen_cahn": R.allen_cahn_fn(0.01)}, label={"allen_cahn": np.zeros((256, 1))}, reduction="mean")
    _, _, go, oo = R.loss_and_grads(om, [ocst])
    cfg_s = {"dataset": {"name": "IterableNamedArrayDataset", "input": sub, "label": {"allen_cahn": np.zeros((256, 1), np.float32)}}}
    cst_s = ppsci.constraint.SupervisedConstraint(cfg_s, ppsci.loss.MSELoss("mean"), eq.equations, name="PDE")
    sol_s = ppsci.solver.Solver(model, {"PDE": cst_s}, os.path.join(tmp, "pirate_s"), opt, epochs=1, iters_per_epoch=1)
    sol_s.engine.forward_backward([sol_s._compiled["PDE"].fused])
    rs = sol_s.predict(sub, eq.equations, batch_size=None, return_numpy=True)
    parity["timed_model_vs_oracle"] = {"points": 256, "residual_rel_l2": rel(rs["allen_cahn"][:, 0], oo[0]["allen_cahn"].detach().numpy()[:, 0]),
                                       "grad_rel_l2": rel(sol_s.engine.grad.cpu().numpy(), go)}
    solver = ppsci.solver.Solver(model, {"PDE": cst}, os.path.join(tmp, "pirate"), opt, epochs=1, iters_per_epoch=1)
    fused = solver._compiled["PDE"].fused

    def step():
        solver.engine.forward_backward([fused])
        opt.step(solver.engine.grad)

    # (the first replays of this ~170-node graph after a launch-bound entry run at twice the steady time: enough warm-up)
    steps, warmup = max(steps, 30), max(warmup, 20)
    t = time_wall(step, steps, warmup)
    S, H, nb = fused.streams.S, 256, 3
    flops = 3 * 2 * (2 + 3 * nb) * H * H * S * n  # the H x H layers: forward + data gradient + weight gradient
    ach = flops / t / 1e12
    return {"config": "extra (not a BASELINE config): PirateNet 3 blocks x 256 tanh, periodic x, Fourier 256, RWF "
                      f"(allen_cahn_piratenet.yaml), Allen-Cahn residual (u_t, u_xx: S = {S} streams), {n} points, MSE-mean, Adam",
            "value": n / t, "unit": "points/s", "ms_per_step": t * 1e3, "steps": steps,
            "roofline": {"bound": "mfma", "kernel": "pw_conv_kernel / pw_wgrad_kernel (dense layers on all Taylor streams) over "
                                                    "the whole step", "achieved": ach, "peak": PEAK_FP32_TFLOPS,
                         "unit": "TFLOP/s", "frac": ach / PEAK_FP32_TFLOPS}, "parity": parity}


def extra_cylinder2d(tmp, steps, warmup):
    """Not a BASELINE config: the case behind the reference's published TIPC throughput (cylinder2d_unsteady_Re100, fp32, one
    unnamed NVIDIA GPU: ips = 1 264 165.641 points/s, /root/reference/test_tipc/README.MD:17) at the sizes of its yaml, built by
    examples/cylinder2d_unsteady.py (same-size synthetic point sets: the CSV files are not shipped).  `ips` of the reference =
    points of all constraints per iteration / batch_cost (train.py:106, printer.py:66) = `value` here."""
    from examples.cylinder2d_unsteady import DEFAULTS, build

    cfg = dict(DEFAULTS, output_dir=os.path.join(tmp, "cyl"), data_dir=os.path.join(tmp, "cyl_data"), epochs=1)
    solver = build(cfg)
    csts = [c.fused for c in solver._compiled.values()]
    n = sum(c.n for c in csts)

    def step():
        solver.engine.forward_backward(csts)
        solver.optimizer.step(solver.engine.grad)

    t = time_wall(step, max(steps, 20), max(warmup, 5))
    S = csts[0].streams.S
    p_mat = 3 * 64 + 4 * 64 * 64 + 64 * 3  # padded width 64
    pub = 1264165.641
    return {"config": "extra (not a BASELINE config): cylinder2d_unsteady_Re100 at the reference yaml's sizes -- MLP (t,x,y)->(u,v,p) "
                      f"5x50 tanh, unsteady NavierStokes, {n} points per iteration in 4 constraints (S = {S} streams for the PDE), "
                      "MSE-mean, Adam",
            "value": n / t, "unit": "points/s", "ms_per_step": t * 1e3, "points_per_iteration": n,
            "published_reference": {"value": pub, "unit": "points/s (TIPC ips, fp32, N1C1, unnamed NVIDIA GPU)",
                                    "source": "test_tipc/README.MD:17", "ratio": n / t / pub},
            "matrix_tflops_step": 6.0 * p_mat * S * csts[0].n / t / 1e12}


def extra_euler_beam(tmp, epochs=2000):
    """Not a BASELINE config: the reference's other published TIPC figure (euler_beam, fp32, one unnamed NVIDIA GPU: ips =
    3 667.54854, test_tipc/README.MD:18), examples/euler_beam.py at its yaml's sizes: 100 interior points with the fourth-order
    Biharmonic residual + the four boundary rows (one-row slices of the batch, lowered to a per-point weight mask), both on the
    fused kernels.  Timed: Solver.train() wall time / iterations."""
    from examples.euler_beam import DEFAULTS, build

    cfg = dict(DEFAULTS, output_dir=os.path.join(tmp, "beam"), epochs=20, log_freq=10 ** 9)
    build(cfg).train()  # warm-up: compilation, graph capture, allocator
    cfg["epochs"] = epochs
    solver = build(cfg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    solver.train()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / epochs
    n = cfg["batch_pde"] + cfg["batch_bc"]
    pub = 3667.54854
    return {"config": "extra (not a BASELINE config): euler_beam at the reference yaml's sizes -- MLP 1->20x3->1 tanh, 100 interior "
                      "points (u_xxxx + 1, 4th-order streams) + 4 boundary rows (u, u_x, u_xx, u_xxx at one point each), Adam",
            "value": n / t, "unit": "points/s", "ms_per_step": t * 1e3, "points_per_iteration": n,
            "published_reference": {"value": pub, "unit": "points/s (TIPC ips, fp32, N1C1, unnamed NVIDIA GPU)",
                                    "source": "test_tipc/README.MD:18", "ratio": n / t / pub},
            "final_loss": solver.last_losses.get("loss")}


# ------------------------------------------------------------------------------------------------------ main
def main():
    # stdout carries exactly ONE line, the JSON record: everything else this process prints (the ppsci logger of the API-level
    # entries writes to sys.stdout, as the reference's does) goes to stderr
    # (also at the file-descriptor level: RCCL / gloo / the HIP runtime print from C)
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    sys.stdout = sys.stderr
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the cfg 1/3/4/5 entries (N = 1 only anyway)")
    ap.add_argument("--no-strong", action="store_true", help="skip the 1 M-point NavierStokes strong-scaling entry")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU) through
        # torch.distributed.run on the loopback interface; rank 0's stdout (the one JSON line) is passed through
        import socket
        import subprocess

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        json_out.flush()
        sys.exit(subprocess.call(cmd, stdout=json_out.fileno()))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks (one rank per GPU)"
    if EMU:
        dev = torch.device("cpu")
        if world > 1:
            torch.distributed.init_process_group("gloo")
    else:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        if world > 1:
            torch.distributed.init_process_group("nccl", device_id=dev)
    comm_world = torch.distributed.get_world_size() if world > 1 else 1
    assert comm_world == world == args.gpus, "one RCCL rank per GPU"

    # the CPU-side checkers (oracle parity legs, cpu_baseline) run small-op graphs: 8 threads is the fastest setting on
    # the GPU box (tools/cpu_threads.py), and a 256-thread pool left spinning behind them slows the host side of the
    # timed GPU steps that follow (TFNO: 0.98 -> 3.6 ms per step when the pool was left at its default size)
    torch.set_num_threads(CPU_THREADS)
    if EMU:
        from tests.emu import build_emu

        build_emu.inject()
    from paddlescience_amd import hotpath as hp
    from paddlescience_amd.engine import Engine

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        _sync()

    flat = bench_weights(2, [WIDTH] * HIDDEN, 1)
    X = np.random.default_rng(42 + rank).uniform([0, -1], [1, 1], (N_PER_GPU, 2)).astype(np.float32)
    parity = parity_allen_cahn(dev, flat, X) if rank == 0 and not EMU else None
    # MSE-mean over the GLOBAL batch: the all-reduce is then a pure SUM
    lay, cst = allen_cahn_constraint(dev, X, N_PER_GPU * world)
    params = torch.tensor(flat, device=dev)
    eng = Engine(lay, params)
    assert eng.world == world

    p_mat = 2 * WIDTH + (HIDDEN - 1) * WIDTH * WIDTH + WIDTH  # matrix weights (P = 12 480)
    S = cst.streams.S
    flops_bwd = 4.0 * p_mat * S * N_PER_GPU   # reverse sweep: 2 GEMMs per layer  (F_T - F_R)
    flops_fwd = 2.0 * p_mat * S * N_PER_GPU   # F_R
    # the 100 000-point step at the INITIAL weights (gradient + loss, no update): `parity.full_batch` compares it with the
    # fp32 CPU port's first step on the same batch (cpu_baseline runs it anyway)
    eng.forward_backward([cst])
    _sync()
    g0, loss0 = eng.grad.detach().cpu().numpy().copy(), cst.losses()["allen_cahn"]
    eng.train_step([cst], 1e-3)  # (plans the step)
    fused = cst.one_launch_ready() and getattr(cst, "_step_kind", 0) == hp.STEP_FUSED_TILE and eng.one_launch
    # Timing.  `--steps K` is ONE window: K steps between barrier + synchronize; WINDOWS such windows
    # run back to back behind the W warm-up steps, `ms_per_step` is the MEDIAN window (min / max printed next to it: boxes
    # and clock states differ by several per cent over a 5 ms window).  The kernel-level figure of the roofline entry is
    # taken with HIP events on the launch stream BETWEEN the windows -- six launches of the main kernel alone after each --
    # i.e. in the same sustained clock state as the steps it is compared with, not on the clock ramp behind the idle phase
    # of the parity legs (round 5 printed kernel_ms 0.2329 > ms_per_step 0.2269 that way).
    # (The full garbage collection of quiet_host -- tens of ms of host time with the GPU idle -- sits in front of the warm-up.)
    WINDOWS = 1 if EMU else 10
    wins, main_samples = [], []
    with quiet_host():
        t_res = time_events(lambda: cst.forward(params, False), 30, median=True)
        for _ in range(args.warmup):
            eng.train_step([cst], 1e-3)
        for _ in range(WINDOWS):
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                eng.train_step([cst], 1e-3)
            barrier()
            wins.append(time.perf_counter() - t0)
            if fused and not EMU:
                main_samples += time_events_list(cst._step_plan.run_main, 6)
    tt = torch.tensor(wins, device=dev, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)  # every window: the slowest rank
    wins = sorted(tt.cpu().tolist())
    dt = wins[(len(wins) - 1) // 2]  # the median window (lower median for an even count)
    t_main = (float(np.median(main_samples)) if main_samples else time_events(cst._step_plan.run_main, 1)) if fused else None
    loss = cst.losses()["allen_cahn"]

    # (t_res: SURVEY.md 8(d) "R", residual evaluation only -- forward streams + epilogue, no stash, no adjoints; measured above)
    if fused:
        # the dominant kernel IS the step: forward + residual program + reverse of every tile in one kernel (F_T flops);
        # HIP events on the launch stream around that kernel alone (no weight split, no reduction kernels)
        kname = f"taylor_fused_kernel<4, {HIDDEN}, 2, 1, 0, {'true' if cst._step_plan.static_program else 'false'}>"
        flops_main = flops_fwd + flops_bwd
        ach = flops_main / t_main / 1e12
        extra = {"kernel_ms": t_main * 1e3, "flops_per_launch": flops_main,
                 "residual_program": ("compile-time table `%s` (csrc/epi_static_programs.h): evaluated by every wave" % cst._step_plan.static_program)
                 if cst._step_plan.static_program else "epilogue VM on wave 0",
                 "step_launches": "tile kernel + one tail kernel (sums, loss terms, Adam, next step's weight fragments)",
                 "what": "forward Taylor streams + residual program + loss seeds + reverse sweep of every 16-point tile in one "
                         "kernel: the activation stash never leaves the CU (registers), U / dL/dU live in LDS"}
        ksub = "taylor_fused_kernel<4, 4, 2, 1"
    else:
        # per-kernel timing of the dominant kernel (reverse sweep) with HIP events on the launch stream;
        # ppsci_taylor_bwd = the reverse kernel + the two small fixed-order reduction kernels behind it (~10 us)
        kname = "taylor_bwd_wx_kernel<4, 1, 4, 2, 1, 0>"
        t_fwd = time_events(lambda: hp.taylor_fwd(cst.desc, params, cst.inputs, cst.U, cst.stash))
        t_bwd = time_events(lambda: cst.backward(params))
        ach = flops_bwd / t_bwd / 1e12
        extra = {"kernel_ms": t_bwd * 1e3, "fwd_kernel_ms": t_fwd * 1e3, "fwd_achieved": flops_fwd / t_fwd / 1e12,
                 "flops_per_launch": flops_bwd}
        ksub = "taylor_bwd_wx_kernel<4, 1, 4, 2, 1"

    # HBM traffic of the dominant kernel per launch and of the whole step: quoted from the newest committed rocprofv3
    # summary (tools/profile_bench.sh; separate --pmc passes, FETCH_SIZE doubled per MI355X_MICROARCH.md) with its file name
    prof = profile_info("bench", ksub, dt / args.steps * 1e3, once_per_step=True)
    traffic, traffic_src = prof.get("kernel_hbm_bytes_per_launch"), prof.get("source")

    strong = None
    import tempfile

    tmp = tempfile.mkdtemp(prefix="ppsci_bench_")
    if not args.no_strong:
        try:
            strong = strong_ns(tmp, world, rank, 1 if EMU else max(5, args.steps // 5), 0 if EMU else max(2, args.warmup // 3),
                               barrier)
        except Exception as e:  # noqa: BLE001
            if world > 1:  # the other ranks are inside collectives: fail the job rather than hang it
                raise
            strong = {"error": f"{type(e).__name__}: {e}"[:300]}

    if rank == 0:
        out = {
            "metric": "collocation-points/sec (PDE residual+grad)",
            "value": N_PER_GPU * world * args.steps / dt,
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "ms_per_step_min": wins[0] / args.steps * 1e3, "ms_per_step_max": wins[-1] / args.steps * 1e3,
            "windows": len(wins),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if not EMU else "synthetic -- EMULATOR TEST MODE (CPU, gloo, tiny sizes): not a measurement",
            "config": {"workload": "Allen-Cahn 1D+t, MLP 2->64x4->1 tanh, 100k collocation pts per GPU, "
                                   "residual+MSE-mean+grad+Adam (BASELINE.json configs[1])",
                       "points_per_gpu": N_PER_GPU, "parallelism": f"dp{world}", "loss": loss,
                       "residual_only_points_per_s_per_gpu": N_PER_GPU / t_res},
            "roofline": {"bound": "mfma", "kernel": kname, "achieved": ach,
                         "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP32_TFLOPS,
                         **pipe_roofline(ach, "bench", ksub),
                         "traffic": traffic, "traffic_source": traffic_src, **extra,
                         "algorithmic_bytes_per_launch": 4.0 * (2 + 1 + 1) * N_PER_GPU,
                         "arithmetic": "fp32 operands as three bf16 terms, six v_mfma_f32_16x16x32_bf16 products per "
                                       "K = 32 step, fp32 accumulate (error <= the fp32 MFMA's); `peak` is the fp32-input "
                                       "MFMA peak the algorithmic flops are priced against, `pipe_peak` the ceiling of the "
                                       "pipe they actually run on",
                         "residual_error_note": "residual rel-L2 vs the fp64 reference run is ~1.1e-6 = 2.5x the fp32 floor of "
                                                "the reference algorithm itself (4.4e-7, SURVEY 8d): the 5-slot tanh "
                                                "(1 - 2/(exp(2x)+1), abs error 1.2e-7), not the bf16 split",
                         "step_profile": prof},
            "parity": parity,
        }
        if strong is not None:
            out["strong_scaling"] = strong
            if "error" not in strong:
                # the north star's multi-GPU target is STRONG scaling of configs[2] (1 M NavierStokes points over the ranks);
                # the numbers also sit in `config`, which every consumer of this line keeps
                out["config"].update({
                    "strong_workload": "LDC NavierStokes 2-D steady, MLP 2->128x5->3 tanh, 1 000 000 points over the ranks (configs[2])",
                    "strong_points_per_s": strong["value"], "strong_ms_per_step": strong["ms_per_step"],
                    "strong_points_per_rank": strong["points_per_rank"], "strong_allreduce_ms": strong["allreduce_ms"],
                    "strong_allreduce_bytes": strong["allreduce_bytes"], "comm_world_size": strong["comm_world_size"],
                    "comm_backend": strong["comm_backend"]})
        # the primary (weak-scaling) step under data parallelism: tile kernel + tail kernel, the SUM all-reduce of the flat
        # gradient on the launch stream, ONE launch for Adam + the next step's weight fragments (engine.Engine.train_step)
        out["config"]["dp_step"] = ("fused tile kernel + tail kernel (sums) -> all-reduce (%d B) -> Adam + fragments kernel"
                                    % (int(eng.grad.numel()) * 4)) if world > 1 else "fused tile kernel + tail kernel (sums, Adam, fragments)"
        if world == 1 and not args.no_cpu_baseline and not EMU:
            out["cpu_baseline"] = cpu_baseline("allen_cahn", flat, X)
            out["speedup_vs_cpu_best_thread"] = out["value"] / out["cpu_baseline"]["value"]
            l_cpu, g_cpu = out["cpu_baseline"].pop("_first_step")
            out["parity"]["full_batch"] = {
                "points": N_PER_GPU, "against": "the fp32 torch-CPU port's first step (oracle/ref_torch.py, reverse-over-reverse), "
                                                "same batch, initial weights",
                "loss_rel": abs(loss0 / l_cpu - 1.0), "grad_rel_l2": rel(g0, g_cpu),
                "note": "both sides are fp32 sums over 100 000 points in different orders: the comparison carries the port's "
                        "own rounding (the fp64-anchored figures are the 2 048-point ones above)"}
        if world == 1 and not args.no_secondary and not EMU:
            k, w = max(10, args.steps // 2), max(3, args.warmup // 2)
            sec = []
            for fn in (lambda: secondary_laplace(tmp, 4 * k, w, not args.no_cpu_baseline),
                       lambda: secondary_ns(tmp, k, w), lambda: secondary_ac256(tmp, k, w), lambda: secondary_tfno(k, w), lambda: extra_uno(k, w), lambda: extra_sfno(k, w),
                       lambda: secondary_spinn(tmp, k, w),
                       lambda: extra_piratenet(tmp, k, w), lambda: extra_cylinder2d(tmp, k, w),
                       lambda: extra_euler_beam(tmp)):
                try:
                    sec.append(fn())
                except Exception as e:  # noqa: BLE001 -- a secondary entry must not cost the primary line
                    sec.append({"error": f"{type(e).__name__}: {e}"[:300]})
            out["secondary"] = sec
        print(json.dumps(out), file=json_out, flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
