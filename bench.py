"""bench.py -- collocation-points/sec of the PINN hot path (PDE residual + gradient + Adam).

Workload (BASELINE.json configs[1]): Allen-Cahn 1D+t, MLP 2->64x4->1 tanh, 100 000 collocation
points per GPU, residual u_t - eps^2 u_xx + 5u^3 - 5u (eps = 0.01), MSE-mean, Adam.  Synthetic
points `default_rng(42+rank).uniform([0,-1],[1,1])`, Xavier-uniform weights `default_rng(1234)`
(SURVEY.md 8d).  A "step" = taylor_fwd + epilogue + taylor_bwd + gradient reduce (+ RCCL all-reduce
of the flat gradient when N > 1) + fused Adam, inputs resident in HBM.

One process per GPU (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`);
per-GPU work is fixed => weak scaling.  Rank 0 prints ONE JSON line.
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
PEAK_FP32_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 MFMA (f32 in) peak == fp32 vector peak
EPS = 0.01


def build_constraint(dev, rank):
    from oracle import taylor_np as T  # only for the seeded weight draw shared with the CPU baseline
    from paddlescience_amd import _lib as L
    from paddlescience_amd import hotpath as hp
    from paddlescience_amd.engine import Engine, FusedConstraint

    net = T.make_net(2, [WIDTH] * HIDDEN, 1, seed=1234)
    lay = hp.NetLayout(2, HIDDEN, WIDTH, 1, "tanh")
    X = np.random.default_rng(42 + rank).uniform([0, -1], [1, 1], (N_PER_GPU, 2)).astype(np.float32)
    xs = [torch.tensor(X[:, j].copy(), device=dev) for j in range(2)]  # (t, x)
    streams = hp.StreamSpec([[0.0, 1.0], [1.0, 0.0]], 1)  # streams: u, u_x, u_t, u_xx
    pr = hp.Program(4, 2)
    u, ut, uxx = pr.ld_u(0), pr.ld_u(2), pr.ld_u(3)
    five = pr.const(5.0)
    r = pr.op(L.OP_SUB,
              pr.op(L.OP_ADD, pr.op(L.OP_SUB, ut, pr.op(L.OP_MUL, pr.const(EPS**2), uxx)),
                    pr.op(L.OP_MUL, pr.op(L.OP_MUL, pr.op(L.OP_MUL, five, u), u), u)),
              pr.op(L.OP_MUL, five, u))
    return net, lay, X, xs, streams, pr, r


CPU_THREADS = 8  # measured on the GPU box (tools/cpu_threads.py): 8 threads is the fastest setting for this
                 # graph of small ops; 32+ threads are slower, 256 threads 100x slower


def cpu_baseline(net, X, steps=10):
    """The oracle's restatement of the reference algorithm (reverse-over-reverse autodiff, fp32,
    torch-CPU), timed on the same batch: residual + MSE + backward + Adam."""
    from oracle import ref_torch as R

    torch.set_num_threads(min(CPU_THREADS, os.cpu_count()))
    model = R.MLP(("t", "x"), ("u",), net, dtype=torch.float32)
    n = X.shape[0]
    cst = dict(name="EQ", input={"t": X[:, :1], "x": X[:, 1:]}, exprs={"allen_cahn": R.allen_cahn_fn(EPS)},
               label={"allen_cahn": np.zeros((n, 1), np.float32)}, reduction="mean")
    opt = R.Adam(sum(p.numel() for p in model.parameters()), 1e-3, dtype=np.float32)
    flat = np.concatenate([p.detach().numpy().ravel() for p in model.parameters()])
    times = []
    for i in range(steps + 1):
        t0 = time.perf_counter()
        _, _, g, _ = R.loss_and_grads(model, [cst])
        flat = opt.step(flat, g).astype(np.float32)
        off = 0
        with torch.no_grad():
            for p in model.parameters():
                k = p.numel()
                p.copy_(torch.from_numpy(flat[off:off + k].reshape(p.shape)))
                off += k
        if i > 0:
            times.append(time.perf_counter() - t0)
    return n / float(np.median(times)), len(times)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        torch.distributed.init_process_group("nccl", device_id=dev)

    from paddlescience_amd import hotpath as hp
    from paddlescience_amd.engine import Engine, FusedConstraint
    from oracle import taylor_np as T

    net, lay, X, xs, streams, pr, r = build_constraint(dev, rank)
    # MSE-mean over the GLOBAL batch: the all-reduce is then a pure SUM
    pr.residual(r, scale=1.0 / (N_PER_GPU * world))
    cst = FusedConstraint("EQ", lay, streams, pr.build(), xs, [], ["allen_cahn"])
    params = torch.tensor(T.flat_params(net), dtype=torch.float32, device=dev)
    eng = Engine(lay, params)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.train_step([cst], 1e-3)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.train_step([cst], 1e-3)
    barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
    dt = float(tt[0])
    loss = cst.losses()["allen_cahn"]

    # per-kernel timing of the dominant kernel (reverse sweep) with HIP events on the launch stream
    def time_kernel(fn, reps=20):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        ev[0].record()
        for i in range(reps):
            fn()
            ev[i + 1].record()
        torch.cuda.synchronize()
        return float(np.mean([ev[i].elapsed_time(ev[i + 1]) for i in range(reps)])) * 1e-3

    from paddlescience_amd import _lib

    t_fwd = time_kernel(lambda: hp.taylor_fwd(cst.desc, params, cst.inputs, cst.U, cst.stash))
    _lib.lib().ppsci_set_bwd_main_only(1)  # time the dominant kernel alone (not its two small reduce kernels)
    t_bwd = time_kernel(lambda: cst.backward(params))
    _lib.lib().ppsci_set_bwd_main_only(0)
    # SURVEY.md 8(d) "R": residual evaluation only (forward streams + epilogue, no stash, no adjoints)
    t_res = time_kernel(lambda: cst.forward(params, False))
    p_mat = 2 * WIDTH + (HIDDEN - 1) * WIDTH * WIDTH + WIDTH  # matrix weights (P = 12 480)
    S = streams.S
    flops_bwd = 4.0 * p_mat * S * N_PER_GPU   # reverse sweep: 2 GEMMs per layer  (F_T - F_R)
    flops_fwd = 2.0 * p_mat * S * N_PER_GPU   # F_R
    ach = flops_bwd / t_bwd / 1e12

    # HBM traffic of the dominant kernel per launch: PMC counters (FETCH_SIZE x2 + WRITE_SIZE, KiB) collected by
    # tools/profile_bench.sh in separate rocprofv3 --pmc passes of this same command and condensed by
    # tools/summarize_profile.py into profiles/*_pmc_summary.json (a live bench run cannot read PMCs itself)
    traffic = None
    try:
        import glob

        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_bench_pmc_summary.json")))
        if files:
            for k, v in json.load(open(files[-1])).items():
                if k.startswith("void taylor_bwd_kernel<4, 2, 1, 0"):
                    traffic = v.get("hbm_bytes_per_launch")
    except Exception:  # noqa: BLE001
        traffic = None

    if rank == 0:
        out = {
            "metric": "collocation-points/sec (PDE residual+grad)",
            "value": N_PER_GPU * world * args.steps / dt,
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "Allen-Cahn 1D+t, MLP 2->64x4->1 tanh, 100k collocation pts per GPU, "
                                   "residual+MSE-mean+grad+Adam (BASELINE.json configs[1])",
                       "points_per_gpu": N_PER_GPU, "parallelism": f"dp{world}", "loss": loss,
                       "residual_only_points_per_s_per_gpu": N_PER_GPU / t_res},
            "roofline": {"bound": "mfma", "kernel": "taylor_bwd_kernel<4, 2, 1, 0>", "achieved": ach,
                         "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP32_TFLOPS,
                         "traffic": traffic, "kernel_ms": t_bwd * 1e3,
                         "fwd_kernel_ms": t_fwd * 1e3, "fwd_achieved": flops_fwd / t_fwd / 1e12},
        }
        if world == 1 and not args.no_cpu_baseline:
            v, k = cpu_baseline(net, X)
            out["cpu_baseline"] = {"value": v, "unit": "points/s", "cores": min(CPU_THREADS, os.cpu_count()),
                                   "kind": "port",
                                   "sample": f"same 100k-point batch, median of {k} full training steps of the "
                                             "torch-CPU fp32 reverse-over-reverse restatement (oracle/ref_torch.py), "
                                             f"{min(CPU_THREADS, os.cpu_count())} threads of {os.cpu_count()} host cores"}
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
