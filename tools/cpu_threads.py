"""Sweep torch-CPU thread counts for the oracle training step (sizes bench.py's cpu_baseline leg)."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_torch as R, taylor_np as T
net = T.make_net(2, [64] * 4, 1, seed=1234)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
X = np.random.default_rng(42).uniform([0, -1], [1, 1], (n, 2)).astype(np.float32)
for th in [1, 8, 16, 32, 64, 128]:
    if th > os.cpu_count():
        continue
    torch.set_num_threads(th)
    model = R.MLP(("t", "x"), ("u",), net, dtype=torch.float32)
    cst = dict(name="EQ", input={"t": X[:, :1], "x": X[:, 1:]}, exprs={"allen_cahn": R.allen_cahn_fn(0.01)},
               label={"allen_cahn": np.zeros((n, 1), np.float32)}, reduction="mean")
    ts = []
    for i in range(3):
        t0 = time.perf_counter(); R.loss_and_grads(model, [cst]); ts.append(time.perf_counter() - t0)
    print(th, "threads:", n / min(ts), "pts/s", flush=True)
