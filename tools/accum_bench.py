"""Times the Allen-Cahn bench step and its reverse kernel with the workgroup-level LDS accumulation of the
hidden-weight gradient on (default) and off (per-tile streaming + tree reduction), on the GPU box."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from paddlescience_amd import _lib as L  # noqa: E402
from paddlescience_amd.engine import Engine  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    flat = bench.bench_weights(2, [64] * 4, 1)
    X = np.random.default_rng(42).uniform([0, -1], [1, 1], (100_000, 2)).astype(np.float32)
    grads = {}
    for mode in (1, 0, 1):
        L.lib().ppsci_set_bwd_accum(mode)
        lay, cst = bench.allen_cahn_constraint(dev, X, 100_000)
        params = torch.tensor(flat, device=dev)
        eng = Engine(lay, params)
        eng.forward_backward([cst])
        torch.cuda.synchronize()
        grads[mode] = eng.grad.cpu().numpy().copy()
        t_step = bench.time_wall(lambda: eng.train_step([cst], 1e-3), 50, 10)
        t_bwd_all = bench.time_events(lambda: cst.backward(params))
        bench.main_kernel_only(True)
        t_bwd = bench.time_events(lambda: cst.backward(params))
        bench.main_kernel_only(False)
        print(json.dumps({"accum": mode, "ms_per_step": t_step * 1e3, "bwd_call_ms": t_bwd_all * 1e3,
                          "bwd_main_kernel_ms": t_bwd * 1e3, "ws_MB": cst.workspace.numel() * 4 / 1e6}), flush=True)
    print(json.dumps({"grad_rel_diff_accum_vs_stream": bench.rel(grads[1], grads[0])}))


if __name__ == "__main__":
    main()
