"""Latency of the width-256 kernels on launches that cannot fill the chip (reference Allen-Cahn batches:
4 096 PDE points, 512 initial-condition points).  Used for the eight-waves-per-tile experiment (DESIGN 3a'):
forward 84 -> 79 us, reverse 273 -> 276 us at 4 096 points, i.e. the per-tile latency is not per-wave MFMA work."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from paddlescience_amd import hotpath as hp  # noqa: E402
from tools.wide_bench import timeit  # noqa: E402


def main():
    dev = "cuda"
    for (label, dirs, n2, N) in [("4x256 S4", [[0.0, 1.0], [1.0, 0.0]], 1, 4096), ("4x256 S1", [], 0, 512),
                                 ("4x256 S4", [[0.0, 1.0], [1.0, 0.0]], 1, 2048), ("4x256 S4", [[0.0, 1.0], [1.0, 0.0]], 1, 8192)]:
        lay = hp.NetLayout(2, 4, 256, 1, "tanh")
        spec = hp.StreamSpec(dirs, n2)
        desc = lay.desc(spec)
        params = (torch.rand(lay.n_params, device=dev) - 0.5) * 0.2
        xs = [torch.rand(N, device=dev) for _ in range(2)]
        U = torch.zeros((spec.S, N), device=dev)
        stash = torch.zeros(hp.stash_bytes(desc, N) // 4, device=dev)
        f = timeit(lambda: hp.taylor_fwd(desc, params, xs, U, stash), reps=20)
        rows = hp.bwd_partial_rows(desc, N)
        Ubar = torch.randn((spec.S, N), device=dev)
        gp = torch.zeros((rows, lay.n_params), device=dev)
        ws = torch.zeros(max(4, hp.bwd_workspace_bytes(desc, N) // 4), device=dev)
        b = timeit(lambda: hp.taylor_bwd(desc, params, xs, Ubar, stash, ws, gp), reps=20)
        print(json.dumps({"shape": label, "N": N, "fwd_ms": round(f, 4), "bwd_ms": round(b, 4)}), flush=True)


if __name__ == "__main__":
    main()
