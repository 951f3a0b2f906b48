"""Times taylor_fwd / taylor_bwd on wide nets with the feature-split kernels on (knob 8) and off (knob 16)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from paddlescience_amd import _lib as L  # noqa: E402
from paddlescience_amd import hotpath as hp  # noqa: E402


def timeit(fn, reps=8):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(reps)]))


def main():
    dev = "cuda"
    shapes = [("NS 5x128 S5 125k", 2, 5, 128, 3, [[1.0, 0.0], [0.0, 1.0]], 2, 125_000),
              ("AC 4x256 S4 100k", 2, 4, 256, 1, [[0.0, 1.0], [1.0, 0.0]], 1, 100_000)]
    if "--ac64" in sys.argv:  # the primary bench shape on the single-wave (knob 8) and feature-split (knob 4) kernels
        shapes = [("AC 4x64 S4 100k", 2, 4, 64, 1, [[0.0, 1.0], [1.0, 0.0]], 1, 100_000)]
    ns_only = "--ns-only" in sys.argv  # profiling runs: the BASELINE config 3 shard on the default kernels only
    if ns_only:
        shapes = shapes[:1]
    for (label, d_raw, nh, w, m, dirs, n2, N) in shapes:
        for knob in ((8,) if ns_only else ((4, 8) if "--ac64" in sys.argv else (8, 16))):
            L.lib().ppsci_set_wide_min_nb(knob)
            try:
                lay = hp.NetLayout(d_raw, nh, w, m, "tanh")
                spec = hp.StreamSpec(dirs, n2)
                desc = lay.desc(spec)
                params = (torch.rand(lay.n_params, device=dev) - 0.5) * 0.2
                xs = [torch.rand(N, device=dev) for _ in range(d_raw)]
                U = torch.zeros((m * spec.S, N), device=dev)
                stash = torch.zeros(hp.stash_bytes(desc, N) // 4, device=dev)
                f = timeit(lambda: hp.taylor_fwd(desc, params, xs, U, stash))
                out = {"shape": label, "wide_min_nb": knob, "fwd_ms": round(f, 4)}
                P = d_raw * w + (nh - 1) * w * w + w * m
                out["fwd_TF"] = round(2.0 * P * spec.S * N / f / 1e9, 1)
                rows = hp.bwd_partial_rows(desc, N)
                if rows > 0:
                    Ubar = torch.randn((m * spec.S, N), device=dev)
                    gp = torch.zeros((rows, lay.n_params), device=dev)
                    ws = torch.zeros(max(4, hp.bwd_workspace_bytes(desc, N) // 4), device=dev)
                    b = timeit(lambda: hp.taylor_bwd(desc, params, xs, Ubar, stash, ws, gp))  # incl. the reduce kernels
                    out["bwd_main_ms"] = round(b, 4)
                    out["bwd_TF"] = round(4.0 * P * spec.S * N / b / 1e9, 1)
                print(json.dumps(out), flush=True)
            except Exception as e:  # noqa: BLE001
                print(json.dumps({"shape": label, "wide_min_nb": knob, "error": str(e)[:120]}), flush=True)
    L.lib().ppsci_set_wide_min_nb(8)


if __name__ == "__main__":
    main()
