"""Times taylor_fwd / taylor_bwd of every library in build/variants on the Allen-Cahn bench shape
(run on the GPU box).  Ablation variants compute wrong numbers by design; only time matters."""
import ctypes as C
import glob
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from paddlescience_amd import _lib as L  # noqa: E402
from paddlescience_amd import hotpath as hp  # noqa: E402


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(reps)]
    return float(np.median(ts)), float(np.min(ts))


def main():
    shapes = [("AC 4x64 S4 100k", 2, 4, 64, 1, [[0.0, 1.0], [1.0, 0.0]], 1, 100_000)]
    if "--more" in sys.argv:
        shapes += [("LAP 3x20 S5 10k", 2, 3, 20, 1, [[1.0, 0.0], [0.0, 1.0]], 2, 10_201),
                   ("NS 5x128 S5 125k", 2, 5, 128, 3, [[1.0, 0.0], [0.0, 1.0]], 2, 125_000)]
    if "--wide" in sys.argv:  # width 256 (feature-split kernels): one tile per CU, half a wave of tiles, full load
        shapes = [(f"AC 4x256 S4 {n}", 2, 4, 256, 1, [[0.0, 1.0], [1.0, 0.0]], 1, n) for n in (4096, 100_000)]
    libs = sorted(glob.glob(os.path.join(ROOT, "variants_out", "*.so")))
    only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--only=")]
    if only:
        libs = [p for p in libs if os.path.basename(p)[:-3] in only[0].split(",")]
    dev = "cuda"
    for path in libs:
        name = os.path.basename(path)[:-3]
        L._lib = L._bind(path)
        for (label, d_raw, nh, w, m, dirs, n2, N) in shapes:
          try:
            lay = hp.NetLayout(d_raw, nh, w, m, "tanh")
            spec = hp.StreamSpec(dirs, n2)
            desc = lay.desc(spec)
            params = (torch.rand(lay.n_params, device=dev) - 0.5) * 0.3
            xs = [torch.rand(N, device=dev) for _ in range(d_raw)]
            U = torch.zeros((m * spec.S, N), device=dev)
            Ubar = torch.randn((m * spec.S, N), device=dev)
            stash = torch.zeros(hp.stash_bytes(desc, N) // 4, device=dev)
            rows = hp.bwd_partial_rows(desc, N)
            gp = torch.zeros((rows, lay.n_params), device=dev)
            f_med, f_min = timeit(lambda: hp.taylor_fwd(desc, params, xs, U, stash))
            fn_med, fn_min = timeit(lambda: hp.taylor_fwd(desc, params, xs, U, None))
            grads = {}
            for accum in (1, 0):
                L._lib.ppsci_set_bwd_accum(accum)
                ws = torch.full((max(4, hp.bwd_workspace_bytes(desc, N) // 4),), float("nan"), device=dev)
                b_med, b_min = timeit(lambda: hp.taylor_bwd(desc, params, xs, Ubar, stash, ws, gp))
                grads[accum] = gp.clone()
                L._lib.ppsci_set_bwd_main_only(1)
                m_med, m_min = timeit(lambda: hp.taylor_bwd(desc, params, xs, Ubar, stash, ws, gp))
                L._lib.ppsci_set_bwd_main_only(0)
                try:
                    import ctypes
                    raw = ctypes.CDLL(path)
                    buf = (ctypes.c_ulonglong * 8)()
                    raw.ppsci_bwd_read_phase_timers(buf, 1)
                    L._lib.ppsci_set_bwd_main_only(1)
                    hp.taylor_bwd(desc, params, xs, Ubar, stash, ws, gp)
                    torch.cuda.synchronize()
                    L._lib.ppsci_set_bwd_main_only(0)
                    raw.ppsci_bwd_read_phase_timers(buf, 1)
                    tot = float(sum(buf)) or 1.0
                    print(json.dumps({"variant": name, "accum": accum,
                                      "phase_cycles_per_tile": [round(v / ((N + 15) // 16)) for v in buf],
                                      "phase_pct": [round(100.0 * v / tot, 1) for v in buf]}), flush=True)
                except AttributeError:
                    pass
                print(json.dumps({"variant": name, "accum": accum, "shape": label, "rows": rows, "fwd_ms": round(f_med, 4),
                                  "fwd_nostash_ms": round(fn_med, 4), "bwd_ms": round(b_med, 4),
                                  "bwd_min_ms": round(b_min, 4), "bwd_main_ms": round(m_med, 4)}), flush=True)
            L._lib.ppsci_set_bwd_accum(1)
            print(json.dumps({"variant": name, "grad_rel_diff_between_modes":
                              float((grads[1] - grads[0]).norm() / grads[1].norm())}), flush=True)
          except Exception as e:  # noqa: BLE001
            print(json.dumps({"variant": name, "shape": label, "error": str(e)[:100]}), flush=True)


if __name__ == "__main__":
    main()
