"""Run-to-run determinism probe of the fused step: one step, several runs; which parameter segments / outputs differ."""
import os
import sys
import numpy as np
from paddlescience_amd import device, hotpath as hp
from tests.test_fused_step import _run, _weights

d = device.get_device()
lay = hp.NetLayout(2, 4, 64, 1, "tanh")
flat = _weights(lay, 3)
n = int(os.environ.get("PROBE_N", "20000"))
runs_n = int(os.environ.get("PROBE_RUNS", "6"))
segs, off = [], 0
for name, shp in lay.param_shapes():
    k = int(np.prod(shp))
    segs.append((name, off, off + k))
    off += k
for static in (1, 0):
    for tail in (3, 1):
        runs = [_run(d, lay, [("allen_cahn", n)], flat, True, 1, tail=tail, static_program=static) for _ in range(runs_n)]
        a = runs[0]
        for k in range(1, runs_n):
            b = runs[k]
            parts = []
            for name, lo, hi in segs:
                x, y = a[1][0][lo:hi], b[1][0][lo:hi]
                nd = int((x != y).sum())
                if nd:
                    parts.append(f"{name}:{nd}/{hi - lo} max|d|/max|x| {np.abs(x - y).max() / np.abs(x).max():.1e}")
            outs = [nm for nm, i in (("resid", 3), ("U", 4), ("Ubar", 5)) if not np.array_equal(a[i][0], b[i][0])]
            print(f"static={static} tail={tail} run0 vs run{k}: loss eq {np.array_equal(a[2][0][0], b[2][0][0])}; outputs differing {outs}; " + ("; ".join(parts) or "gradients identical"), flush=True)
