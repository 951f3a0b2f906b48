"""Determinism probe 2: which launch geometry shows it, and whether recycled (poisoned) device memory matters."""
import os
import numpy as np
import torch
from paddlescience_amd import device, hotpath as hp
from tests.test_fused_step import _run, _weights

d = device.get_device()
lay = hp.NetLayout(2, 4, 64, 1, "tanh")
flat = _weights(lay, 3)
segs, off = [], 0
for name, shp in lay.param_shapes():
    k = int(np.prod(shp))
    segs.append((name.replace("linears.", "L").replace("weight", "w").replace("bias", "b"), off, off + k))
    off += k


def poison(val):
    if val is None:
        return
    torch.cuda.synchronize()
    t = torch.full((64 << 20,), val, device=d)  # 256 MiB: everything the caching allocator hands out next is this
    torch.cuda.synchronize()
    del t


def summary(a, b):
    parts = []
    for name, lo, hi in segs:
        x, y = a[1][0][lo:hi], b[1][0][lo:hi]
        nd = int((x != y).sum())
        nan = int(np.isnan(y).sum())
        if nd or nan:
            parts.append(f"{name}:{nd}" + (f"(nan {nan})" if nan else ""))
    return " ".join(parts) or "identical"


def exp(tag, n, max_grid, static, tail, pois, runs=4):
    rs = []
    for _ in range(runs):
        poison(pois)
        rs.append(_run(d, lay, [("allen_cahn", n)], flat, True, 1, max_grid=max_grid, tail=tail, static_program=static))
    print(f"[{tag}] n={n} grid={max_grid} static={static} tail={tail} poison={pois}: " + " | ".join(summary(rs[0], r) for r in rs[1:]), flush=True)
    return rs[0]


for static in (1, 0):
    base = exp("base", 20000, 0, static, 3, None)
    exp("grid1", 20000, 1, static, 3, None)
    exp("1tile/wg", 8192, 0, static, 3, None)
    exp("1tile", 16, 0, static, 3, None)
    z = exp("zero", 20000, 0, static, 3, 0.0)
    q = exp("nan", 20000, 0, static, 3, float("nan"))
    o = exp("one", 20000, 0, static, 3, 1.0)
    print("  zero-run0 vs nan-run0:", summary(z, q), "; vs one-run0:", summary(z, o), "; base-run0 vs zero-run0:", summary(base, z), flush=True)
    exp("zero tail1", 20000, 0, static, 1, 0.0)
