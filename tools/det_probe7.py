"""Determinism probe 7 (variant dbg_dump): zbar_3, its row sums and hbar_2 per workgroup, compared between runs."""
import ctypes as C
import os
import numpy as np
import torch
from paddlescience_amd import _lib as L, device, hotpath as hp
from paddlescience_amd.engine import Engine
from tests.test_one_launch import _constraint, _weights

d = device.get_device()
lay = hp.NetLayout(2, 4, 64, 1, "tanh")
flat = _weights(lay, 3)
lib = L.lib()
raw = C.CDLL(os.environ["PPSCI_HIP_LIB"])
raw.ppsci_dbg_set.argtypes = [C.c_void_p]
n, grid, stride = 8192, 512, 16384
buf = torch.zeros(grid * stride, device=d)
assert raw.ppsci_dbg_set(buf.data_ptr()) == 0


def run():
    lib.ppsci_set_step_tail(1)
    buf.zero_()
    params = torch.tensor(flat, device=d)
    eng = Engine(lay, params)
    eng.one_launch = True
    c = _constraint(d, "allen_cahn", lay, n, 100)
    eng.train_step([c], 1e-2)
    torch.cuda.synchronize()
    assert c._step_plan.static_program != ""
    return buf.cpu().numpy().reshape(grid, stride).copy(), eng.grad.cpu().numpy().copy()


runs = [run() for _ in range(8)]
for k in range(1, 8):
    a, b = runs[0][0], runs[k][0]
    print(f"run0 vs run{k}: grads differ {int((runs[0][1] != runs[k][1]).sum())}")
    for name, lo, hi in (("zbar3", 0, 4096), ("rowsum", 4096, 5120), ("D2D3 early", 5120, 7168), ("D2D3D1 late", 12288, 15360), ("hbar2", 8192, 12288)):
        wg = np.flatnonzero((a[:, lo:hi] != b[:, lo:hi]).any(1))
        msg = f"   {name}: workgroups differing {wg.tolist()[:8]}"
        for w in wg[:2]:
            x, y = a[w, lo:hi], b[w, lo:hi]
            idx = np.flatnonzero(x != y)
            if name != "rowsum":
                s_, t_, r_ = idx // 1024, (idx % 1024) // 4, idx % 4
                msg += f"\n      wg {w}: {idx.size} values; streams {sorted(set(s_.tolist()))} waves {sorted(set((t_ // 64).tolist()))} lanes {sorted(set((t_ % 64).tolist()))[:20]} r {sorted(set(r_.tolist()))}; e.g. {x[idx[0]]:.7e} vs {y[idx[0]]:.7e}"
            else:
                t_, r_ = idx // 4, idx % 4
                msg += f"\n      wg {w}: {idx.size} values; waves {sorted(set((t_ // 64).tolist()))} lanes {sorted(set((t_ % 64).tolist()))[:20]} r {sorted(set(r_.tolist()))}; e.g. {x[idx[0]]:.7e} vs {y[idx[0]]:.7e}"
        print(msg)
