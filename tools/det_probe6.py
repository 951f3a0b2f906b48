"""Determinism probe 6: does it need two workgroups on a CU?  (n, max_grid) sweeps, distinct gradient vectors of R runs."""
import os
import numpy as np
from paddlescience_amd import device, hotpath as hp
from tests.test_fused_step import _run, _weights

d = device.get_device()
lay = hp.NetLayout(2, 4, 64, 1, "tanh")
flat = _weights(lay, 3)
R = int(os.environ.get("PROBE_RUNS", "16"))
for static in (1, 0):
    for n, mg in ((4096, 0), (8192, 256), (16384, 256), (8192, 0), (8192, 384), (6144, 0), (4096 + 16, 0), (4096 + 1024, 0)):
        keys = [_run(d, lay, [("allen_cahn", n)], flat, True, 1, max_grid=mg, tail=1, static_program=static)[1][0].tobytes() for _ in range(R)]
        uniq = sorted(set(keys), key=keys.index)
        print(f"static={static} n={n} tiles={n // 16} max_grid={mg}: {len(uniq)} distinct of {R}", flush=True)
