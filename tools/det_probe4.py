"""Determinism probe 4: N runs of one fused step; how many distinct gradient vectors come out."""
import os
import sys
import numpy as np
from paddlescience_amd import device, hotpath as hp
from tests.test_fused_step import _run, _weights

d = device.get_device()
lay = hp.NetLayout(2, 4, 64, 1, "tanh")
flat = _weights(lay, 3)
R = int(os.environ.get("PROBE_RUNS", "12"))
for static in (1, 0):
    for n in (8192, 20000, 6144):
        for tail in (1, 3):
            gs = [_run(d, lay, [("allen_cahn", n)], flat, True, 1, tail=tail, static_program=static)[1][0] for _ in range(R)]
            keys = [g.tobytes() for g in gs]
            uniq = sorted(set(keys), key=keys.index)
            print(f"lib={os.path.basename(os.environ.get('PPSCI_HIP_LIB', 'default'))} static={static} n={n} tail={tail}: {len(uniq)} distinct of {R}: "
                  + " ".join(str(uniq.index(k)) for k in keys), flush=True)
