"""Where the time of the one-launch step kernel (cfg 1: Laplace2D 3 x 20, 10 k points) goes: run on a -DPPSCI_STEP_TIMERS build
(python tools/build_variant.py steptimers taylor_step_tanh.hip -DPPSCI_STEP_TIMERS;
 PPSCI_HIP_LIB=paddlescience_amd/libppsci_hip.steptimers.so python tools/step_phases.py [points])."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from paddlescience_amd import _lib as L  # noqa: E402
from paddlescience_amd import hotpath as hp  # noqa: E402
from paddlescience_amd.engine import Engine  # noqa: E402
from tests.test_one_launch import _constraint, _weights  # noqa: E402

PHASES = ["fwd body", "sync (U, stash stores)", "residual program", "sync (dL/dU stores)", "reverse body", "reduction tree + Adam"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
dev = torch.device("cuda", 0)
lay = hp.NetLayout(2, 3, 20, 1, "tanh")
eng = Engine(lay, torch.tensor(_weights(lay, 1), device=dev))
cst = _constraint(dev, "laplace", lay, n, 1)
for _ in range(5):
    eng.train_step([cst], 1e-3)
torch.cuda.synchronize()
t_step = bench.time_events(lambda: eng.train_step([cst], 1e-3), 50, median=True)
nwg = ((n + 15) // 16 + 3) // 4  # one tile per wave, four waves per workgroup
buf = torch.zeros(nwg * 4 * 8, dtype=torch.int64, device=dev)
lib = L.lib()
lib.ppsci_step_timers_set.argtypes = [C.c_void_p]
lib.ppsci_step_timers_set.restype = C.c_int
assert lib.ppsci_step_timers_set(buf.data_ptr()) == 0
eng.train_step([cst], 1e-3)
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(nwg, 4, 8).astype(np.float64)
live = (t[:, :, 0] > 0) & (t[:, :, 6] > 0)
t0 = t[:, :, 0][live].min()
d = np.diff(t[:, :, :7], axis=2)  # [wg][wave][6]
out = {"points": n, "step_us (hip events, median of 50)": t_step * 1e6, "workgroups": int(live[:, 0].sum()),
       "clock_note": "s_memtime ticks (the shader clock, ~2.4 GHz under load); the whole kernel in ticks is `span`",
       "phases_ticks_median_p90_max": {}}
for k, name in enumerate(PHASES):
    v = d[:, :, k][live]
    out["phases_ticks_median_p90_max"][name] = [float(np.median(v)), float(np.percentile(v, 90)), float(v.max())]
out["start_skew_ticks (last wave start - first)"] = float(t[:, :, 0][live].max() - t0)
out["span_ticks (first start .. last end)"] = float(t[:, :, 6][live].max() - t0)
out["end_of_reverse_ticks (median, max) since first start"] = [float(np.median(t[:, :, 5][live] - t0)), float((t[:, :, 5][live] - t0).max())]
print(json.dumps(out, indent=1))
