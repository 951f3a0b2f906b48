"""Times the fused Allen-Cahn step (primary config: 4 x 64 tanh, 100 000 points) of the library PPSCI_HIP_LIB selects:
main kernel alone and the whole step, HIP events, one JSON line.  Used for A/B and ablation builds (tools/fused_ablate.sh)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from paddlescience_amd import _lib as L  # noqa: E402
from paddlescience_amd import hotpath as hp  # noqa: E402
from paddlescience_amd.engine import Engine, FusedConstraint  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
tail = int(os.environ.get("PPSCI_STEP_TAIL", "-1"))
dev = torch.device("cuda", 0)
flat = bench.bench_weights(2, [64] * 4, 1)
X = np.random.default_rng(42).uniform([0, -1], [1, 1], (n, 2)).astype(np.float32)
L.lib().ppsci_set_step_tail(tail)
L.lib().ppsci_set_static_program(0 if os.environ.get("PPSCI_STATIC_PROGRAM", "1") == "0" else 1)
lay = hp.NetLayout(2, 4, 64, 1, "tanh")
xs = [torch.tensor(X[:, j].copy(), device=dev) for j in range(2)]
cst = FusedConstraint("EQ", lay, hp.StreamSpec([[0.0, 1.0], [1.0, 0.0]], 1), bench.allen_cahn_program(n), xs, [], ["allen_cahn"])
eng = Engine(lay, torch.tensor(flat, device=dev))
for _ in range(5):
    eng.train_step([cst], 1e-3)
torch.cuda.synchronize()
reps = int(os.environ.get("REPS", "40"))
out = {"lib": os.path.basename(os.environ.get("PPSCI_HIP_LIB", "libppsci_hip.so")), "tail": tail, "points": n,
       "static_program": cst._step_plan.static_program,
       "step_us": round(bench.time_events(lambda: eng.train_step([cst], 1e-3), reps) * 1e6, 2),
       "main_us": round(bench.time_events(cst._step_plan.run_main, reps) * 1e6, 2)}
out["step_us_2"] = round(bench.time_events(lambda: eng.train_step([cst], 1e-3), reps) * 1e6, 2)
out["loss"] = float(cst.loss_terms[0])
print(json.dumps(out), flush=True)
