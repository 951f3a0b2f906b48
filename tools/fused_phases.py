"""Where a tile's time goes in the fused tile kernel: run on a -DPPSCI_FUSED_TIMERS build
(python tools/build_variant.py timers taylor_fused_tanh.hip -DPPSCI_FUSED_TIMERS;
 PPSCI_HIP_LIB=paddlescience_amd/libppsci_hip.timers.so python tools/fused_phases.py [points])."""
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

PHASES = ["0 tile top", "1 fwd layer 0 (VALU)", "2 fwd split+publish+barrier", "3 fwd GEMM", "4 fwd act+stash", "5 last linear+barrier",
          "6 program (wave 0) + barrier", "7 bwd pointwise+split+publish+barrier", "8 bwd hbar GEMM", "9 bwd h split+publish+barrier",
          "10 bwd Wbar GEMM", "11 bwd layer 0", "12 end-of-tile barrier", "13 (of 6) U store", "14 (of 6) program"]

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
dev = torch.device("cuda", 0)
flat = bench.bench_weights(2, [64] * 4, 1)
X = np.random.default_rng(42).uniform([0, -1], [1, 1], (n, 2)).astype(np.float32)
L.lib().ppsci_set_step_tail(1)
lay = hp.NetLayout(2, 4, 64, 1, "tanh")
xs = [torch.tensor(X[:, j].copy(), device=dev) for j in range(2)]
cst = FusedConstraint("EQ", lay, hp.StreamSpec([[0.0, 1.0], [1.0, 0.0]], 1), bench.allen_cahn_program(n), xs, [], ["allen_cahn"],
                      want_residual=False)
eng = Engine(lay, torch.tensor(flat, device=dev))
for _ in range(3):
    eng.forward_backward([cst])
torch.cuda.synchronize()
t_main = bench.time_events(cst._step_plan.run_main, 20)
eng.forward_backward([cst])
torch.cuda.synchronize()
grid = min((n + 15) // 16, 512)
tiles = (n + 15) // 16
per_tile_f, psmall = 3 * 64 * 64, 2 * 64 + 4 * 64 + 64 + 1
pad4 = lambda v: (v + 3) & ~3  # noqa: E731
off = pad4(grid * per_tile_f) + pad4(grid * psmall) + pad4(grid)  # taylor_api.hip step_layout: the tree rows
t = cst._step_ws[off: off + grid * 4 * 16].cpu().numpy().reshape(grid, 4, 16).astype(np.float64)
per_tile = t / (tiles / grid)
out = {"points": n, "main_us": t_main * 1e6, "cycles_per_tile_wave0": {}, "cycles_per_tile_wave3": {}}
for k, name in enumerate(PHASES):
    out["cycles_per_tile_wave0"][name] = round(float(np.median(per_tile[:, 0, k])), 0)
    out["cycles_per_tile_wave3"][name] = round(float(np.median(per_tile[:, 3, k])), 0)
    out.setdefault("wave0_p10_p90", {})[name] = [round(float(np.percentile(per_tile[:, 0, k], q)), 0) for q in (10, 90)]
VM = ["0 entry", "1 init + label prefetch + tin zero", "2 loads / constants", "3 forward steps", "4 residual terms", "5 reverse steps", "6 dL/dU"]
out["program_inner_cycles_per_tile"] = {nm: round(float(np.median(per_tile[:, 1, k])), 0) for k, nm in enumerate(VM)}
out["sum_wave0"] = round(float(np.median(per_tile[:, 0, :13].sum(axis=1))), 0)
print(json.dumps(out, indent=1))
