"""One TFNO-2D training configuration (BASELINE config 4 shape) stepped through the OperatorEngine: used under
rocprofv3 to list the kernels of a step (tools: rocprofv3 --kernel-trace --stats -- python tools/tfno_step.py)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ppsci  # noqa: E402
from paddlescience_amd.engine import step_with_adam  # noqa: E402
from paddlescience_amd.operator_engine import OperatorConstraint, OperatorEngine  # noqa: E402

B, H, W = 16, 64, 64
torch.manual_seed(0)
model = ppsci.arch.TFNO2dNet(("x",), ("y",), 12, 12, hidden_channels=32, in_channels=3, out_channels=1,
                             lifting_channels=256, projection_channels=64, n_layers=4, norm="group_norm")
x = torch.as_tensor(np.random.default_rng(42).standard_normal((B, 3, H, W)).astype(np.float32)).cuda()
y = torch.as_tensor(np.random.default_rng(43).standard_normal((B, 1, H, W)).astype(np.float32)).cuda()
opt = ppsci.optimizer.Adam(1e-3)(model)
cst = OperatorConstraint("Sup", model, {"y": lambda d: d["y"]}, ppsci.loss.MSELoss("mean"), x.device, ["y"], B)
cst.bind({"x": x}, {"y": y})
eng = OperatorEngine(model)


def step():  # what Solver.train runs: the sums over the weight-gradient partials + Adam in one launch (PPSCI_FUSED_REDUCE_ADAM=0: two)
    step_with_adam(eng, [cst], opt, model.flat_params)


n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
if os.environ.get("PPSCI_PW_NPX"):  # tuning: pixels per lane of the 1x1 convolutions (default: chosen per call)
    from paddlescience_amd import _lib as L

    L.lib().ppsci_set_pw_pixels_per_lane(int(os.environ["PPSCI_PW_NPX"]))
for i in range(n):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(n):
    step()
e1.record()
torch.cuda.synchronize()
print("native:", eng.native is not None, "loss", cst.losses(), "ms_per_step_hip_events", e0.elapsed_time(e1) / n)
