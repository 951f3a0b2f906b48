"""The reference's SFNO configuration (examples/neuraloperator/conf/sfno_swe_pretrain.yaml: in 3, out 3, hidden 32, projection 64, 4
layers, 32 degrees x 16 orders, GroupNorm, batch 4 on the 32 x 64 grid) stepped through the OperatorEngine: used under rocprofv3 to
list the kernels of a step.    python tools/sfno_step.py [steps] [nlat] [nlon]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ppsci  # noqa: E402
from paddlescience_amd.engine import step_with_adam  # noqa: E402
from paddlescience_amd.operator_engine import OperatorConstraint, OperatorEngine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
H = int(sys.argv[2]) if len(sys.argv) > 2 else 32
W = int(sys.argv[3]) if len(sys.argv) > 3 else 64
B = 4
torch.manual_seed(0)
model = ppsci.arch.SFNONet(("x",), ("y",), (32, 32), 32, in_channels=3, out_channels=3, lifting_channels=256, projection_channels=64,
                           n_layers=4, norm="group_norm")
x = torch.as_tensor(np.random.default_rng(42).standard_normal((B, 3, H, W)).astype(np.float32)).cuda()
y = torch.as_tensor(np.random.default_rng(43).standard_normal((B, 3, H, W)).astype(np.float32)).cuda()
opt = ppsci.optimizer.Adam(1e-3)(model)
cst = OperatorConstraint("Sup", model, {"y": lambda d: d["y"]}, ppsci.loss.MSELoss("mean"), x.device, ["y"], B)
cst.bind({"x": x}, {"y": y})
eng = OperatorEngine(model)


def step():
    step_with_adam(eng, [cst], opt, model.flat_params)


for i in range(n):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(n):
    step()
e1.record()
torch.cuda.synchronize()
print("native:", type(eng.native).__name__, "sht", eng.native.sht, "params", model.flat_params.numel(), "loss", cst.losses(),
      "ms_per_step_hip_events", e0.elapsed_time(e1) / n)
