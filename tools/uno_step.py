"""The reference's UNO configuration (examples/neuraloperator/conf/uno_darcyflow_pretrain.yaml: batch 16 at 16 x 16, domain padding
0.2 -> 19 x 19, five Fourier layers 32-64-64-64-32 on the grids 19 / 10 / 10 / 20 / 19, U skips) stepped through the OperatorEngine:
used under rocprofv3 to list the kernels of a step.    python tools/uno_step.py [steps] [resolution]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ppsci  # noqa: E402
from paddlescience_amd.engine import step_with_adam  # noqa: E402
from paddlescience_amd.operator_engine import OperatorConstraint, OperatorEngine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
R = int(sys.argv[2]) if len(sys.argv) > 2 else 16
B = 16
torch.manual_seed(0)
model = ppsci.arch.UNONet(("x",), ("y",), 3, 1, 64, 256, 64, n_layers=5, uno_out_channels=[32, 64, 64, 64, 32],
                          uno_n_modes=[[16, 16], [8, 8], [8, 8], [8, 8], [16, 16]],
                          uno_scalings=[[1.0, 1.0], [0.5, 0.5], [1, 1], [2, 2], [1, 1]], norm="group_norm", domain_padding=0.2,
                          domain_padding_mode="one-sided", fft_norm="forward")
x = torch.as_tensor(np.random.default_rng(42).standard_normal((B, 3, R, R)).astype(np.float32)).cuda()
y = torch.as_tensor(np.random.default_rng(43).standard_normal((B, 1, R, R)).astype(np.float32)).cuda()
opt = ppsci.optimizer.Adam(1e-3)(model)
cst = OperatorConstraint("Sup", model, {"y": lambda d: d["y"]}, ppsci.loss.H1Loss_train(d=2) if os.environ.get("UNO_H1") else ppsci.loss.MSELoss("mean"),
                         x.device, ["y"], B)
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
print("native:", type(eng.native).__name__, "params", model.flat_params.numel(), "loss", cst.losses(), "ms_per_step_hip_events",
      e0.elapsed_time(e1) / n)
