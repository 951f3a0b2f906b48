"""FNO lifting / projection backward: the round-5 kernels (ppsci_fno_lift0_wgrad with both lifting gradients, ppsci_fno_proj_hidden_grad)
against the launches they replace (PPSCI_FNO_LIFT0_FUSED / _LIFT1_FUSED / _PROJ_STREAMED = 0), every parameter gradient of a small TFNO.
    python tools/lift0_check.py            (CPU emulator)        DEV=gpu python tools/lift0_check.py   (MI355X, 16 x 64 x 64)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import paddlescience_amd as ppsci
from paddlescience_amd import device
dev = os.environ.get("DEV", "emu")
if dev == "emu":
    from tests.emu import build_emu
    build_emu.inject(); device.set_device("cpu")
def run(flag):
    os.environ["PPSCI_FNO_LIFT0_FUSED"] = flag
    os.environ["PPSCI_FNO_PROJ_STREAMED"] = flag
    os.environ["PPSCI_FNO_LIFT1_FUSED"] = flag
    torch.manual_seed(0)
    model = ppsci.arch.TFNO2dNet(("x",), ("y",), 4, 4, hidden_channels=8, in_channels=3, out_channels=1, lifting_channels=24,
                                 projection_channels=16, n_layers=2, norm="group_norm")
    B, H, W = (2, 12, 20) if dev == "emu" else (16, 64, 64)
    d = model.flat_params.device
    x = torch.as_tensor(np.random.default_rng(1).standard_normal((B, 3, H, W)).astype(np.float32)).to(d)
    gy = torch.as_tensor(np.random.default_rng(2).standard_normal((B, 1, H, W)).astype(np.float32)).to(d)
    nat = model.native()
    y = nat.forward(x.contiguous())
    model.flat_grad.zero_()
    nat.backward(gy)
    g = {n: p.grad.detach().cpu().numpy().copy() for n, p in torch.nn.Module.named_parameters(model)}
    return g
a, b = run("1"), run("0")
for n in a:
    r = np.linalg.norm(a[n] - b[n]) / max(np.linalg.norm(b[n]), 1e-30)
    if "lifting" in n or "projection" in n or r > 1e-6: print(n, a[n].shape, "rel diff new vs old path:", r)
print("max rel", max(np.linalg.norm(a[n] - b[n]) / max(np.linalg.norm(b[n]), 1e-30) for n in a))
