"""FNO parity against fixtures produced by the REFERENCE's own FNO code (tests/golden/make_fno_golden.py runs
/root/reference/ppsci/arch/{fno_block,tfnonet}.py under the torch-backed paddle shim, fp64):
  * the oracle restatement (oracle/ref_torch.fno_forward) reproduces them to fp64 round-off  -> oracle pinned;
  * TFNO2dNet on the HIP spectral kernel reproduces them within fp32 tolerance (forward rel-L2 <= 2e-5,
    parameter gradients rel-L2 <= 2e-4)."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_torch as R
from tests.common import make_dev_fixture, rel

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "fno.npz"))
CASES = sorted({k.split("/")[0] for k in G.files})
dev = make_dev_fixture()


def _case(c):
    mx, my, hid, lift, proj, nl, gn = [int(v) for v in G[f"{c}/config"]]
    P = {k[len(c) + 7:]: G[k] for k in G.files if k.startswith(f"{c}/param/")}
    Gr = {k[len(c) + 6:]: G[k] for k in G.files if k.startswith(f"{c}/grad/")}
    return (mx, my, hid, lift, proj, nl, "group_norm" if gn else None), P, Gr


def _padding(c):
    """DomainPadding of the case (fno_block.py:19-140): (fractions or None, mode)."""
    fh, fw, sym = [float(v) for v in G[f"{c}/domain_padding"]]
    return ([fh, fw] if fh or fw else None), ("symmetric" if sym else "one-sided")


@pytest.mark.parametrize("c", CASES)
def test_oracle_reproduces_reference_fno(c):
    (mx, my, hid, lift, proj, nl, norm), P, Gr = _case(c)
    Pt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in P.items()}
    pad, pad_mode = _padding(c)
    y = R.fno_forward(torch.tensor(G[f"{c}/x"]), Pt, nl, (mx, my), norm, domain_padding=pad, domain_padding_mode=pad_mode)
    np.testing.assert_allclose(y.detach().numpy(), G[f"{c}/y"], rtol=0, atol=1e-11)
    loss = ((y - torch.tensor(G[f"{c}/target"])) ** 2).mean()
    assert abs(float(loss.detach()) - float(G[f"{c}/loss"])) < 1e-12
    names = sorted(Gr)
    for n, g in zip(names, torch.autograd.grad(loss, [Pt[n] for n in names])):
        np.testing.assert_allclose(g.numpy(), Gr[n], rtol=0, atol=1e-12, err_msg=n)


@pytest.mark.parametrize("c", CASES)
def test_hip_path_reproduces_reference_fno(c, dev):
    import ppsci

    (mx, my, hid, lift, proj, nl, norm), P, Gr = _case(c)
    pad, pad_mode = _padding(c)
    model = ppsci.arch.TFNO2dNet(("x",), ("y",), mx, my, hid, 3, 1, lift, proj, nl, norm=norm, domain_padding=pad,
                                 domain_padding_mode=pad_mode)
    model.set_state_dict({k: v.astype(np.float32) for k, v in P.items()})
    d = model.flat_params.device
    y = model({"x": G[f"{c}/x"].astype(np.float32)})["y"]  # FNONet.forward == the kernels' executor
    assert rel(y.detach().cpu().numpy(), G[f"{c}/y"]) < 1e-5
    nat = model.native()
    yn = nat.forward(torch.as_tensor(G[f"{c}/x"].astype(np.float32)).to(d))
    losses, gy = ppsci.loss.MSELoss("mean").value_and_grad(yn, torch.as_tensor(G[f"{c}/target"].astype(np.float32)).to(d), "y")
    loss = losses["y"]
    model.flat_grad.fill_(float("nan"))
    nat.backward(gy)  # the hand-written backward writes every parameter gradient
    assert abs(float(loss.detach()) - float(G[f"{c}/loss"])) < 1e-5 * float(G[f"{c}/loss"])
    for n, p in torch.nn.Module.named_parameters(model):
        assert rel(p.grad.cpu().numpy(), Gr[n]) < 1e-4, n
