"""SPINN / Helmholtz3D parity against fixtures produced by the REFERENCE's own code
(tests/golden/make_spinn_golden.py runs /root/reference/ppsci/arch/spinn.py, arch/mlp.py ModifiedMLP and
equation/pde/helmholtz.py -- nested jvp -- under the torch-backed paddle shim, fp64):
  * the oracle restatement (oracle/ref_torch.ModifiedMLP1 + spinn_helmholtz) reproduces them to round-off;
  * the HIP path through ppsci.arch.SPINN / ppsci.equation.Helmholtz / Solver reproduces them within fp32
    tolerance (u rel-L2 <= 5e-6, loss rel <= 3e-5, parameter gradient rel-L2 <= 1e-4)."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_torch as R
from tests.common import make_dev_fixture, rel

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "spinn.npz"))
CASES = sorted({k.split("/")[0] for k in G.files})
dev = make_dev_fixture()


def _branch_params(c, b):
    P = {k.split("/", 3)[3]: G[k] for k in G.files if k.startswith(f"{c}/param/{b}/")}
    L = sum(1 for k in P if k.startswith("linears.") and k.endswith(".weight"))
    return P, dict(wu=P["embed_u.0.weight"], bu=P["embed_u.0.bias"], wv=P["embed_v.0.weight"], bv=P["embed_v.0.bias"],
                   w=[P[f"linears.{l}.weight"] for l in range(L)], b=[P[f"linears.{l}.bias"] for l in range(L)],
                   wl=P["last_fc.weight"], bl=P["last_fc.bias"])


@pytest.mark.parametrize("c", CASES)
def test_oracle_reproduces_reference_spinn(c):
    r, nl, hid, k = G[f"{c}/config"]
    nets = [R.ModifiedMLP1(_branch_params(c, b)[1], "tanh") for b in range(3)]
    xs = [torch.tensor(G[f"{c}/{a}"], requires_grad=True) for a in "xyz"]
    u, res = R.spinn_helmholtz(nets, xs, float(k))
    np.testing.assert_allclose(u.detach().numpy(), G[f"{c}/u"][..., 0], rtol=0, atol=1e-11)
    np.testing.assert_allclose(res.detach().numpy(), G[f"{c}/residual"][..., 0], rtol=0, atol=1e-9)
    loss = ((res - torch.tensor(G[f"{c}/label"][..., 0])) ** 2).mean()
    assert abs(float(loss.detach()) - float(G[f"{c}/loss"])) < 1e-10


@pytest.mark.parametrize("c", CASES)
def test_hip_path_reproduces_reference_spinn(c, dev, tmp_path):
    import ppsci

    r, nl, hid, k = G[f"{c}/config"]
    model = ppsci.arch.SPINN(("x", "y", "z"), ("u",), int(r), int(nl), int(hid), "tanh")
    state, gref = {}, []
    for b in range(3):
        P, _ = _branch_params(c, b)
        for n, v in P.items():
            state[f"branch_nets.{b}.{n}"] = v.astype(np.float32)
    missing, unexpected = model.set_state_dict(state)
    assert not missing and not unexpected
    for name in model._names:  # flat gradient order = state-dict order
        b, n = name.split(".", 2)[1], name.split(".", 2)[2]
        gref.append(G[f"{c}/grad/{b}/{n}"].ravel())
    gref = np.concatenate(gref)
    eq = ppsci.equation.Helmholtz(3, float(k))
    eq.model = model
    xs = {a: G[f"{c}/{a}"].astype(np.float32) for a in "xyz"}
    label = G[f"{c}/label"].astype(np.float32)
    data = dict(xs, uc=label)
    pde = ppsci.constraint.SupervisedConstraint(
        {"dataset": {"name": "ContinuousNamedArrayDataset", "input": lambda: data, "label": lambda d: {"helmholtz": d["uc"]}}},
        output_expr=eq.equations, loss=ppsci.loss.MSELoss("mean"), name="PDE")
    opt = ppsci.optimizer.Adam(1e-3)(model)
    solver = ppsci.solver.Solver(model, {"PDE": pde}, str(tmp_path), opt, epochs=1, iters_per_epoch=1,
                                 equation={"Helmholtz": eq})
    cc = solver._compiled["PDE"]
    inp, lab, _ = next(pde.data_iter)
    cc.bind(inp, lab)
    solver.engine.forward_backward([cc])
    assert cc.loss() == pytest.approx(float(G[f"{c}/loss"]), rel=3e-5)
    assert rel(solver.engine.grad.cpu().numpy(), gref) < 1e-4
    pred = solver.predict(xs, batch_size=None, return_numpy=True)["u"]
    assert rel(pred, G[f"{c}/u"]) < 5e-6
