"""Generates tests/golden/modified_mlp.npz by executing the REFERENCE's own ModifiedMLP (ppsci/arch/mlp.py:318-528, with
PeriodEmbedding / FourierEmbedding / RandomWeightFactorization), autodiff/ad.py, utils/symbolic.py and loss/mse.py in float64
under the torch-backed paddle shim: outputs, per-point residuals, loss terms and the gradient w.r.t. every named parameter.

    python tests/golden/make_modified_mlp_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from make_piratenet_golden import draw_params, equations  # noqa: E402

CASES = {
    "plain_tanh_ns": dict(inputs=("x", "y"), outputs=("u", "v"), layers=3, hidden=20, act="tanh", periods=None, fourier=None,
                          rwf=None, eq="two_out", n=35, lo=[-1, -1], hi=[1, 1], reduction="sum"),
    "fourier_rwf_allen_cahn": dict(inputs=("t", "x"), outputs=("u",), layers=2, hidden=24, act="tanh", periods={"x": (2.0, False)},
                                   fourier={"dim": 16, "scale": 1.5}, rwf={"mean": 0.5, "std": 0.1}, eq="allen_cahn", n=40,
                                   lo=[0, -1], hi=[1, 1], reduction="mean"),
    "periods_silu_heat": dict(inputs=("x", "y", "t"), outputs=("u",), layers=4, hidden=16, act="silu",
                              periods={"y": (3.0, False)}, fourier=None, rwf=None, eq="heat", n=21, lo=[-1, -1, 0],
                              hi=[1, 1, 1], reduction="mean"),
    "one_layer_sin_value": dict(inputs=("x",), outputs=("u",), layers=1, hidden=16, act="sin", periods=None, fourier=None,
                                rwf={"mean": 1.0, "std": 0.1}, eq="value", n=19, lo=[-2], hi=[2], reduction="mean"),
}


def main():
    import _paddle_shim as S

    mods = S.import_hotpath()
    ModifiedMLP = mods["mlp"].ModifiedMLP
    lambdify = mods["symbolic"].lambdify
    MSELoss = mods["mse"].MSELoss
    clear = mods["ad"].clear
    out = {}
    for ci, (name, c) in enumerate(CASES.items()):
        model = ModifiedMLP(c["inputs"], c["outputs"], c["layers"], c["hidden"], c["act"], periods=c["periods"],
                            fourier=c["fourier"], random_weight=c["rwf"])
        named = [(n, p) for n, p in model.named_parameters()]
        cc = dict(c, fourier=c["fourier"] or {"scale": 1.0}, rwf=c["rwf"] or {"mean": 1.0, "std": 0.1})
        vals = draw_params(named, cc, 700 + ci)
        train = []
        with torch.no_grad():
            for n, p in named:
                if n in vals:
                    p.copy_(torch.tensor(vals[n]))
                    train.append((n, p))
        rng = np.random.default_rng(7000 + ci)
        X = rng.uniform(c["lo"], c["hi"], (c["n"], len(c["inputs"]))).astype(np.float32).astype(np.float64)
        data = {k: torch.tensor(X[:, j:j + 1], requires_grad=True) for j, k in enumerate(c["inputs"])}
        eqs = equations(c)
        output_dict = model(data)
        data_dict = dict(data)
        data_dict.update(output_dict)
        for k, ex in eqs.items():
            output_dict[k] = lambdify(ex, model, fuse_derivative=True)(data_dict)
        clear()
        keys = list(eqs.keys())
        label = {k: torch.tensor(rng.standard_normal((c["n"], 1)).astype(np.float32).astype(np.float64) * 0.05) for k in keys}
        losses = MSELoss(c["reduction"])(output_dict, label, None)
        total = 0.0
        for i, k in enumerate(losses):
            total = losses[k] if i == 0 else total + losses[k]
        grads = torch.autograd.grad(total, [p for _, p in train], allow_unused=True)
        out[f"{name}/X"] = X
        out[f"{name}/total"] = np.asarray(float(total.detach()))
        for (n, p), g in zip(train, grads):
            out[f"{name}/param/{n}"] = vals[n]
            out[f"{name}/grad/{n}"] = (torch.zeros_like(p) if g is None else g).detach().numpy()
        for k in c["outputs"]:
            out[f"{name}/out/{k}"] = output_dict[k].detach().numpy()[:, 0]
        for k in keys:
            out[f"{name}/res/{k}"] = output_dict[k].detach().numpy()[:, 0]
            out[f"{name}/loss/{k}"] = np.asarray(float(losses[k].detach()))
            out[f"{name}/label/{k}"] = label[k].numpy()[:, 0]
        print(name, [n for n, _ in train], "total loss", float(total.detach()), flush=True)
    np.savez_compressed(os.path.join(HERE, "modified_mlp.npz"), **out)


if __name__ == "__main__":
    main()
