"""Generates tests/golden/spinn.npz by executing the REFERENCE's own SPINN / Helmholtz code
(/root/reference/ppsci/arch/spinn.py, arch/mlp.py ModifiedMLP, equation/pde/helmholtz.py incl. hvp_revrev's
nested jvp) under the torch-backed paddle shim, in float64.

    python tests/golden/make_spinn_golden.py

`paddle.incubate.autograd.jvp(f, xs)` (unit tangents by default) is provided through
torch.autograd.functional.jvp(create_graph=True), so that the residual stays differentiable with respect to the
parameters for the gradient of the loss.  Per case: branch-net parameters (names of the reference state dict
per branch), coordinates, label grid, u, the Helmholtz residual, the MSE-mean loss and d loss / d parameters."""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import _paddle_shim as S  # noqa: E402

D = torch.float64


def install():
    mods = S.import_hotpath()
    paddle = sys.modules["paddle"]
    nn = sys.modules["paddle.nn"]

    class Sequential(S.Layer):
        def __init__(self, *layers):
            super().__init__()
            self._seq = list(layers)
            for i, l in enumerate(self._seq):
                self._subs[str(i)] = l

        def forward(self, x):
            for l in self._seq:
                x = l(x)
            return x

        def __getitem__(self, i):
            return self._seq[i]

    nn.Sequential = Sequential

    def sublayers(self, include_self=False):
        out = [self] if include_self else []
        for s in self._subs.values():
            out += sublayers(s, True)
        return out

    S.Layer.sublayers = sublayers

    def jvp(func, xs, v=None):
        xs_t = tuple(xs) if isinstance(xs, (list, tuple)) else (xs,)
        v_t = tuple(torch.ones_like(x) for x in xs_t) if v is None else (tuple(v) if isinstance(v, (list, tuple)) else (v,))
        is_seq = [False]

        def f(*a):
            o = func(*a)
            if isinstance(o, (list, tuple)):
                is_seq[0] = True
                return tuple(o)
            return o

        out, tang = torch.autograd.functional.jvp(f, xs_t, v_t, create_graph=True)
        if is_seq[0]:
            return list(out), list(tang)
        return out, tang

    incubate = importlib.import_module("paddle.incubate")
    autograd = importlib.import_module("paddle.incubate.autograd")
    autograd.jvp = jvp
    incubate.autograd = autograd
    paddle.incubate = incubate
    # paddle's unsqueeze accepts a list of axes (applied one after the other)
    _unsq = torch.Tensor.unsqueeze

    def unsqueeze(self, axis):
        if isinstance(axis, (list, tuple)):
            t = self
            for a in axis:
                t = _unsq(t, a)
            return t
        return _unsq(self, axis)

    torch.Tensor.unsqueeze = unsqueeze
    spinn = importlib.import_module("ppsci.arch.spinn")
    helm = importlib.import_module("ppsci.equation.pde.helmholtz")
    return spinn, helm


CASES = {
    # name: (r, num_layers, hidden, activation, (nx, ny, nz), k)
    "spinn_tanh_7x5x6": (4, 3, 16, "tanh", (7, 5, 6), 1.0),
    "spinn_tanh_4x9x3_k2": (3, 2, 8, "tanh", (4, 9, 3), 2.0),
}


def main():
    spinn, helm = install()
    out = {}
    for cname, (r, nl, hid, act, shape, k) in CASES.items():
        rng = np.random.default_rng(len(cname) * 131)
        model = spinn.SPINN(("x", "y", "z"), ("u",), r, nl, hid, act)
        names = []
        with torch.no_grad():
            for b, net in enumerate(model.branch_nets):
                for n, p in net.named_parameters():
                    fan_in = p.shape[0] if p.ndim == 2 else 1
                    scale = (1.0 / np.sqrt(fan_in)) if p.ndim == 2 else 0.1
                    v = (rng.standard_normal(tuple(p.shape)) * scale).astype(np.float32).astype(np.float64)
                    p.copy_(torch.tensor(v))
                    names.append((b, n, p))
                    out[f"{cname}/param/{b}/{n}"] = v
        xs = [torch.tensor(rng.uniform(-1, 1, (n, 1)).astype(np.float32).astype(np.float64)) for n in shape]
        label = torch.tensor(rng.standard_normal(shape + (1,)).astype(np.float32).astype(np.float64))
        eq = helm.Helmholtz(3, k)
        eq.model = model
        data = {"x": xs[0], "y": xs[1], "z": xs[2]}
        data.update(model(data))
        res = eq.equations["helmholtz"](data)
        loss = ((res - label) ** 2).mean()
        grads = torch.autograd.grad(loss, [p for _, _, p in names], allow_unused=True)
        for (b, n, p), g in zip(names, grads):
            out[f"{cname}/grad/{b}/{n}"] = (torch.zeros_like(p) if g is None else g).numpy()
        for i, key in enumerate("xyz"):
            out[f"{cname}/{key}"] = xs[i].numpy()
        out[f"{cname}/label"] = label.numpy()
        out[f"{cname}/u"] = data["u"].detach().numpy()
        out[f"{cname}/residual"] = res.detach().numpy()
        out[f"{cname}/loss"] = np.asarray(float(loss.detach()))
        out[f"{cname}/config"] = np.asarray([r, nl, hid, k])
        print(cname, "u", tuple(data["u"].shape), "loss", float(loss.detach()))
    np.savez_compressed(os.path.join(HERE, "spinn.npz"), **out)
    print("wrote spinn.npz", len(out), "arrays")


if __name__ == "__main__":
    main()
