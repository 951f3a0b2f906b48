"""Generates tests/golden/neuralop.npz by executing the REFERENCE's own operator-learning data path under the
torch-backed paddle shim: ppsci/data/dataset/darcyflow_dataset.py (UnitGaussianNormalizer, PositionalEmbedding2D,
DarcyFlowDataset on small synthetic `darcy_*.npy` files written here) and examples/neuraloperator/metric.py
(LpLoss / LpLoss_train / H1Loss / H1Loss_train).

    python tests/golden/make_neuralop_golden.py
"""
import importlib.util
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
REF = "/root/reference"


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def synthetic(n, res, seed):
    rng = np.random.default_rng(seed)
    x = np.where(rng.standard_normal((n, res, res)) > 0, 12.0, 3.0).astype(np.float32)
    y = rng.standard_normal((n, res, res)).astype(np.float32) * 0.02 + 0.01 * x
    return x, y


def main():
    import _paddle_shim as S

    paddle = S.install()
    S.DTYPE = torch.float32
    # the few extra paddle calls of the data path / metrics (paddle -> torch)
    paddle.to_tensor = lambda data, dtype=None, place=None, stop_gradient=True: torch.as_tensor(np.asarray(data))
    paddle.mean = lambda x, axis=None, keepdim=False: torch.mean(x, dim=tuple(axis) if isinstance(axis, (list, tuple)) else axis, keepdim=keepdim)
    paddle.std = lambda x, axis=None, keepdim=False: torch.std(x, dim=tuple(axis) if isinstance(axis, (list, tuple)) else axis, keepdim=keepdim)
    paddle.sum = lambda x, axis=None, keepdim=False: torch.sum(x, dim=axis, keepdim=keepdim)
    paddle.linspace = lambda a, b, n: torch.linspace(a, b, n, dtype=torch.float32)
    paddle.meshgrid = lambda *a, indexing="ij": torch.meshgrid(*a, indexing=indexing)
    paddle.roll = lambda x, shifts, axis: torch.roll(x, shifts, dims=axis)
    paddle.flatten = lambda x, start_axis=0, stop_axis=-1: torch.flatten(x, start_axis, stop_axis)
    paddle.norm = lambda x, p=2, axis=None, keepdim=False: torch.linalg.vector_norm(x, ord=p, dim=axis, keepdim=keepdim)
    torch.Tensor.astype = lambda self, dt: self.to({"float32": torch.float32, "float64": torch.float64}.get(dt, dt) if isinstance(dt, str) else dt)
    _expand = torch.Tensor.expand
    torch.Tensor.expand = lambda self, *s: _expand(self, *(s[0] if len(s) == 1 and isinstance(s[0], (list, tuple)) else s))
    _tile = torch.Tensor.tile
    torch.Tensor.tile = lambda self, *s: _tile(self, *(s[0] if len(s) == 1 and isinstance(s[0], (list, tuple)) else s))
    io = types.ModuleType("paddle.io")
    io.Dataset = type("Dataset", (), {"__init__": lambda self, *a, **k: None})
    sys.modules["paddle.io"] = io
    paddle.io = io

    ds_mod = load(os.path.join(REF, "ppsci/data/dataset/darcyflow_dataset.py"), "ref_darcy")
    metric = load(os.path.join(REF, "examples/neuraloperator/metric.py"), "ref_metric")
    out = {}
    tmp = tempfile.mkdtemp()
    raw = {"train_16": synthetic(6, 16, 1), "test_16": synthetic(4, 16, 2), "test_32": synthetic(3, 32, 3)}
    for k, (x, y) in raw.items():
        np.save(os.path.join(tmp, f"darcy_{k}.npy"), {"x": x, "y": y}, allow_pickle=True)
        out[f"raw/{k}/x"], out[f"raw/{k}/y"] = x, y
    for tag, kw in {"default": {}, "enc_in": dict(encode_input=True),
                    "nopos": dict(positional_encoding=False, encode_output=False)}.items():
        for split in ("train", "test_16x16", "test_32x32"):
            if tag == "nopos":
                # the reference only defines `transform_x` when positional_encoding is on (AttributeError otherwise):
                # stored for the two encoders' statistics only
                continue
            ds = ds_mod.DarcyFlowDataset(("x",), ("y",), tmp, test_resolutions=[16, 32], train_resolution=16,
                                         data_split=split, **kw)
            out[f"{tag}/{split}/len"] = np.asarray(len(ds))
            for i in (0, len(ds) - 1):
                inp, lab, w = ds[i]
                out[f"{tag}/{split}/{i}/x"] = inp["x"].numpy()
                out[f"{tag}/{split}/{i}/y"] = lab["y"].numpy()
        ds = ds_mod.DarcyFlowDataset(("x",), ("y",), tmp, test_resolutions=[16, 32], train_resolution=16, **kw) if tag != "nopos" else None
        if ds is not None and ds.output_encoder is not None:
            out[f"{tag}/out_mean"], out[f"{tag}/out_std"] = ds.output_encoder.mean.numpy(), ds.output_encoder.std.numpy()
            z = torch.as_tensor(raw["test_16"][1][:2, None])
            out[f"{tag}/decode"] = ds.output_encoder.decode(ds.output_encoder.encode(z.clone())).numpy()
        if ds is not None and ds.input_encoder is not None:
            out[f"{tag}/in_mean"], out[f"{tag}/in_std"] = ds.input_encoder.mean.numpy(), ds.input_encoder.std.numpy()
    pe = ds_mod.PositionalEmbedding2D([[0, 1], [-1, 1]])
    out["posenc/3x5x4"] = pe(torch.arange(60, dtype=torch.float32).reshape(3, 5, 4)).numpy()
    # metrics: fp64 values of the reference formulas
    rng = np.random.default_rng(7)
    x = torch.tensor(rng.standard_normal((5, 1, 12, 10)))
    y = torch.tensor(rng.standard_normal((5, 1, 12, 10)))
    out["metric/x"], out["metric/y"] = x.numpy(), y.numpy()
    L = 1.0
    cases = {
        "lp_d2": metric.LpLoss(d=2, p=2), "lp_d2_p1_mean": metric.LpLoss(d=2, p=1, reductions="mean"),
        "lp_train_d2": metric.LpLoss_train(d=2, p=2),
        "h1_d2": metric.H1Loss(d=2), "h1_d2_fix": metric.H1Loss(d=2, L=L, fix_x_bnd=True, fix_y_bnd=True),
        "h1_train_d2": metric.H1Loss_train(d=2),
    }
    for name, fn in cases.items():
        res = fn({"y": x.clone()}, {"y": y.clone()})
        for k, v in res.items():
            out[f"metric/{name}/{k}"] = np.asarray(v.numpy(), dtype=np.float64)
        if hasattr(fn, "abs") and name in ("lp_d2", "h1_d2_fix"):
            out[f"metric/{name}/abs"] = np.asarray(fn.abs(x.clone(), y.clone()).numpy(), dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "neuralop.npz"), **out)
    print({k: (v.shape if v.ndim else float(v)) for k, v in out.items() if k.startswith("metric/") and "/x" not in k and "/y" != k[-2:]})


if __name__ == "__main__":
    main()
