"""Checkpointing with the file set and key names of /root/reference/ppsci/utils/save_load.py:213-290:
`<output_dir>/checkpoints/<prefix>.{pdparams,pdopt,pdstates}` written by rank 0 only.

`.pdparams` is written the way `paddle.save(model.state_dict(), path)` writes it -- a protocol-4 pickle of
{structured name: numpy array} plus the "StructuredToParameterName@@" name table (Paddle behaviour, recalled
from paddle/framework/io.py, not visible in the reference tree) -- with the reference's parameter keys
(`linears.0.weight`, ..., `last_fc.bias`, mlp.py:264-277), so the reference can `paddle.load` it and files
saved by the reference (its published `*_pretrained.pdparams`) load here.  Reading uses a restricted unpickler
(numpy array reconstruction only).  `.pdopt` holds this framework's flat optimizer state (npz)."""
from __future__ import annotations

import json
import os
import pickle
from typing import Dict, Optional

import numpy as np
import torch

from . import logger


def _rank0() -> bool:
    try:
        import torch.distributed as dist

        return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0
    except Exception:  # noqa: BLE001
        return True


def _save_npz(path: str, arrays: Dict[str, np.ndarray]):
    with open(path, "wb") as f:
        np.savez(f, **arrays)


def _load_npz(path: str) -> Dict[str, np.ndarray]:
    with open(path, "rb") as f:
        z = np.load(f, allow_pickle=False)
        return {k: z[k] for k in z.files}


NAME_TABLE_KEY = "StructuredToParameterName@@"


def _save_pdparams(path: str, arrays: Dict[str, np.ndarray]):
    obj = {k: np.ascontiguousarray(v) for k, v in arrays.items()}
    obj[NAME_TABLE_KEY] = {k: k for k in arrays}
    with open(path, "wb") as f:
        pickle.dump(obj, f, protocol=4)


class _ArrayUnpickler(pickle.Unpickler):
    """Only what a pickled dict of numpy arrays needs; anything else in the stream is refused."""

    _ALLOWED = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
                ("numpy", "ndarray"), ("numpy", "dtype"), ("numpy.core.multiarray", "scalar"),
                ("numpy._core.multiarray", "scalar"), ("collections", "OrderedDict")}

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"refusing to load {module}.{name} from a checkpoint")


def _load_pdparams(path: str) -> Dict[str, np.ndarray]:
    with open(path, "rb") as f:
        head = f.read(2)
    if head == b"PK":  # archives written by earlier versions of this package
        return _load_npz(path)
    with open(path, "rb") as f:
        obj = _ArrayUnpickler(f).load()
    if not isinstance(obj, dict):
        raise ValueError(f"{path}: expected a state dict, got {type(obj).__name__}")
    obj.pop(NAME_TABLE_KEY, None)
    return {k: np.asarray(v) for k, v in obj.items()}


def _state_arrays(model) -> Dict[str, np.ndarray]:
    """state_dict as numpy arrays; models whose tensors are views of one flat buffer take ONE device->host copy."""
    sd = model.state_dict()
    flat = getattr(model, "flat_params", None)
    if isinstance(flat, torch.Tensor) and sd and all(
            v.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr() for v in sd.values()):
        host = flat.detach().cpu().numpy()
        base = flat.storage_offset()
        return {k: host[v.storage_offset() - base: v.storage_offset() - base + v.numel()].reshape(tuple(v.shape))
                for k, v in sd.items()}
    return {k: v.detach().cpu().numpy() for k, v in sd.items()}


def _load_equation(path: str, equation) -> None:
    """save_load.py:65-81 / :168-197: `<path>.pdeqn` = {equation name: state dict}."""
    if equation is None:
        return
    if not os.path.exists(f"{path}.pdeqn"):
        n = sum(len(eq.learnable_parameters) for eq in equation.values())
        if n > 0:
            logger.warning(f"There are a total of {n} learnable parameters in the equation, but {path}.pdeqn not found.")
        return
    with open(f"{path}.pdeqn", "rb") as f:
        eq_dict = _ArrayUnpickler(f).load()
    for name, eq in equation.items():
        eq.set_state_dict(eq_dict[name])
    logger.message(f"Finish loading equation parameters from: {path}.pdeqn")


def _set_model_state(model, path: str) -> None:
    """paddle's set_state_dict warns about missing / unexpected keys; a file none of whose keys matches a parameter
    (e.g. a plain-MLP file loaded into a weight_norm / ModelList model) would silently leave the model at its random
    initialisation: that raises."""
    state = _load_pdparams(path)
    result = model.set_state_dict(state)
    if not isinstance(result, tuple) or len(result) != 2:
        return
    missing, unexpected = result
    if missing:
        logger.warning(f"{path}: {len(missing)} parameter(s) not found in the file and left unchanged: {list(missing)[:8]}")
    if unexpected:
        logger.warning(f"{path}: {len(unexpected)} key(s) in the file match no parameter: {list(unexpected)[:8]}")
    if state and len(unexpected) == len(state):
        raise ValueError(f"{path}: no key of the file matches a parameter of the model")


def save_checkpoint(model, optimizer, metric: Optional[Dict[str, float]] = None, grad_scaler=None,
                    output_dir: Optional[str] = None, prefix: str = "model", equation=None, print_log: bool = True,
                    ema_model=None, aggregator=None):
    if not _rank0():
        return
    if output_dir is None:
        logger.warning("output_dir is None, skip save_checkpoint")
        return
    ckpt_dir = os.path.join(output_dir, "checkpoints")
    os.makedirs(ckpt_dir, exist_ok=True)
    path = os.path.join(ckpt_dir, prefix)
    _save_pdparams(path + ".pdparams", _state_arrays(model))
    if optimizer is not None:
        # the optimizer's WHOLE state (paddle.save(optimizer.state_dict()), save_load.py:246): every moment / buffer
        # tensor, the step count, the moments of learnable equation parameters and the LR scheduler's position
        st = optimizer.state_dict()
        if isinstance(st, list):  # OptimizerList: one state dict per optimizer
            st = {f"opt{i}.{k}": v for i, d in enumerate(st) for k, v in d.items()}
        arrays = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in st.items()}
        if aggregator is not None and getattr(aggregator, "should_persist", False):
            for k, v in aggregator.state_dict().items():  # GradNorm / NTK weights (save_load.py:277-279, .pdagg)
                arrays["agg_" + k] = np.asarray(v)
        _save_npz(path + ".pdopt", arrays)
    if equation is not None and sum(len(eq.learnable_parameters) for eq in equation.values()) > 0:
        # save_load.py:267-276: {equation name: ParameterList state dict}
        with open(path + ".pdeqn", "wb") as f:
            pickle.dump({key: {k: np.asarray(v) for k, v in eq.state_dict().items()} for key, eq in equation.items()},
                        f, protocol=4)
    with open(path + ".pdstates", "w") as f:
        json.dump({"metric": float(metric["metric"]) if metric else float("inf"),
                   "epoch": int(metric["epoch"]) if metric else 0}, f)
    if print_log:
        logger.message(f"Finish saving checkpoint to: {path}")


def load_checkpoint(path: str, model, optimizer=None, equation=None, grad_scaler=None, ema_model=None, aggregator=None
                    ) -> Dict[str, float]:
    if not os.path.exists(f"{path}.pdparams"):
        raise FileNotFoundError(f"{path}.pdparams not exist.")
    _set_model_state(model, f"{path}.pdparams")
    if optimizer is not None and os.path.exists(f"{path}.pdopt"):
        st = _load_npz(f"{path}.pdopt")
        optimizer.set_state_dict({k: (v.item() if v.ndim == 0 else v) for k, v in st.items() if not k.startswith("agg_")})
        if aggregator is not None and getattr(aggregator, "should_persist", False):
            agg = {k[4:]: v for k, v in st.items() if k.startswith("agg_")}
            if agg:
                aggregator.set_state_dict(agg)
    _load_equation(path, equation)
    with open(f"{path}.pdstates") as f:
        metric = json.load(f)
    logger.message(f"Finish loading checkpoint from {path}")
    return metric


def load_pretrain(model, path: str, equation=None):
    if path.startswith("http"):
        raise NotImplementedError("downloading pretrained weights needs network access")
    path = path[:-len(".pdparams")] if path.endswith(".pdparams") else path
    if not os.path.exists(f"{path}.pdparams"):
        raise FileNotFoundError(f"{path}.pdparams not exist.")
    _set_model_state(model, f"{path}.pdparams")
    logger.message(f"Finish loading pretrained model from: {path}.pdparams")
    _load_equation(path, equation)
