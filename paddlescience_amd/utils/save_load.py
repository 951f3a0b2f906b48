"""Checkpointing with the file set and key names of /root/reference/ppsci/utils/save_load.py:213-290:
`<output_dir>/checkpoints/<prefix>.{pdparams,pdopt,pdstates}` written by rank 0 only.  The payload is a
numpy .npz archive (paddle's pickle format needs PaddlePaddle); parameter keys are the reference's
(`linears.0.weight`, ..., `last_fc.bias`, mlp.py:264-277) so a converter to/from .pdparams is a pure
renaming-free re-serialisation."""
from __future__ import annotations

import json
import os
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
    _save_npz(path + ".pdparams", {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()})
    if optimizer is not None:
        st = optimizer.state_dict()
        _save_npz(path + ".pdopt", {"m": st["m"].detach().cpu().numpy(), "v": st["v"].detach().cpu().numpy(),
                                    "t": np.asarray(st["t"])})
    with open(path + ".pdstates", "w") as f:
        json.dump({"metric": float(metric["metric"]) if metric else float("inf"),
                   "epoch": int(metric["epoch"]) if metric else 0}, f)
    if print_log:
        logger.message(f"Finish saving checkpoint to: {path}")


def load_checkpoint(path: str, model, optimizer=None, equation=None, grad_scaler=None, ema_model=None, aggregator=None
                    ) -> Dict[str, float]:
    if not os.path.exists(f"{path}.pdparams"):
        raise FileNotFoundError(f"{path}.pdparams not exist.")
    model.set_state_dict(_load_npz(f"{path}.pdparams"))
    if optimizer is not None and os.path.exists(f"{path}.pdopt"):
        st = _load_npz(f"{path}.pdopt")
        optimizer.set_state_dict({"m": st["m"], "v": st["v"], "t": int(st["t"])})
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
    model.set_state_dict(_load_npz(f"{path}.pdparams"))
    logger.message(f"Finish loading pretrained model from: {path}.pdparams")
