"""Minimal rank-0 logger with the call surface of /root/reference/ppsci/utils/logger.py (init_logger,
info / message / warning / error / debug, scalar)."""
from __future__ import annotations

import logging
import os
import sys
from typing import Dict, Optional

_logger: Optional[logging.Logger] = None
MESSAGE = 25
logging.addLevelName(MESSAGE, "MESSAGE")


def _rank() -> int:
    try:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            return dist.get_rank()
    except Exception:  # noqa: BLE001
        pass
    return int(os.environ.get("RANK", "0"))


def _level(log_level) -> int:
    """logger.py:84-85: a level may be given by name in any case ("info", "DEBUG", "message")."""
    if isinstance(log_level, str):
        name = log_level.upper()
        if name == "MESSAGE":
            return MESSAGE
        lvl = getattr(logging, name, None)
        if not isinstance(lvl, int):
            raise ValueError(f"unknown log level {log_level!r}")
        return lvl
    return int(log_level)


def init_logger(name: str = "ppsci", log_file: Optional[str] = None, log_level=logging.INFO) -> None:
    global _logger
    log_level = _level(log_level)
    _logger = logging.getLogger(name)
    _logger.handlers.clear()
    _logger.setLevel(log_level if _rank() == 0 else logging.ERROR)
    h = logging.StreamHandler(sys.stdout)
    h.setFormatter(logging.Formatter("[%(asctime)s] %(name)s %(levelname)s: %(message)s", "%Y/%m/%d %H:%M:%S"))
    _logger.addHandler(h)
    if log_file is not None and _rank() == 0:
        os.makedirs(os.path.dirname(log_file) or ".", exist_ok=True)
        fh = logging.FileHandler(log_file, "a")
        fh.setFormatter(h.formatter)
        _logger.addHandler(fh)
    _logger.propagate = False


def _get() -> logging.Logger:
    if _logger is None:
        init_logger()
    return _logger


def set_log_level(level):
    _get().setLevel(_level(level))


def debug(msg, *args):
    _get().debug(msg, *args)


def info(msg, *args):
    _get().info(msg, *args)


def message(msg, *args):
    _get().log(MESSAGE, msg, *args)


def warning(msg, *args):
    _get().warning(msg, *args)


def error(msg, *args):
    _get().error(msg, *args)


def scalar(metric_dict: Dict[str, float], step: int, vdl_writer=None, wandb_writer=None, tbd_writer=None):
    """logger.py:200-231: fan out scalars to the optional writers (none are bundled here)."""
    for w in (vdl_writer, tbd_writer):
        if w is not None:
            for k, v in metric_dict.items():
                w.add_scalar(k, v, step)
    if wandb_writer is not None:
        wandb_writer.log({"step": step, **metric_dict})
