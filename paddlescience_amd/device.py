"""Device selection.  The product runs on the GPU only; CPU tensors are used solely when tests inject the
CPU SIMT emulator build of the kernels (paddlescience_amd._lib._inject_for_tests)."""
import torch

from . import _lib

_forced = None


def set_device(dev) -> None:
    global _forced
    _forced = torch.device(dev) if dev is not None else None


def get_device() -> torch.device:
    if _forced is not None:
        return _forced
    if torch.cuda.is_available():
        return torch.device("cuda", torch.cuda.current_device())
    if _lib.is_emulated():
        return torch.device("cpu")
    raise RuntimeError("no GPU visible: paddlescience_amd runs on MI355X only (no CPU fallback)")
