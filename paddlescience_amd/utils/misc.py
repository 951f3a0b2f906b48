"""Host-side helpers with the names and semantics of /root/reference/ppsci/utils/misc.py."""
from __future__ import annotations

import collections
import random
import time
from contextlib import ContextDecorator
from typing import Callable, Dict, List, Sequence, Tuple, Union

import numpy as np
import torch

DEFAULT_DTYPE = "float32"  # paddle.get_default_dtype() in the reference


class AverageMeter:  # misc.py:59-110
    def __init__(self, name="", fmt="f", postfix="", need_avg=True):
        self.name, self.fmt, self.postfix, self.need_avg = name, fmt, postfix, need_avg
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0
        self.history = []

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count
        self.history.append(val)

    @property
    def avg_info(self):
        return f"{self.name}: {float(self.avg):.5f}"

    @property
    def total(self):
        return f"{self.name}_sum: {self.sum:{self.fmt}}{self.postfix}"

    @property
    def total_minute(self):
        return f"{self.name} {self.sum / 60:{self.fmt}}{self.postfix} min"

    @property
    def mean(self):
        return f"{self.name}: {self.avg:{self.fmt}}{self.postfix}" if self.need_avg else ""

    @property
    def value(self):
        return f"{self.name}: {self.val:{self.fmt}}{self.postfix}"


class PrettyOrderedDict(collections.OrderedDict):
    def __str__(self):
        return "".join([str((k, v)) for k, v in self.items()])


class Prettydefaultdict(collections.defaultdict):
    def __str__(self):
        return "".join([str((k, v)) for k, v in self.items()])


class Timer(ContextDecorator):  # misc.py:192-258
    interval: float

    def __init__(self, name: str = "Timer", auto_print: bool = True):
        self.name, self.auto_print = name, auto_print

    def __enter__(self):
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        self.start_time = time.perf_counter()
        return self

    def __exit__(self, *exc):
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        self.interval = time.perf_counter() - self.start_time
        if self.auto_print:
            print(f"{self.name}.time_cost = {self.interval:.2f} s")


def convert_to_dict(array: np.ndarray, keys: Tuple[str, ...]) -> Dict[str, np.ndarray]:
    if array.shape[-1] != len(keys):
        raise ValueError(f"dim of array({array.shape[-1]}) must equal to len(keys)({len(keys)})")
    parts = np.split(array, len(keys), axis=-1)
    return {k: parts[i] for i, k in enumerate(keys)}


def convert_to_array(dict_: Dict[str, np.ndarray], keys: Tuple[str, ...]) -> np.ndarray:
    return np.concatenate([dict_[k] for k in keys], axis=-1)


def concat_dict_list(dict_list: Sequence[Dict[str, np.ndarray]]) -> Dict[str, np.ndarray]:
    return {k: np.concatenate([d[k] for d in dict_list], axis=0) for k in dict_list[0].keys()}


def stack_dict_list(dict_list: Sequence[Dict[str, np.ndarray]]) -> Dict[str, np.ndarray]:
    return {k: np.stack([d[k] for d in dict_list], axis=0) for k in dict_list[0].keys()}


def typename(obj: object) -> str:
    return obj.__class__.__name__


def combine_array_with_time(x: np.ndarray, t: Tuple[int, ...]) -> np.ndarray:
    nx = len(x)
    return np.vstack([np.hstack((np.full([nx, 1], float(ti), dtype=DEFAULT_DTYPE), x)) for ti in t])


def cartesian_product(*arrays: np.ndarray) -> np.ndarray:
    la = len(arrays)
    arr = np.empty([len(a) for a in arrays] + [la], dtype=np.result_type(*arrays))
    for i, a in enumerate(np.ix_(*arrays)):
        arr[..., i] = a
    return arr.reshape(-1, la)


def set_random_seed(seed: int):  # misc.py:510-518
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)


def all_gather(tensor, concat: bool = True, axis: int = 0):  # misc.py:293-335
    import torch.distributed as dist

    if isinstance(tensor, dict):
        return {k: all_gather(v, concat, axis) for k, v in tensor.items()}
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return tensor
    out = [torch.empty_like(tensor) for _ in range(dist.get_world_size())]
    dist.all_gather(out, tensor.contiguous())
    return torch.cat(out, dim=axis) if concat else out


def run_at_rank0(func: Callable) -> Callable:
    import functools

    @functools.wraps(func)
    def wrapped(*args, **kwargs):
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0:
            return func(*args, **kwargs)
        return None

    return wrapped


def plot_curve(data: Dict[str, List], xlabel: str = "X", ylabel: str = "Y", output_dir: str = "./output/", smooth_step: int = 1,
               use_semilogy: bool = False) -> None:
    """misc.py:582-640 of the reference: the curves of `data` (equal lengths) over their index, every `smooth_step` points averaged
    into one, saved as `<output_dir>/<xlabel>-<ylabel>_curve.jpg`; without matplotlib the averaged arrays go to an .npz."""
    import os

    arr = np.stack([np.asarray(v, dtype=np.float64).reshape(-1) for v in data.values()], axis=1)
    keep = arr.shape[0] - arr.shape[0] % smooth_step
    if keep == 0:
        keep, smooth_step = arr.shape[0], 1
    arr = arr[:keep].reshape(-1, smooth_step, arr.shape[1]).mean(axis=1)
    os.makedirs(output_dir, exist_ok=True)
    stem = os.path.join(output_dir, f"{xlabel}-{ylabel}_curve")
    try:
        import matplotlib

        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except Exception:  # noqa: BLE001
        np.savez(stem + ".npz", x=np.arange(arr.shape[0]) * smooth_step, **{k: arr[:, i] for i, k in enumerate(data)})
        return
    fig = plt.figure()
    if use_semilogy:
        plt.yscale("log")
        plt.xscale("log")
    plt.plot(np.arange(arr.shape[0]) * smooth_step, arr)
    plt.legend(list(data.keys()), loc="upper left", bbox_to_anchor=(1, 1))
    plt.xlabel(xlabel)
    plt.ylabel(ylabel)
    plt.grid()
    plt.yticks(size=10)
    plt.xticks(size=10)
    fig.savefig(stem + ".jpg", dpi=200, bbox_inches="tight")
    plt.close(fig)
