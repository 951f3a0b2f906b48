"""ppsci.utils.reader (/root/reference/ppsci/utils/reader.py:31-260): the file formats on the input side of the hot path --
*.csv, *.mat, *.npz point / label tables -> {key: float32 [n, 1] array}.  `alias_dict` maps the caller's key to the column /
variable name in the file ({inner_key: outer_key}); integer tables keep their dtype, everything else becomes float32.
load_dat_file unpickles arbitrary objects in the reference; here only a pickled dict of numpy arrays is accepted (the same
restricted unpickler as the checkpoint reader).  *.vtu readers need meshio, which is not in this image: they raise."""
from __future__ import annotations

import collections
import csv
from typing import Dict, Optional, Tuple

import numpy as np

__all__ = ["load_csv_file", "load_mat_file", "load_npz_file", "load_dat_file", "load_vtk_file", "load_vtk_with_time_file"]

_DTYPE = "float32"  # paddle.get_default_dtype()


def _fetch(raw, keys, alias_dict, column: bool) -> Dict[str, np.ndarray]:
    alias_dict = alias_dict or {}
    out = {}
    for key in keys:
        fetch_key = alias_dict[key] if key in alias_dict else key
        if fetch_key not in raw:
            raise KeyError(f"fetch_key({fetch_key}) do not exist in raw_data.")
        a = np.asarray(raw[fetch_key])
        if column:  # reader.py:83-86, :120-123
            if not np.issubdtype(a.dtype, np.integer):
                a = a.astype(_DTYPE)
            a = a.reshape([-1, 1])
        elif a.dtype in (np.float16, np.float32, np.float64):  # load_npz_file keeps the stored shape (reader.py:156-158)
            a = a.astype(_DTYPE)
        out[key] = a
    return out


def load_csv_file(file_path: str, keys: Tuple[str, ...], alias_dict: Optional[Dict[str, str]] = None, delimiter: str = ",",
                  encoding: str = "utf-8") -> Dict[str, np.ndarray]:
    raw = collections.defaultdict(list)
    with open(file_path, "r", encoding=encoding) as f:
        for line in csv.DictReader(f, delimiter=delimiter):
            for k, v in line.items():
                raw[k].append(v)
    return _fetch(raw, keys, alias_dict, True)


def load_mat_file(file_path: str, keys: Tuple[str, ...], alias_dict: Optional[Dict[str, str]] = None) -> Dict[str, np.ndarray]:
    import scipy.io as sio

    return _fetch(sio.loadmat(file_path), keys, alias_dict, True)


def load_npz_file(file_path: str, keys: Tuple[str, ...], alias_dict: Optional[Dict[str, str]] = None) -> Dict[str, np.ndarray]:
    with np.load(file_path, allow_pickle=False) as raw:
        return _fetch(raw, keys, alias_dict, False)


def load_dat_file(file_path: str, keys: Optional[Tuple[str, ...]] = None,
                  alias_dict: Optional[Dict[str, str]] = None) -> Dict[str, np.ndarray]:
    from .save_load import _ArrayUnpickler

    with open(file_path, "rb") as f:
        raw = _ArrayUnpickler(f).load()
    if keys is None:
        keys = tuple(raw.keys())
    return _fetch(raw, keys, alias_dict, False)


def load_vtk_file(*args, **kwargs):
    raise NotImplementedError("*.vtu readers need meshio, which is not available in this environment")


load_vtk_with_time_file = load_vtk_file
