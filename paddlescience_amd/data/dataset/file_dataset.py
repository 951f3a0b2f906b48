"""File-backed point / label tables (/root/reference/ppsci/data/dataset/csv_dataset.py:30-287, mat_dataset.py, npz_dataset.py):
{CSV,Mat,NPZ}Dataset (batch-indexed) and Iterable{CSV,Mat,NPZ}Dataset (one full batch per iteration).  After reading they ARE
the named-array datasets of array_dataset.py; what they add is `alias_dict`, `weight_dict` (a number or a callable of the input
dict per label key) and `timestamps`: a table with a "t" column is filtered to the given time stamps (in their order), a table
without one is repeated at every time stamp with a new leading input key "t" (time-major)."""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple, Union

import numpy as np

from ...utils import misc, reader
from .array_dataset import IterableNamedArrayDataset, NamedArrayDataset


def _load(loader, file_path, input_keys, label_keys, alias_dict, weight_dict, timestamps):
    input_keys, label_keys = tuple(input_keys), tuple(label_keys)
    raw = loader(file_path, input_keys + label_keys, alias_dict)
    if timestamps is not None:
        if "t" in raw:  # csv_dataset.py:94-108: rows at the given time stamps, in the order of `timestamps`
            t = raw["t"]
            mask = np.concatenate([np.nonzero(np.isclose(t, ti).flatten())[0] for ti in timestamps], 0)
            arr = misc.convert_to_array(raw, input_keys + label_keys)[mask]
        else:  # csv_dataset.py:109-119: every row at every time stamp
            arr = misc.combine_array_with_time(misc.convert_to_array(raw, input_keys + label_keys), timestamps)
            input_keys = ("t",) + input_keys
        raw = misc.convert_to_dict(arr, input_keys + label_keys)
    inp = {k: v for k, v in raw.items() if k in input_keys}
    lab = {k: v for k, v in raw.items() if k in label_keys}
    weight: Dict[str, np.ndarray] = {}
    if weight_dict is not None:  # csv_dataset.py:130-152
        like = next(iter(lab.values()))
        weight = {k: np.ones_like(like) for k in lab}
        for k, value in weight_dict.items():
            if isinstance(value, (int, float)):
                weight[k] = np.full_like(like, value)
            elif callable(value):
                w = value(inp)
                weight[k] = np.full_like(like, w) if isinstance(w, (int, float)) else w
            else:
                raise NotImplementedError(f"type of {type(value)} is invalid yet.")
    return inp, lab, weight


def _make(loader, iterable: bool, name: str):
    base = IterableNamedArrayDataset if iterable else NamedArrayDataset

    class _FileDataset(base):
        def __init__(self, file_path: str, input_keys: Tuple[str, ...], label_keys: Tuple[str, ...] = (),
                     alias_dict: Optional[Dict[str, str]] = None,
                     weight_dict: Optional[Dict[str, Union[Callable, float]]] = None,
                     timestamps: Optional[Tuple[float, ...]] = None, transforms=None):
            inp, lab, weight = _load(loader, file_path, input_keys, label_keys, alias_dict, weight_dict, timestamps)
            super().__init__(inp, lab, weight if (weight or not iterable) else None, transforms)

    _FileDataset.__name__ = _FileDataset.__qualname__ = name
    return _FileDataset


def _npz_columns(file_path, keys, alias_dict=None):
    return reader.load_npz_file(file_path, keys, alias_dict)


CSVDataset = _make(reader.load_csv_file, False, "CSVDataset")
IterableCSVDataset = _make(reader.load_csv_file, True, "IterableCSVDataset")
MatDataset = _make(reader.load_mat_file, False, "MatDataset")
IterableMatDataset = _make(reader.load_mat_file, True, "IterableMatDataset")
NPZDataset = _make(_npz_columns, False, "NPZDataset")
IterableNPZDataset = _make(_npz_columns, True, "IterableNPZDataset")
