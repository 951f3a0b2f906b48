"""ppsci.constraint.Constraint (/root/reference/ppsci/constraint/base.py:29-62) plus the label / weight
preparation that Interior / Boundary / Initial constraints share (interior_constraint.py:112-165)."""
from __future__ import annotations

from typing import Any, Callable, Dict, Optional

import numpy as np
import sympy

from .. import data


class Constraint:
    def __init__(self, dataset, dataloader_cfg: Dict[str, Any], loss, name: str):
        self.data_loader = data.build_dataloader(dataset, dataloader_cfg)
        self.data_iter = iter(data.InfiniteDataLoader(self.data_loader))
        self.loss = loss
        self.name = name

    def __str__(self):
        return ", ".join([self.__class__.__name__, f"name = {self.name}", f"input_keys = {self.input_keys}",
                          f"output_keys = {self.output_keys}", f"output_expr = {self.output_expr}",
                          f"label_dict = {getattr(self, 'label_dict', None)}", f"loss = {self.loss}"])


_AMAX = [{"amax": lambda xy, _axis=None: np.maximum(xy[0], xy[1])}, "numpy"]


def _evaluate(value, input: Dict[str, np.ndarray], dim_keys, like: np.ndarray, as_float: bool = False):
    """number | sympy expression of the coordinates | callable(input dict) -> [n,1] array."""
    if isinstance(value, (int, float)):
        return np.full_like(like, float(value) if as_float else value)
    if isinstance(value, sympy.Basic):
        fn = sympy.lambdify(sympy.symbols(dim_keys), value, _AMAX)
        return fn(**{k: v for k, v in input.items() if k in dim_keys})
    if callable(value):
        out = value(input)
        if isinstance(out, (int, float)):
            out = np.full_like(next(iter(input.values())), out)
        return out
    raise NotImplementedError(f"type of {type(value)} is invalid yet.")


def prepare_label_weight(input: Dict[str, np.ndarray], label_dict, weight_dict, dim_keys):
    first_in = next(iter(input.values()))
    label = {k: _evaluate(v, input, dim_keys, first_in) for k, v in label_dict.items()}
    weight = None
    if weight_dict is not None:
        first_lab = next(iter(label.values()))
        weight = {k: np.ones_like(first_lab) for k in label}
        for k, v in weight_dict.items():
            if isinstance(v, str):
                if v != "sdf":
                    raise NotImplementedError(f"string {v} is invalid yet.")
                weight[k] = input["sdf"]
            else:
                weight[k] = _evaluate(v, input, dim_keys, first_lab, as_float=True)
    return label, weight


def finish_dataset(dataloader_cfg: Dict[str, Any], input, label, weight):
    from ..data import dataset

    if isinstance(dataloader_cfg["dataset"], str):
        dataloader_cfg["dataset"] = {"name": dataloader_cfg["dataset"]}
    dataloader_cfg["dataset"].update({"input": input, "label": label, "weight": weight})
    return dataset.build_dataset(dataloader_cfg["dataset"])
