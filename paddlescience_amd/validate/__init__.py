"""ppsci.validate: GeometryValidator / SupervisedValidator
(/root/reference/ppsci/validate/{base,geo_validator,sup_validator}.py)."""
from __future__ import annotations

from typing import Any, Callable, Dict, Optional, Union

import numpy as np

from .. import data, geometry
from ..constraint.base import prepare_label_weight
from ..data import dataset


class Validator:
    def __init__(self, _dataset, dataloader_cfg, loss, metric, name):
        self.data_loader = data.build_dataloader(_dataset, dataloader_cfg)
        self.loss, self.metric, self.name = loss, metric, name

    def __str__(self):
        return ", ".join([self.__class__.__name__, f"name = {self.name}", f"input_keys = {self.input_keys}",
                          f"output_keys = {self.output_keys}", f"output_expr = {self.output_expr}",
                          f"len(dataloader) = {len(self.data_loader)}", f"loss = {self.loss}",
                          f"metric = {list(self.metric.keys())}"])


class GeometryValidator(Validator):  # geo_validator.py:35-161
    def __init__(self, output_expr: Dict[str, Callable], label_dict: Dict[str, Union[float, Callable]], geom,
                 dataloader_cfg: Dict[str, Any], loss, random: str = "pseudo", criteria: Optional[Callable] = None,
                 evenly: bool = False, metric=None, with_initial: bool = False, name: Optional[str] = None):
        self.output_expr, self.label_dict = output_expr, label_dict
        self.input_keys, self.output_keys = geom.dim_keys, tuple(label_dict.keys())
        nx = dataloader_cfg["total_size"]
        self.num_timestamps = 1
        if isinstance(geom, geometry.TimeXGeometry):
            nts = getattr(geom.timedomain, "num_timestamps", None)
            if nts is None:
                raise NotImplementedError("TimeXGeometry with random timestamp not implemented yet.")
            self.num_timestamps = nts if with_initial else nts - 1
            assert nx % self.num_timestamps == 0, f"{nx} % {self.num_timestamps} != 0"
            nx //= self.num_timestamps
            input = geom.sample_interior(nx * (nts - 1), random, criteria, evenly)
            if with_initial:
                initial = geom.sample_initial_interior(nx, random, criteria, evenly)
                input = {k: np.vstack((initial[k], input[k])) for k in input}
        else:
            input = geom.sample_interior(nx, random, criteria, evenly)
        label, _ = prepare_label_weight(input, label_dict, None, geom.dim_keys)
        weight = {k: np.ones_like(next(iter(label.values()))) for k in label}
        ds_name = dataloader_cfg["dataset"] if isinstance(dataloader_cfg["dataset"], str) else dataloader_cfg["dataset"]["name"]
        super().__init__(getattr(dataset, ds_name)(input, label, weight), dataloader_cfg, loss, metric, name)


class SupervisedValidator(Validator):
    def __init__(self, dataloader_cfg: Dict[str, Any], loss, output_expr: Optional[Dict[str, Callable]] = None,
                 metric=None, name: Optional[str] = None):
        _dataset = dataset.build_dataset(dataloader_cfg["dataset"])
        self.input_keys = _dataset.input_keys
        self.output_keys = tuple(output_expr.keys()) if output_expr is not None else _dataset.label_keys
        self.output_expr = output_expr if output_expr is not None else {k: (lambda out, k=k: out[k]) for k in self.output_keys}
        super().__init__(_dataset, dataloader_cfg, loss, metric, name)


__all__ = ["Validator", "GeometryValidator", "SupervisedValidator"]
