"""Interior / Boundary / Initial constraints: sample `batch_size * iters_per_epoch` points from the
geometry ONCE, evaluate labels and weights on the host, wrap them in an array dataset
(/root/reference/ppsci/constraint/interior_constraint.py:77-174, boundary_constraint.py:75-163,
initial_constraint.py)."""
from __future__ import annotations

from typing import Any, Callable, Dict, Optional, Union

from .base import Constraint, finish_dataset, prepare_label_weight


class _GeometricConstraint(Constraint):
    def _setup(self, output_expr, label_dict, geom, dataloader_cfg, loss, input, weight_dict, name):
        self.label_dict = label_dict
        self.input_keys = geom.dim_keys
        self.output_keys = tuple(label_dict.keys())
        self.output_expr = {k: v for k, v in output_expr.items() if k in self.output_keys}
        if "area" in input:
            input["area"] *= dataloader_cfg["iters_per_epoch"]
        label, weight = prepare_label_weight(input, label_dict, weight_dict, geom.dim_keys)
        Constraint.__init__(self, finish_dataset(dataloader_cfg, input, label, weight), dataloader_cfg, loss, name)


def _crit(criteria):
    return eval(criteria) if isinstance(criteria, str) else criteria


class InteriorConstraint(_GeometricConstraint):
    def __init__(self, output_expr: Dict[str, Callable], label_dict: Dict[str, Union[float, Callable]], geom,
                 dataloader_cfg: Dict[str, Any], loss, random: str = "pseudo", criteria: Optional[Callable] = None,
                 evenly: bool = False, weight_dict: Optional[Dict[str, Union[Callable, float]]] = None,
                 compute_sdf_derivatives: bool = False, name: str = "EQ"):
        n = dataloader_cfg["batch_size"] * dataloader_cfg["iters_per_epoch"]
        input = geom.sample_interior(n, random, _crit(criteria), evenly, compute_sdf_derivatives)
        self._setup(output_expr, label_dict, geom, dataloader_cfg, loss, input, weight_dict, name)


class BoundaryConstraint(_GeometricConstraint):
    def __init__(self, output_expr: Dict[str, Callable], label_dict: Dict[str, Union[float, Callable]], geom,
                 dataloader_cfg: Dict[str, Any], loss, random: str = "pseudo", criteria: Optional[Callable] = None,
                 evenly: bool = False, weight_dict: Optional[Dict[str, Union[float, Callable]]] = None, name: str = "BC"):
        n = dataloader_cfg["batch_size"] * dataloader_cfg["iters_per_epoch"]
        input = geom.sample_boundary(n, random, _crit(criteria), evenly)
        self._setup(output_expr, label_dict, geom, dataloader_cfg, loss, input, weight_dict, name)


class InitialConstraint(_GeometricConstraint):
    def __init__(self, output_expr: Dict[str, Callable], label_dict: Dict[str, Union[float, Callable]], geom,
                 dataloader_cfg: Dict[str, Any], loss, random: str = "pseudo", criteria: Optional[Callable] = None,
                 evenly: bool = False, weight_dict: Optional[Dict[str, Callable]] = None,
                 compute_sdf_derivatives: bool = False, name: str = "IC"):
        n = dataloader_cfg["batch_size"] * dataloader_cfg["iters_per_epoch"]
        input = geom.sample_initial_interior(n, random, _crit(criteria), evenly, compute_sdf_derivatives)
        self._setup(output_expr, label_dict, geom, dataloader_cfg, loss, input, weight_dict, name)
