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


class PeriodicConstraint(Constraint):
    """/root/reference/ppsci/constraint/periodic_constraint.py:60-166: boundary points and their images on the opposite
    side along `periodic_key`, interleaved per iteration so that every batch is [half ; images of that half]; labels are
    zeros (the Periodic*Loss compares the two halves of each output)."""

    def __init__(self, output_expr: Dict[str, Callable], label_dict: Dict[str, Union[float, Callable]], geom,
                 periodic_key: str, dataloader_cfg: Dict[str, Any], loss, random: str = "pseudo",
                 criteria: Optional[Callable] = None, evenly: bool = False,
                 weight_dict: Optional[Dict[str, Callable]] = None, name: str = "PeriodicBC"):
        import numpy as np

        from ..utils.misc import DEFAULT_DTYPE

        self.input_keys = geom.dim_keys
        self.output_keys = tuple(output_expr.keys())
        self.output_expr = dict(output_expr)
        bs, iters = dataloader_cfg["batch_size"], dataloader_cfg["iters_per_epoch"]
        if bs % 2 > 0:
            raise ValueError(f"batch_size({bs}) should be positive and even when using PeriodicConstraint")
        if dataloader_cfg.get("shuffle", False):
            raise ValueError("shuffle should be False when using PeriodicConstraint")
        half = bs // 2
        side = geom.sample_boundary(half * iters, random, _crit(criteria), evenly)
        if "area" in side:
            side["area"] *= iters
        space = getattr(geom, "geometry", geom)  # TimeXGeometry: the component indexes the spatial keys
        image = geom.periodic_point(side, space.dim_keys.index(periodic_key))
        mixed = {k: np.vstack([part[k][i * half:(i + 1) * half] for i in range(iters) for part in (side, image)])
                 for k in side}
        n_rows = next(iter(mixed.values())).shape[0]
        label = {k: np.full((n_rows, 1), 0, DEFAULT_DTYPE) for k in label_dict}
        weight = None
        if weight_dict is not None:
            _, weight = prepare_label_weight(mixed, label_dict, weight_dict, geom.dim_keys)
        Constraint.__init__(self, finish_dataset(dataloader_cfg, mixed, label, weight), dataloader_cfg, loss, name)
