"""ppsci.constraint.SupervisedConstraint (/root/reference/ppsci/constraint/supervised_constraint.py:56-80)."""
from __future__ import annotations

from typing import Any, Callable, Dict, Optional

from ..data import dataset
from .base import Constraint


class SupervisedConstraint(Constraint):
    def __init__(self, dataloader_cfg: Dict[str, Any], loss, output_expr: Optional[Dict[str, Callable]] = None,
                 name: str = "Sup"):
        _dataset = dataset.build_dataset(dataloader_cfg["dataset"])
        self.input_keys = _dataset.input_keys
        self.output_keys = tuple(output_expr.keys()) if output_expr is not None else _dataset.label_keys
        self.output_expr = output_expr
        if self.output_expr is None:
            self.output_expr = {key: (lambda out, k=key: out[k]) for key in self.output_keys}
        super().__init__(_dataset, dataloader_cfg, loss, name)

    def __str__(self):
        return ", ".join([self.__class__.__name__, f"name = {self.name}", f"input_keys = {self.input_keys}",
                          f"output_keys = {self.output_keys}", f"output_expr = {self.output_expr}", f"loss = {self.loss}"])
