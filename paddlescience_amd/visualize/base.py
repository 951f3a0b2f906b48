"""ppsci.visualize.base.Visualizer (/root/reference/ppsci/visualize/base.py:24-65)."""
from __future__ import annotations

from typing import Callable, Dict

import numpy as np


class Visualizer:
    """Points to evaluate (`input_dict`), what to evaluate there (`output_expr`), how to write it (`save`)."""

    def __init__(self, input_dict: Dict[str, np.ndarray], output_expr: Dict[str, Callable], batch_size: int,
                 num_timestamps: int, prefix: str):
        self.input_dict = input_dict
        self.input_keys = tuple(input_dict.keys())
        self.output_expr = output_expr
        self.output_keys = tuple(output_expr.keys())
        self.batch_size = batch_size
        self.num_timestamps = num_timestamps
        self.prefix = prefix

    def save(self, filename: str, data_dict: Dict[str, np.ndarray]):
        raise NotImplementedError(f"{type(self).__name__}.save")

    def __str__(self):
        return ", ".join([f"input_keys: {self.input_keys}", f"output_keys: {self.output_keys}",
                          f"output_expr: {self.output_expr}", f"batch_size: {self.batch_size}",
                          f"num_timestamps: {self.num_timestamps}", f"output file prefix: {self.prefix}"])
