"""ppsci.metric: MSE, RMSE, MAE, L2Rel (/root/reference/ppsci/metric/{mse,rmse,mae,l2_rel}.py) on torch tensors
(evaluation-time host/torch arithmetic; not part of the training hot path)."""
from typing import Dict

import numpy as np
import torch


from .base import Metric  # noqa: E402


class MSE(Metric):
    def forward(self, output_dict, label_dict) -> Dict[str, torch.Tensor]:
        out = {}
        for key in label_dict:
            mse = (output_dict[key] - label_dict[key]) ** 2
            out[key] = mse.mean(dim=tuple(range(1, mse.ndim))) if self.keep_batch else mse.mean()
        return out


class RMSE(Metric):
    def __init__(self, keep_batch: bool = False):
        if keep_batch:
            raise ValueError(f"keep_batch should be False, but got {keep_batch}.")
        super().__init__(keep_batch)

    def forward(self, output_dict, label_dict):
        return {k: ((output_dict[k] - label_dict[k]) ** 2).mean() ** 0.5 for k in label_dict}


class MAE(Metric):
    def forward(self, output_dict, label_dict):
        out = {}
        for key in label_dict:
            mae = (output_dict[key] - label_dict[key]).abs()
            out[key] = mae.mean(dim=tuple(range(1, mae.ndim))) if self.keep_batch else mae.mean()
        return out


class L2Rel(Metric):
    EPS: float = float(np.finfo(np.float32).eps)  # l2_rel.py:59-61

    def __init__(self, keep_batch: bool = False):
        if keep_batch:
            raise ValueError(f"keep_batch should be False, but got {keep_batch}.")
        super().__init__(keep_batch)

    def forward(self, output_dict, label_dict):
        return {k: torch.linalg.vector_norm(label_dict[k] - output_dict[k])
                / torch.linalg.vector_norm(label_dict[k]).clamp(min=self.EPS) for k in label_dict}


__all__ = ["Metric", "MSE", "RMSE", "MAE", "L2Rel", "FunctionalMetric"]


class FunctionalMetric(Metric):
    """ppsci.metric.FunctionalMetric (/root/reference/ppsci/metric/func.py): user function of
    (output_dict, label_dict) returning a dict of tensors."""

    def __init__(self, metric_expr, keep_batch: bool = False):
        super().__init__(keep_batch)
        self.metric_expr = metric_expr

    def forward(self, output_dict, label_dict):
        return self.metric_expr(output_dict, label_dict)
