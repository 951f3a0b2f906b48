"""ppsci.loss.{L1Loss, MAELoss, L2Loss, L2RelLoss} (/root/reference/ppsci/loss/l1.py:27-118, mae.py:27-108,
l2.py:27-113, :239-310) for the `[N, 1]` variables of the PINN path, where the norm over the feature axis is the
absolute value.  On the fused path they are epilogue loss kinds (`term_kind`, include/ppsci_hip.h PPSCI_LOSS_*);
`forward` is the same arithmetic in torch for host-side use (validators, tests)."""
from typing import Dict, Optional, Union

import torch

from .. import hotpath as hp
from .base import Loss


class _PointLoss(Loss):
    term_kind = hp.LOSS_ABS

    def __init__(self, reduction: str = "mean", weight: Optional[Union[float, Dict[str, float]]] = None):
        if reduction not in ["mean", "sum"]:
            raise ValueError(f"reduction should be 'mean' or 'sum', but got {reduction}")
        super().__init__(reduction, weight)

    def key_weight(self, key: str) -> float:
        if isinstance(self.weight, (float, int)):
            return float(self.weight)
        if isinstance(self.weight, dict) and key in self.weight:
            return float(self.weight[key])
        return 1.0

    def term_scale(self, key: str, n_global: int) -> float:
        return self.key_weight(key) / (n_global if self.reduction == "mean" else 1.0)

    def _point(self, x, y, w, area):
        raise NotImplementedError

    def forward(self, output_dict, label_dict, weight_dict=None) -> Dict[str, torch.Tensor]:
        losses = {}
        for key in label_dict:
            w = weight_dict[key] if (weight_dict and key in weight_dict) else None
            loss = self._point(output_dict[key], label_dict[key], w, output_dict.get("area"))
            loss = loss.sum() if self.reduction == "sum" else loss.mean()
            losses[key] = loss * self.key_weight(key)
        return losses


class MAELoss(_PointLoss):
    def _point(self, x, y, w, area):
        loss = (x - y).abs()
        if w is not None:
            loss = loss * w
        return loss * area if area is not None else loss


class L1Loss(MAELoss):
    def _point(self, x, y, w, area):
        return super()._point(x, y, w, area).sum(dim=1)


class L2Loss(_PointLoss):
    term_kind = hp.LOSS_SQRTABS

    def _point(self, x, y, w, area):
        loss = (x - y) ** 2
        if w is not None:
            loss = loss * w
        if area is not None:
            loss = loss * area
        return loss.sum(dim=1).sqrt()


class L2RelLoss(_PointLoss):
    term_kind = hp.LOSS_ABSREL

    def _point(self, x, y, w, area):
        # l2.py:296-297: the [N] relative errors times the [N, 1] weights broadcast to [N, N]; its mean / sum is
        # mean(err) * mean(w) / sum(err) * sum(w), which the fused path reproduces through `batch_weight`
        n = x.shape[0]
        loss = torch.linalg.norm((x - y).reshape(n, -1), dim=1) / torch.linalg.norm(y.reshape(n, -1), dim=1)
        return loss * w if w is not None else loss

    def batch_weight(self, w):
        """The per-point weight column the epilogue multiplies by, such that its reduction equals the reference's
        reduction of the broadcast [N, N] product (see `_point`)."""
        if isinstance(w, torch.Tensor):
            if w.dim() < 2:
                return w
            return torch.full_like(w, 1.0) * (w.mean() if self.reduction == "mean" else w.sum())
        import numpy as np

        w = np.asarray(w)
        if w.ndim < 2:
            return w
        return np.full_like(w, w.mean(dtype=np.float64) if self.reduction == "mean" else w.sum(dtype=np.float64))
