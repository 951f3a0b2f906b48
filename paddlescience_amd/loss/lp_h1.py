"""LpLoss / H1Loss of the neural-operator examples (/root/reference/examples/neuraloperator/metric.py:69-412): relative /
absolute L^p and H^1 (value + central-difference gradient) errors per sample over the last `d` dimensions, reduced over
`reduce_dims`.  `LpLoss` / `H1Loss` are the validation metrics (rel error / batch size, keys "l2" / "h1"),
`LpLoss_train` / `H1Loss_train` the training losses (rel error summed over the batch, key "y").

Operator-learning path: plain tensor arithmetic on the device (the FNO engine differentiates the loss w.r.t. the network
output and hands that adjoint to the hand-written backward kernels)."""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Union

import torch


def central_diff_1d(x, h, fix_x_bnd: bool = False):
    dx = (torch.roll(x, -1, dims=-1) - torch.roll(x, 1, dims=-1)) / (2.0 * h)
    if fix_x_bnd:
        dx = dx.clone()
        dx[..., 0] = (x[..., 1] - x[..., 0]) / h
        dx[..., -1] = (x[..., -1] - x[..., -2]) / h
    return dx


def central_diff_2d(x, h, fix_x_bnd: bool = False, fix_y_bnd: bool = False):
    """metric.py:36-55."""
    if isinstance(h, float):
        h = [h, h]
    dx = (torch.roll(x, -1, dims=-2) - torch.roll(x, 1, dims=-2)) / (2.0 * h[0])
    dy = (torch.roll(x, -1, dims=-1) - torch.roll(x, 1, dims=-1)) / (2.0 * h[1])
    if fix_x_bnd:
        dx = dx.clone()
        dx[..., 0, :] = (x[..., 1, :] - x[..., 0, :]) / h[0]
        dx[..., -1, :] = (x[..., -1, :] - x[..., -2, :]) / h[0]
    if fix_y_bnd:
        dy = dy.clone()
        dy[..., :, 0] = (x[..., :, 1] - x[..., :, 0]) / h[1]
        dy[..., :, -1] = (x[..., :, -1] - x[..., :, -2]) / h[1]
    return dx, dy


def central_diff_3d(x, h, fix_x_bnd: bool = False, fix_y_bnd: bool = False, fix_z_bnd: bool = False):
    if isinstance(h, float):
        h = [h, h, h]
    dx = (torch.roll(x, -1, dims=-3) - torch.roll(x, 1, dims=-3)) / (2.0 * h[0])
    dy = (torch.roll(x, -1, dims=-2) - torch.roll(x, 1, dims=-2)) / (2.0 * h[1])
    dz = (torch.roll(x, -1, dims=-1) - torch.roll(x, 1, dims=-1)) / (2.0 * h[2])
    if fix_x_bnd:
        dx = dx.clone()
        dx[..., 0, :, :] = (x[..., 1, :, :] - x[..., 0, :, :]) / h[0]
        dx[..., -1, :, :] = (x[..., -1, :, :] - x[..., -2, :, :]) / h[0]
    if fix_y_bnd:
        dy = dy.clone()
        dy[..., :, 0, :] = (x[..., :, 1, :] - x[..., :, 0, :]) / h[1]
        dy[..., :, -1, :] = (x[..., :, -1, :] - x[..., :, -2, :]) / h[1]
    if fix_z_bnd:
        dz = dz.clone()
        dz[..., :, :, 0] = (x[..., :, :, 1] - x[..., :, :, 0]) / h[2]
        dz[..., :, :, -1] = (x[..., :, :, -1] - x[..., :, :, -2]) / h[2]
    return dx, dy, dz


class _Reduced:
    def _init_reduce(self, d, L, reduce_dims, reductions):
        self.d = d
        self.reduce_dims = [reduce_dims] if isinstance(reduce_dims, int) else reduce_dims
        if self.reduce_dims is not None:
            if isinstance(reductions, str):
                assert reductions == "sum" or reductions == "mean"
                self.reductions = [reductions] * len(self.reduce_dims)
            else:
                for r in reductions:
                    assert r == "sum" or r == "mean"
                self.reductions = reductions
        self.L = [L] * self.d if isinstance(L, float) else L

    def uniform_h(self, x):
        h = [0.0] * self.d
        for j in range(self.d, 0, -1):
            h[-j] = self.L[-j] / x.shape[-j]
        return h

    def reduce_all(self, x):
        for j in range(len(self.reduce_dims)):
            if self.reductions[j] == "sum":
                x = torch.sum(x, dim=self.reduce_dims[j], keepdim=True)
            else:
                x = torch.mean(x, dim=self.reduce_dims[j], keepdim=True)
        return x


class LpLoss(_Reduced):
    """metric.py:69-176."""

    def __init__(self, d: int = 1, p: int = 2, L: Union[float, Sequence[float]] = 2 * math.pi, reduce_dims=0,
                 reductions="sum"):
        self.p = p
        self._init_reduce(d, L, reduce_dims, reductions)

    def abs(self, x, y, h=None):
        h = self.uniform_h(x) if h is None else ([h] * self.d if isinstance(h, float) else h)
        const = math.prod(h) ** (1.0 / self.p)
        diff = const * torch.linalg.vector_norm(torch.flatten(x, -self.d) - torch.flatten(y, -self.d), ord=self.p, dim=-1)
        if self.reduce_dims is not None:
            diff = self.reduce_all(diff).squeeze()
        return diff

    def rel(self, x, y):
        diff = torch.linalg.vector_norm(torch.flatten(x, -self.d) - torch.flatten(y, -self.d), ord=self.p, dim=-1)
        ynorm = torch.linalg.vector_norm(torch.flatten(y, -self.d), ord=self.p, dim=-1)
        diff = diff / ynorm
        if self.reduce_dims is not None:
            diff = self.reduce_all(diff).squeeze()
        return diff

    def __call__(self, output_dict: Dict[str, torch.Tensor], label_dict: Dict[str, torch.Tensor]):
        x, y = output_dict["y"], label_dict["y"]
        return {"l2": self.rel(x, y) / x.shape[0]}


class LpLoss_train(LpLoss):  # noqa: N801 -- the reference's name (metric.py:179-193)
    def __call__(self, output_dict, label_dict, weight_dict=None):
        return {"y": self.rel(output_dict["y"], label_dict["y"])}


class H1Loss(_Reduced):
    """metric.py:196-383."""

    def __init__(self, d: int = 1, L: Union[float, Sequence[float]] = 2 * math.pi, reduce_dims=0, reductions="sum",
                 fix_x_bnd: bool = False, fix_y_bnd: bool = False, fix_z_bnd: bool = False):
        assert d > 0 and d < 4, "Currently only implemented for 1, 2, and 3-D."
        self.fix_x_bnd, self.fix_y_bnd, self.fix_z_bnd = fix_x_bnd, fix_y_bnd, fix_z_bnd
        self._init_reduce(d, L, reduce_dims, reductions)

    def compute_terms(self, x, y, h):
        if self.d == 1:
            return ({0: x, 1: central_diff_1d(x, h[0], self.fix_x_bnd)}, {0: y, 1: central_diff_1d(y, h[0], self.fix_x_bnd)})
        if self.d == 2:
            xs, ys = central_diff_2d(x, h, self.fix_x_bnd, self.fix_y_bnd), central_diff_2d(y, h, self.fix_x_bnd, self.fix_y_bnd)
        else:
            xs = central_diff_3d(x, h, self.fix_x_bnd, self.fix_y_bnd, self.fix_z_bnd)
            ys = central_diff_3d(y, h, self.fix_x_bnd, self.fix_y_bnd, self.fix_z_bnd)
        dx = {0: torch.flatten(x, -self.d)}
        dy = {0: torch.flatten(y, -self.d)}
        for j in range(self.d):
            dx[j + 1], dy[j + 1] = torch.flatten(xs[j], -self.d), torch.flatten(ys[j], -self.d)
        return dx, dy

    def abs(self, x, y, h=None):
        h = self.uniform_h(x) if h is None else ([h] * self.d if isinstance(h, float) else h)
        dict_x, dict_y = self.compute_terms(x, y, h)
        const = math.prod(h)
        diff = const * torch.linalg.vector_norm(dict_x[0] - dict_y[0], ord=2, dim=-1) ** 2
        for j in range(1, self.d + 1):
            diff = diff + const * torch.linalg.vector_norm(dict_x[j] - dict_y[j], ord=2, dim=-1) ** 2
        diff = diff ** 0.5
        if self.reduce_dims is not None:
            diff = self.reduce_all(diff).squeeze()
        return diff

    def rel(self, x, y, h=None):
        h = self.uniform_h(x) if h is None else ([h] * self.d if isinstance(h, float) else h)
        dict_x, dict_y = self.compute_terms(x, y, h)
        diff = torch.linalg.vector_norm(dict_x[0] - dict_y[0], ord=2, dim=-1) ** 2
        ynorm = torch.linalg.vector_norm(dict_y[0], ord=2, dim=-1) ** 2
        for j in range(1, self.d + 1):
            diff = diff + torch.linalg.vector_norm(dict_x[j] - dict_y[j], ord=2, dim=-1) ** 2
            ynorm = ynorm + torch.linalg.vector_norm(dict_y[j], ord=2, dim=-1) ** 2
        diff = (diff ** 0.5) / (ynorm ** 0.5)
        if self.reduce_dims is not None:
            diff = self.reduce_all(diff).squeeze()
        return diff

    def __call__(self, output_dict, label_dict, weight_dict: Optional[dict] = None, h=None):
        x, y = output_dict["y"], label_dict["y"]
        return {"h1": self.rel(x, y, h=h) / x.shape[0]}


class H1Loss_train(H1Loss):  # noqa: N801 -- the reference's name (metric.py:386-412)
    def __call__(self, output_dict, label_dict, weight_dict: Optional[dict] = None, h=None):
        return {"y": self.rel(output_dict["y"], label_dict["y"], h=h)}
