"""ppsci.optimizer.Adam (/root/reference/ppsci/optimizer/optimizer.py:179-248): a factory called with the
model(s); the returned object owns the Adam moments and performs the fused HIP update on the model's
flat parameter buffer (paddle.optimizer.Adam semantics, beta1=0.9 beta2=0.999 epsilon=1e-8).
weight_decay / grad_clip / amsgrad / lazy_mode of the reference signature are rejected when set."""
from __future__ import annotations

from typing import Optional, Union

import torch

from .. import hotpath as hp
from . import lr_scheduler


class _AdamState:
    def __init__(self, model, learning_rate, beta1, beta2, epsilon):
        self.model = model
        self._lr = learning_rate
        self.beta1, self.beta2, self.epsilon = beta1, beta2, epsilon
        p = model.flat_params
        self.m = torch.zeros_like(p)
        self.v = torch.zeros_like(p)
        self.t = 0
        self._parameter_list = model.parameters()

    def get_lr(self) -> float:
        return float(self._lr.get_lr()) if hasattr(self._lr, "get_lr") else float(self._lr)

    def set_lr(self, lr: float):
        self._lr = lr

    def step(self, grad: torch.Tensor, grad_scale: float = 1.0):
        self.t += 1
        hp.adam_step(self.model.flat_params, grad, self.m, self.v, self.get_lr(), self.t, self.beta1, self.beta2,
                     self.epsilon, grad_scale)

    def clear_grad(self):
        pass  # the flat gradient is overwritten by every reduce_rows

    def state_dict(self):
        return {"m": self.m, "v": self.v, "t": self.t}

    def set_state_dict(self, state):
        self.m.copy_(torch.as_tensor(state["m"]))
        self.v.copy_(torch.as_tensor(state["v"]))
        self.t = int(state["t"])


class Adam:
    def __init__(self, learning_rate: Union[float, "lr_scheduler._Scheduler"] = 1e-3, beta1: float = 0.9,
                 beta2: float = 0.999, epsilon: float = 1e-8, weight_decay=None, grad_clip=None, lazy_mode: bool = False,
                 amsgrad: bool = False):
        if weight_decay is not None or grad_clip is not None or lazy_mode or amsgrad:
            raise NotImplementedError("weight_decay / grad_clip / lazy_mode / amsgrad have no fused HIP kernel yet")
        self.learning_rate, self.beta1, self.beta2, self.epsilon = learning_rate, beta1, beta2, epsilon

    def __call__(self, model_list) -> _AdamState:
        if isinstance(model_list, (list, tuple)):
            if len(model_list) != 1:
                raise NotImplementedError("one model per optimizer on the fused HIP path")
            model_list = model_list[0]
        return _AdamState(model_list, self.learning_rate, self.beta1, self.beta2, self.epsilon)
