"""Learning-rate schedules (/root/reference/ppsci/optimizer/lr_scheduler.py): factories that return a
scheduler with paddle.optimizer.lr.LRScheduler's surface (`step()`, `get_lr()`/`()`, `last_epoch`,
`by_epoch`).  Closed forms of paddle's ExponentialDecay / CosineAnnealingDecay / PiecewiseDecay /
StepDecay with LinearWarmup in front (lr_scheduler.py:98-119)."""
from __future__ import annotations

import math
from typing import Callable, Sequence, Tuple, Union

from ..utils import logger


class _Scheduler:
    by_epoch = False

    def __init__(self, fn: Callable[[int], float], last_epoch: int = -1):
        self._fn = fn
        self.last_epoch = last_epoch
        self.step()  # paddle's LRScheduler.__init__ performs the first step

    def step(self, epoch: int = None):
        self.last_epoch = self.last_epoch + 1 if epoch is None else epoch
        self.last_lr = float(self._fn(self.last_epoch))

    def get_lr(self) -> float:
        return self.last_lr

    __call__ = get_lr

    def state_dict(self):
        return {"last_epoch": self.last_epoch, "last_lr": self.last_lr}

    def set_state_dict(self, state):
        self.last_epoch = state["last_epoch"]
        self.last_lr = state["last_lr"]


class LRBase:
    def __init__(self, epochs: int, iters_per_epoch: int, learning_rate: float, warmup_epoch: int,
                 warmup_start_lr: float, last_epoch: int, by_epoch: bool, verbose: bool = False) -> None:
        if warmup_epoch >= epochs:
            logger.warning("When using warm up, the value of 'Global.epochs' should be greater than value of "
                           f"'Optimizer.lr.warmup_epoch'. The value of 'Optimizer.lr.warmup_epoch' has been set to {epochs}.")
            warmup_epoch = epochs
        self.epochs, self.iters_per_epoch, self.learning_rate = epochs, iters_per_epoch, learning_rate
        self.warmup_epoch = warmup_epoch
        self.warmup_steps = self.warmup_epoch if by_epoch else round(self.warmup_epoch * self.iters_per_epoch)
        self.warmup_start_lr, self.last_epoch, self.by_epoch, self.verbose = warmup_start_lr, last_epoch, by_epoch, verbose

    def _build(self, inner: Callable[[int], float]) -> _Scheduler:
        ws, s0, s1 = self.warmup_steps, self.warmup_start_lr, self.learning_rate

        def fn(t: int) -> float:
            if ws > 0:  # paddle LinearWarmup: the wrapped schedule starts counting after the warm-up
                if t < ws:
                    return (s1 - s0) * float(t) / float(ws) + s0
                return inner(t - ws)
            return inner(t)

        sch = _Scheduler(fn, self.last_epoch)
        sch.by_epoch = self.by_epoch
        return sch


class Constant(_Scheduler):
    def __init__(self, learning_rate: float, last_epoch: int = -1):
        self.learning_rate = learning_rate
        super().__init__(lambda t: learning_rate, last_epoch)


class ConstLR(LRBase):
    def __init__(self, epochs, iters_per_epoch, learning_rate, warmup_epoch=0, warmup_start_lr=0.0, last_epoch=-1,
                 by_epoch=False):
        super().__init__(epochs, iters_per_epoch, learning_rate, warmup_epoch, warmup_start_lr, last_epoch, by_epoch)

    def __call__(self):
        return self._build(lambda t: self.learning_rate)


class ExponentialDecay(LRBase):  # lr_scheduler.py:212-269
    def __init__(self, epochs, iters_per_epoch, learning_rate, gamma, decay_steps, warmup_epoch=0, warmup_start_lr=0.0,
                 last_epoch=-1, by_epoch=False):
        super().__init__(epochs, iters_per_epoch, learning_rate, warmup_epoch, warmup_start_lr, last_epoch, by_epoch)
        self.decay_steps, self.gamma = decay_steps, gamma
        self.warmup_steps = round(self.warmup_epoch * iters_per_epoch)
        if self.by_epoch:
            self.decay_steps /= iters_per_epoch

    def __call__(self):
        g = self.gamma ** (1 / self.decay_steps)
        return self._build(lambda t: self.learning_rate * (g ** t))


class Cosine(LRBase):  # lr_scheduler.py:272-334
    def __init__(self, epochs, iters_per_epoch, learning_rate, eta_min=0.0, warmup_epoch=0, warmup_start_lr=0.0,
                 last_epoch=-1, by_epoch=False):
        super().__init__(epochs, iters_per_epoch, learning_rate, warmup_epoch, warmup_start_lr, last_epoch, by_epoch)
        self.T_max = (self.epochs - self.warmup_epoch) * self.iters_per_epoch
        self.eta_min = eta_min
        if self.by_epoch:
            self.T_max = self.epochs - self.warmup_epoch

    def __call__(self):
        if self.T_max > 0:
            return self._build(lambda t: self.eta_min + (self.learning_rate - self.eta_min)
                               * (1 + math.cos(math.pi * t / self.T_max)) / 2)
        return self._build(lambda t: self.learning_rate)


class Step(LRBase):
    def __init__(self, epochs, iters_per_epoch, learning_rate, step_size, gamma, warmup_epoch=0, warmup_start_lr=0.0,
                 last_epoch=-1, by_epoch=False):
        super().__init__(epochs, iters_per_epoch, learning_rate, warmup_epoch, warmup_start_lr, last_epoch, by_epoch)
        self.step_size = step_size if by_epoch else step_size * iters_per_epoch
        self.gamma = gamma

    def __call__(self):
        return self._build(lambda t: self.learning_rate * (self.gamma ** (t // self.step_size)))


class Piecewise(LRBase):
    def __init__(self, epochs, iters_per_epoch, decay_epochs: Tuple[int, ...], values: Tuple[float, ...], warmup_epoch=0,
                 warmup_start_lr=0.0, last_epoch=-1, by_epoch=False):
        super().__init__(epochs, iters_per_epoch, values[0], warmup_epoch, warmup_start_lr, last_epoch, by_epoch)
        self.values = values
        self.boundaries = list(decay_epochs) if by_epoch else [e * iters_per_epoch for e in decay_epochs]

    def __call__(self):
        def fn(t):
            for i, b in enumerate(self.boundaries):
                if t < b:
                    return self.values[i]
            return self.values[len(self.values) - 1]

        return self._build(fn)


__all__ = ["Constant", "ConstLR", "ExponentialDecay", "Cosine", "Step", "Piecewise"]
