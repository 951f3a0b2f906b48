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


class Linear(LRBase):  # lr_scheduler.py:140-209 -> paddle PolynomialDecay
    def __init__(self, epochs, iters_per_epoch, learning_rate, end_lr=0.0, power=1.0, cycle=False, warmup_epoch=0,
                 warmup_start_lr=0.0, last_epoch=-1, by_epoch=False):
        super().__init__(epochs, iters_per_epoch, learning_rate, warmup_epoch, warmup_start_lr, last_epoch, by_epoch)
        self.decay_steps = (epochs - self.warmup_epoch) * iters_per_epoch
        self.end_lr, self.power, self.cycle = end_lr, power, cycle
        self.warmup_steps = round(self.warmup_epoch * iters_per_epoch)
        if self.by_epoch:
            self.decay_steps = self.epochs - self.warmup_epoch

    def __call__(self):
        if self.decay_steps <= 0:
            return self._build(lambda t: self.learning_rate)

        def fn(t):
            n, steps = t, self.decay_steps
            if self.cycle:
                div = math.ceil(float(t) / float(self.decay_steps)) if t != 0 else 1
                steps = self.decay_steps * div
            else:
                n = min(t, self.decay_steps)
            return (self.learning_rate - self.end_lr) * ((1 - float(n) / float(steps)) ** self.power) + self.end_lr

        return self._build(fn)


class MultiStepDecay(LRBase):  # lr_scheduler.py:461-520
    def __init__(self, epochs, iters_per_epoch, learning_rate, milestones: Tuple[int, ...], gamma: float = 0.1,
                 warmup_epoch=0, warmup_start_lr=0.0, last_epoch=-1, by_epoch=False):
        super().__init__(epochs, iters_per_epoch, learning_rate, warmup_epoch, warmup_start_lr, last_epoch, by_epoch)
        self.milestones = list(milestones) if by_epoch else [x * iters_per_epoch for x in milestones]
        self.gamma = gamma

    def __call__(self):
        def fn(t):
            for i, ms in enumerate(self.milestones):
                if t < ms:
                    return self.learning_rate * (self.gamma ** i)
            return self.learning_rate * (self.gamma ** len(self.milestones))

        return self._build(fn)


class CosineWarmRestarts(LRBase):  # lr_scheduler.py:523-658 (CosineAnnealingWarmRestarts, stepped sequentially)
    def __init__(self, epochs, iters_per_epoch, learning_rate, T_0: int, T_mult: int, eta_min: float = 0.0,
                 warmup_epoch=0, warmup_start_lr=0.0, last_epoch=-1, by_epoch=False):
        super().__init__(epochs, iters_per_epoch, learning_rate, warmup_epoch, warmup_start_lr, last_epoch, by_epoch)
        if T_0 <= 0 or not isinstance(T_0, int):
            raise ValueError(f"Expected positive integer T_0, but got {T_0}")
        if T_mult < 1 or not isinstance(T_mult, int):
            raise ValueError(f"Expected integer T_mult >= 1, but got {T_mult}")
        self.T_0 = T_0 if by_epoch else T_0 * iters_per_epoch
        self.T_mult, self.eta_min = T_mult, eta_min

    def __call__(self):
        def fn(t):
            t_cur, t_i = t, self.T_0  # position inside the current cycle
            while t_cur >= t_i:
                t_cur -= t_i
                t_i *= self.T_mult
            return self.eta_min + (self.learning_rate - self.eta_min) * (1 + math.cos(math.pi * t_cur / t_i)) / 2

        return self._build(fn)


class OneCycleLR(LRBase):  # lr_scheduler.py:661-741 -> paddle.optimizer.lr.OneCycleLR
    def __init__(self, epochs, iters_per_epoch, max_learning_rate, divide_factor=25.0, end_learning_rate=0.0001,
                 phase_pct=0.3, anneal_strategy="cos", three_phase=False, warmup_epoch=0, warmup_start_lr=0.0,
                 last_epoch=-1, by_epoch=False):
        super().__init__(epochs, iters_per_epoch, max_learning_rate, warmup_epoch, warmup_start_lr, last_epoch, by_epoch)
        self.total_steps = epochs if by_epoch else epochs * iters_per_epoch
        self.divide_factor, self.end_learning_rate, self.phase_pct = divide_factor, end_learning_rate, phase_pct
        if anneal_strategy not in ("cos", "linear"):
            raise ValueError(f"'anneal_strategy' must by one of 'cos' or 'linear', but received {anneal_strategy}")
        self.anneal_strategy, self.three_phase = anneal_strategy, three_phase

    def __call__(self):
        max_lr, total = self.learning_rate, self.total_steps
        initial, min_lr = max_lr / float(self.divide_factor), float(self.end_learning_rate)
        if self.three_phase:
            if self.phase_pct >= 0.5:
                raise ValueError("When three_phase is True, 'phase_pct' must be less than 0.5")
            cfg = [0, self.phase_pct * total - 1, 2 * self.phase_pct * total - 2, total - 1, total - 1]
            lrs = [initial, max_lr, initial, min_lr]
        else:
            cfg = [0, self.phase_pct * total - 1, total - 1, total - 1]
            lrs = [initial, max_lr, min_lr]
        sizes = [cfg[i + 1] - cfg[i] for i in range(len(cfg) - 1)]

        def anneal(a, b, pct):
            if self.anneal_strategy == "cos":
                return b + (a - b) / 2.0 * (math.cos(math.pi * pct) + 1)
            return (b - a) * pct + a

        def fn(t):
            for i, (end_step, size) in enumerate(zip(cfg[1:], sizes)):
                if t <= end_step or i == len(lrs) - 2:
                    return anneal(lrs[i], lrs[i + 1], (t - cfg[i]) / size)
            return min_lr

        return self._build(fn)


class LambdaDecay(LRBase):  # lr_scheduler.py:744-804
    def __init__(self, epochs, iters_per_epoch, learning_rate, lr_lambda: Callable, warmup_epoch=0, warmup_start_lr=0.0,
                 last_epoch=-1, by_epoch=False, verbose=False):
        super().__init__(epochs, iters_per_epoch, learning_rate, warmup_epoch, warmup_start_lr, last_epoch, by_epoch, verbose)
        self.lr_lambda = lr_lambda

    def __call__(self):
        return self._build(lambda t: self.learning_rate * self.lr_lambda(t))


class SchedulerList:  # lr_scheduler.py:807-843
    def __init__(self, scheduler_list):
        self._sch_list = tuple(scheduler_list)
        self.by_epoch = False

    def step(self):
        for sch in self._sch_list:
            sch.step()

    def get_lr(self):
        return self._sch_list[0].get_lr()

    def _set_by_epoch(self, by_epoch: bool):
        self.by_epoch = by_epoch
        for sch in self._sch_list:
            sch.by_epoch = by_epoch


__all__ = ["Constant", "ConstLR", "ExponentialDecay", "Cosine", "Step", "Piecewise", "Linear", "MultiStepDecay",
           "CosineWarmRestarts", "OneCycleLR", "LambdaDecay", "SchedulerList"]
