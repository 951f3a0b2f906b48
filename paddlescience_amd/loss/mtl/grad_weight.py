"""ppsci.loss.mtl.GradNorm / NTK (/root/reference/ppsci/loss/mtl/grad_norm.py:28-145, ntk.py:27-86): loss weights
from the gradient norm of every loss term, refreshed every `update_freq` steps.

The reference calls `loss_i.backward(retain_graph=True)` per term inside the aggregator.  On the fused HIP path the
Solver does the same thing with one masked forward + reverse pass of the engine per loss key (the residual scales
of the other keys set to 0) and hands the norms to `update`; the current weights are applied as multipliers of the
residual scales, so the weighted total loss and its gradient come out of the ordinary fused step."""
from typing import Dict, List, Optional, Sequence

import numpy as np

from .base import LossAggregator


class _GradWeighted(LossAggregator):
    should_persist = True
    per_loss_grad = True  # Solver: supply per-key gradient norms at update steps

    def __init__(self, model, num_losses: int, update_freq: int, init_weights: Optional[Sequence[float]] = None) -> None:
        super().__init__(model)
        if init_weights is not None and num_losses != len(init_weights):
            raise ValueError(f"Length of init_weights({len(init_weights)}) should be equal to num_losses({num_losses}).")
        self.num_losses, self.update_freq = num_losses, update_freq
        self.weight = np.asarray(init_weights if init_weights is not None else np.ones(num_losses), dtype=np.float32)

    def __call__(self, losses: Dict[str, float], step: int = 0):
        assert len(losses) == self.num_losses, (
            f"Length of given losses({len(losses)}) should be equal to num_losses({self.num_losses}).")
        self.step = step
        total = 0.0
        for i, key in enumerate(losses):
            total = self.weight[i] * losses[key] if i == 0 else total + self.weight[i] * losses[key]
        return total

    def needs_update(self, step: int) -> bool:
        return step % self.update_freq == 0

    def update(self, grad_norms: List[float]) -> None:
        raise NotImplementedError

    def state_dict(self):
        return {"weight": self.weight.copy()}

    def set_state_dict(self, state):
        self.weight = np.asarray(state["weight"], dtype=np.float32)


class GradNorm(_GradWeighted):
    def __init__(self, model, num_losses: int = 1, update_freq: int = 1000, momentum: float = 0.9,
                 init_weights: Optional[List[float]] = None) -> None:
        super().__init__(model, num_losses, update_freq, init_weights)
        self.momentum = momentum

    def update(self, grad_norms: List[float]) -> None:
        g = np.asarray(grad_norms, dtype=np.float32)
        w = g.mean() / g
        self.weight = (self.momentum * self.weight + (1 - self.momentum) * w).astype(np.float32)


class NTK(_GradWeighted):
    def __init__(self, model, num_losses: int = 1, update_freq: int = 1000) -> None:
        super().__init__(model, num_losses, update_freq)

    def update(self, grad_norms: List[float]) -> None:
        g = np.asarray(grad_norms, dtype=np.float32)
        self.weight = (g.sum() / g).astype(np.float32)
