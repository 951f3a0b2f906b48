"""ppsci.optimizer.{Adam, AdamW, SGD, Momentum, RMSProp, LBFGS}
(/root/reference/ppsci/optimizer/optimizer.py:39-495).  Adam (:179-248): a factory called with the
model(s); the returned object owns the Adam moments and performs the fused HIP update on the model's
flat parameter buffer (paddle.optimizer.Adam semantics, beta1=0.9 beta2=0.999 epsilon=1e-8).
`weight_decay` (a float: paddle's L2Decay(coeff), added to the gradient) and `grad_clip` (the ClipGradBy* classes below,
stand-ins for paddle.nn.ClipGradBy*) are honoured; amsgrad / lazy_mode of the reference signature are rejected when set."""
from __future__ import annotations

from typing import Optional, Union

import torch

from .. import hotpath as hp
from . import lr_scheduler


def _put_scheduler(d: dict, lr) -> None:
    if hasattr(lr, "state_dict"):
        st = lr.state_dict()
        d["lr_last_epoch"], d["lr_last_lr"] = int(st["last_epoch"]), float(st["last_lr"])


def _get_scheduler(state: dict, lr) -> None:
    if hasattr(lr, "set_state_dict") and "lr_last_epoch" in state:
        lr.set_state_dict({"last_epoch": int(state["lr_last_epoch"]), "last_lr": float(state["lr_last_lr"])})


class ClipGradByValue:
    """paddle.nn.ClipGradByValue(max, min=None): g <- clip(g, min, max), min defaults to -max."""

    def __init__(self, max: float, min: Optional[float] = None):  # noqa: A002
        self.max, self.min = float(max), float(-max if min is None else min)

    def __call__(self, model, grad: torch.Tensor) -> torch.Tensor:
        return grad.clamp_(self.min, self.max)


class ClipGradByNorm:
    """paddle.nn.ClipGradByNorm(clip_norm): every parameter TENSOR's gradient is rescaled to at most clip_norm."""

    def __init__(self, clip_norm: float):
        self.clip_norm = float(clip_norm)

    def __call__(self, model, grad: torch.Tensor) -> torch.Tensor:
        base = model.flat_params.storage_offset()
        for p in model.parameters():  # views of the flat buffer: the gradient has the same layout
            o = p.storage_offset() - base
            g = grad[o:o + p.numel()]
            nrm = torch.linalg.vector_norm(g)
            g.mul_(torch.clamp(self.clip_norm / torch.clamp(nrm, min=1e-30), max=1.0))
        return grad


class ClipGradByGlobalNorm:
    """paddle.nn.ClipGradByGlobalNorm(clip_norm): g <- g * clip_norm / max(||g||_2 over all parameters, clip_norm)."""

    def __init__(self, clip_norm: float):
        self.clip_norm = float(clip_norm)

    def __call__(self, model, grad: torch.Tensor) -> torch.Tensor:
        nrm = torch.linalg.vector_norm(grad)
        return grad.mul_(self.clip_norm / torch.clamp(nrm, min=self.clip_norm))


def _clip(grad_clip, model, grad: torch.Tensor, grad_scale: float) -> torch.Tensor:
    """Clipping acts on the gradient the optimizer would see (after the data-parallel / accumulation scale)."""
    if grad_clip is None:
        return grad
    if grad_scale != 1.0:
        grad = grad * grad_scale
    else:
        grad = grad.clone()
    return grad_clip(model, grad)


class _AdamState:
    l2, grad_clip = 0.0, None

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
        if self.grad_clip is not None:
            grad, grad_scale = _clip(self.grad_clip, self.model, grad, grad_scale), 1.0
        if self.l2 != 0.0:
            # paddle Adam(weight_decay=c): L2Decay, g += c * p before the moments -- the fused kernel's l2 term with
            # the decoupled-decay factor of its AdamW branch set to 1
            c2 = (1.0 - self.beta2 ** self.t) ** 0.5
            lr_t = self.get_lr() * c2 / (1.0 - self.beta1 ** self.t)
            hp.optim_step(hp.OPT_ADAMW, self.model.flat_params, grad, [self.m, self.v],
                          [lr_t, grad_scale, self.l2, self.beta1, self.epsilon * c2, 1.0, self.beta2])
        else:
            hp.adam_step(self.model.flat_params, grad, self.m, self.v, self.get_lr(), self.t, self.beta1, self.beta2,
                         self.epsilon, grad_scale)

        if self.eq_store is not None:  # the learnable equation parameters: same rule, their own moments
            hp.adam_step(self.eq_store.values, self.eq_store.grad, self.eq_m, self.eq_v, self.get_lr(), self.t,
                         self.beta1, self.beta2, self.epsilon, grad_scale)

    eq_store = None

    def attach_equation_parameters(self):
        from ..equation.pde.base import EqParamStore

        self.eq_store = EqParamStore.get()
        self.eq_m = torch.zeros_like(self.eq_store.values)
        self.eq_v = torch.zeros_like(self.eq_store.values)

    def clear_grad(self):
        pass  # the flat gradient is overwritten by every reduce_rows

    def state_dict(self):
        """Everything a resumed run needs to continue bit-for-bit (paddle's optimizer.state_dict() carries the
        moments, the beta powers and the 'LR_Scheduler' entry): moments, step count, the moments of the learnable
        equation parameters and the scheduler's position."""
        d = {"m": self.m, "v": self.v, "t": self.t}
        if self.eq_store is not None:
            d["eq_m"], d["eq_v"] = self.eq_m, self.eq_v
        _put_scheduler(d, self._lr)
        return d

    def set_state_dict(self, state):
        self.m.copy_(torch.as_tensor(state["m"]).to(self.m.device))
        self.v.copy_(torch.as_tensor(state["v"]).to(self.v.device))
        self.t = int(state["t"])
        if self.eq_store is not None and "eq_m" in state:
            self.eq_m.copy_(torch.as_tensor(state["eq_m"]).to(self.eq_m.device))
            self.eq_v.copy_(torch.as_tensor(state["eq_v"]).to(self.eq_v.device))
        _get_scheduler(state, self._lr)


class Adam:
    def __init__(self, learning_rate: Union[float, "lr_scheduler._Scheduler"] = 1e-3, beta1: float = 0.9,
                 beta2: float = 0.999, epsilon: float = 1e-8, weight_decay=None, grad_clip=None, lazy_mode: bool = False,
                 amsgrad: bool = False):
        if lazy_mode or amsgrad:
            raise NotImplementedError("lazy_mode / amsgrad have no fused HIP kernel yet")
        self.learning_rate, self.beta1, self.beta2, self.epsilon = learning_rate, beta1, beta2, epsilon
        self.l2, self.grad_clip = _l2(weight_decay), grad_clip

    def __call__(self, model_list) -> _AdamState:
        """`Adam(lr)(model)` or, for inverse problems, `Adam(lr)((model,) + tuple(equation.values()))`
        (examples/fsi/viv.py:121): equations contribute their learnable parameters."""
        eqs = []
        if isinstance(model_list, (list, tuple)):
            eqs = [m for m in model_list if hasattr(m, "learnable_parameters") and hasattr(m, "equations")]
            nets = [m for m in model_list if m not in eqs]
            if len(nets) != 1:
                raise NotImplementedError("one network per optimizer on the fused HIP path")
            model_list = nets[0]
        st = _AdamState(model_list, self.learning_rate, self.beta1, self.beta2, self.epsilon)
        st.l2, st.grad_clip = self.l2, self.grad_clip
        if any(e.learnable_parameters for e in eqs):
            st.attach_equation_parameters()
        return st


def _single(model_list):
    if isinstance(model_list, (list, tuple)):
        if len(model_list) != 1:
            raise NotImplementedError("one model per optimizer on the fused HIP path")
        return model_list[0]
    return model_list


def _l2(weight_decay) -> float:
    if weight_decay is None:
        return 0.0
    if isinstance(weight_decay, (int, float)):
        return float(weight_decay)  # paddle: a float means L2Decay(coeff)
    raise NotImplementedError("regularizer objects (L1Decay / L2Decay instances) are not supported; pass a float")


class _FusedState:
    """Common part of the fused first-order optimizers: flat state tensors + one ppsci_optim_step per step."""

    kind = hp.OPT_SGD
    n_states = 0
    grad_clip = None

    def _pre(self, grad, grad_scale):
        if self.grad_clip is None:
            return grad, grad_scale
        return _clip(self.grad_clip, self.model, grad, grad_scale), 1.0

    def __init__(self, model, learning_rate):
        self.model = model
        self._lr = learning_rate
        self.t = 0
        self.states = [torch.zeros_like(model.flat_params) for _ in range(self.n_states)]
        self._parameter_list = model.parameters()

    def get_lr(self) -> float:
        return float(self._lr.get_lr()) if hasattr(self._lr, "get_lr") else float(self._lr)

    def set_lr(self, lr: float):
        self._lr = lr

    def clear_grad(self):
        pass

    def state_dict(self):
        d = {f"s{i}": t for i, t in enumerate(self.states)}
        d["t"] = self.t
        # checkpoint files use the Adam field names (utils/save_load.py)
        d["m"] = self.states[0] if self.states else torch.zeros(1)
        d["v"] = self.states[1] if len(self.states) > 1 else torch.zeros(1)
        _put_scheduler(d, self._lr)
        return d

    def set_state_dict(self, state):
        for i, t in enumerate(self.states):
            key = f"s{i}" if f"s{i}" in state else ("m", "v")[i] if i < 2 else None
            if key is not None and key in state:
                t.copy_(torch.as_tensor(state[key]).to(t.device))
        self.t = int(state.get("t", 0))
        _get_scheduler(state, self._lr)


class _SGDState(_FusedState):
    def __init__(self, model, lr, l2):
        super().__init__(model, lr)
        self.l2 = l2

    def step(self, grad, grad_scale: float = 1.0):
        self.t += 1
        grad, grad_scale = self._pre(grad, grad_scale)
        hp.optim_step(hp.OPT_SGD, self.model.flat_params, grad, [], [self.get_lr(), grad_scale, self.l2])


class _MomentumState(_FusedState):
    n_states = 1

    def __init__(self, model, lr, momentum, l2, nesterov):
        super().__init__(model, lr)
        self.momentum, self.l2, self.nesterov = momentum, l2, nesterov

    def step(self, grad, grad_scale: float = 1.0):
        self.t += 1
        grad, grad_scale = self._pre(grad, grad_scale)
        hp.optim_step(hp.OPT_MOMENTUM, self.model.flat_params, grad, self.states,
                      [self.get_lr(), grad_scale, self.l2, self.momentum], self.nesterov)


class _RMSPropState(_FusedState):
    n_states = 3

    def __init__(self, model, lr, rho, epsilon, momentum, l2, centered):
        super().__init__(model, lr)
        self.rho, self.epsilon, self.momentum, self.l2, self.centered = rho, epsilon, momentum, l2, centered

    def step(self, grad, grad_scale: float = 1.0):
        self.t += 1
        grad, grad_scale = self._pre(grad, grad_scale)
        hp.optim_step(hp.OPT_RMSPROP, self.model.flat_params, grad, self.states,
                      [self.get_lr(), grad_scale, self.l2, self.rho, self.epsilon, self.momentum], self.centered)


class _AdamWState(_FusedState):
    n_states = 2

    def __init__(self, model, lr, beta1, beta2, epsilon, weight_decay):
        super().__init__(model, lr)
        self.beta1, self.beta2, self.epsilon, self.weight_decay = beta1, beta2, epsilon, weight_decay

    @property
    def m(self):
        return self.states[0]

    @property
    def v(self):
        return self.states[1]

    def step(self, grad, grad_scale: float = 1.0):
        self.t += 1
        grad, grad_scale = self._pre(grad, grad_scale)
        lr = self.get_lr()
        c2 = (1.0 - self.beta2 ** self.t) ** 0.5
        lr_t = lr * c2 / (1.0 - self.beta1 ** self.t)
        hp.optim_step(hp.OPT_ADAMW, self.model.flat_params, grad, self.states,
                      [lr_t, grad_scale, 0.0, self.beta1, self.epsilon * c2, 1.0 - lr * self.weight_decay, self.beta2])


class SGD:
    def __init__(self, learning_rate=0.001, weight_decay=None, grad_clip=None):
        self.learning_rate, self.l2, self.grad_clip = learning_rate, _l2(weight_decay), grad_clip

    def __call__(self, model_list):
        st = _SGDState(_single(model_list), self.learning_rate, self.l2)
        st.grad_clip = self.grad_clip
        return st


class Momentum:
    def __init__(self, learning_rate, momentum: float, weight_decay=None, grad_clip=None, use_nesterov: bool = False,
                 no_weight_decay_name: Optional[str] = None):
        if no_weight_decay_name:
            raise NotImplementedError("no_weight_decay_name (per-parameter decay masks) has no fused HIP kernel yet")
        self.learning_rate, self.momentum, self.l2, self.nesterov = learning_rate, momentum, _l2(weight_decay), use_nesterov
        self.grad_clip = grad_clip

    def __call__(self, model_list):
        st = _MomentumState(_single(model_list), self.learning_rate, self.momentum, self.l2, self.nesterov)
        st.grad_clip = self.grad_clip
        return st


class RMSProp:
    def __init__(self, learning_rate, rho: float = 0.95, epsilon: float = 1e-6, momentum: float = 0.0, weight_decay=None,
                 grad_clip=None, centered: bool = False):
        self.args = (learning_rate, rho, epsilon, momentum, _l2(weight_decay), centered)
        self.grad_clip = grad_clip

    def __call__(self, model_list):
        st = _RMSPropState(_single(model_list), *self.args)
        st.grad_clip = self.grad_clip
        return st


class AdamW:
    def __init__(self, learning_rate=0.001, beta1: float = 0.9, beta2: float = 0.999, epsilon: float = 1e-8,
                 weight_decay: float = 0.001, grad_clip=None, no_weight_decay_name: Optional[str] = None,
                 one_dim_param_no_weight_decay: bool = False, amsgrad: bool = False):
        if no_weight_decay_name or one_dim_param_no_weight_decay or amsgrad:
            raise NotImplementedError("per-parameter weight-decay masks / amsgrad have no fused HIP kernel yet")
        self.args = (learning_rate, beta1, beta2, epsilon, weight_decay)
        self.grad_clip = grad_clip

    def __call__(self, model_list):
        st = _AdamWState(_single(model_list), *self.args)
        st.grad_clip = self.grad_clip
        return st


class _LBFGSState:
    """paddle.optimizer.LBFGS as wrapped by optimizer.py:251-323: the two-loop recursion and the strong-Wolfe line
    search run on the flat parameter vector (torch.optim.LBFGS, the same minFunc port); every closure evaluation is
    one fused forward + reverse pass of the HIP engine (+ the all-reduce)."""

    is_lbfgs = True

    def __init__(self, model, lr, max_iter, max_eval, tolerance_grad, tolerance_change, history_size, line_search_fn):
        self.model = model
        self._lr = lr
        self.t = 0
        self._p = torch.nn.Parameter(model.flat_params, requires_grad=True)  # shares storage with flat_params
        self._opt = torch.optim.LBFGS([self._p], lr=lr, max_iter=max_iter, max_eval=max_eval,
                                      tolerance_grad=tolerance_grad, tolerance_change=tolerance_change,
                                      history_size=history_size, line_search_fn=line_search_fn)
        self._parameter_list = model.parameters()

    def get_lr(self) -> float:
        return float(self._lr)

    def step(self, closure):
        """closure() -> (loss: float, grad: flat tensor)."""
        self.t += 1

        def _c():
            loss, grad = closure()
            self._p.grad = grad.detach().clone()
            return torch.as_tensor(float(loss), dtype=torch.float32, device=self._p.device)

        return self._opt.step(_c)

    def clear_grad(self):
        self._p.grad = None

    def state_dict(self):
        return {"m": torch.zeros(1), "v": torch.zeros(1), "t": self.t}

    def set_state_dict(self, state):
        self.t = int(state.get("t", 0))  # the curvature history is rebuilt (paddle's LBFGS state is not portable either)


class LBFGS:
    def __init__(self, learning_rate: float = 1.0, max_iter: int = 1, max_eval: Optional[int] = None,
                 tolerance_grad: float = 1e-07, tolerance_change: float = 1e-09, history_size: int = 100,
                 line_search_fn: Optional[str] = "strong_wolfe"):
        self.args = (learning_rate, max_iter, max_eval, tolerance_grad, tolerance_change, history_size, line_search_fn)

    def __call__(self, model_list):
        return _LBFGSState(_single(model_list), *self.args)


class OptimizerList:
    """optimizer.py:498-559: several optimizers, one per model.  On the flat-buffer path the models are the members of
    ONE ppsci.arch.ModelList (whose parameters live in one buffer): `step(grad)` hands every optimizer its member's
    slice of the flat gradient.  LBFGS is refused like in the reference."""

    def __init__(self, optimizer_list):
        self._opt_list = tuple(optimizer_list)
        if any(getattr(o, "is_lbfgs", False) for o in self._opt_list):
            raise ValueError("LBFGS is not supported in OptimizerList yet.")

    def _slice(self, opt, grad: torch.Tensor) -> torch.Tensor:
        m = opt.model
        n = m.flat_params.numel()
        if hasattr(m, "_train_offset"):
            return grad[m._train_offset:m._train_offset + n]
        if len(self._opt_list) == 1 and grad.numel() == n:
            return grad
        raise NotImplementedError("OptimizerList: the optimizers' models must be members of one ModelList")

    def step(self, grad: torch.Tensor, grad_scale: float = 1.0):
        for opt in self._opt_list:
            opt.step(self._slice(opt, grad), grad_scale)

    def clear_grad(self):
        for opt in self._opt_list:
            opt.clear_grad()

    def get_lr(self) -> float:
        """Return learning rate of first optimizer"""
        return self._opt_list[0].get_lr()

    @property
    def t(self) -> int:
        return self._opt_list[0].t

    def set_state_dict(self, state_dicts):
        if isinstance(state_dicts, dict):  # checkpoint form: keys prefixed "opt<i>."
            state_dicts = [{k.split(".", 1)[1]: v for k, v in state_dicts.items() if k.startswith(f"opt{i}.")}
                           for i in range(len(self._opt_list))]
        for i, opt in enumerate(self._opt_list):
            opt.set_state_dict(state_dicts[i])

    def state_dict(self):
        return [opt.state_dict() for opt in self._opt_list]

    def __len__(self) -> int:
        return len(self._opt_list)

    def __getitem__(self, idx):
        return self._opt_list[idx]

    def __setitem__(self, idx, opt):
        raise NotImplementedError("Can not modify any item in OptimizerList.")

    def __iter__(self):
        yield from iter(self._opt_list)
