"""ppsci.equation.PDE (/root/reference/ppsci/equation/pde/base.py:31-243): a named collection of
residual definitions, each either a sympy expression or a Python callable on the data dict."""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple, Union

import sympy as sp

DETACH_FUNC_NAME = "detach"


class EqParamStore:
    """All learnable equation parameters of the process (`PDE.learnable_parameters`, base.py:38 of the reference) live
    in ONE small device vector of _lib.MAX_EPARAM slots next to their gradient: the epilogue kernel reads a slot
    with OP_LD_PARAM and sums its adjoint over the points (ParameterNode, symbolic.py:471-485)."""

    _inst = None

    def __init__(self):
        import torch

        from ... import _lib as L
        from ...device import get_device

        self.values = torch.zeros(L.MAX_EPARAM, dtype=torch.float32, device=get_device())
        self.grad = torch.zeros_like(self.values)
        self.used = 0

    @classmethod
    def get(cls) -> "EqParamStore":
        from ...device import get_device

        if cls._inst is None or cls._inst.values.device != get_device():
            cls._inst = cls()
        return cls._inst

    @classmethod
    def reset(cls) -> None:
        cls._inst = None


class EqParam:
    """One scalar learnable parameter (paddle.create_parameter(shape=[]) in the reference's equations)."""

    _count = 0

    def __init__(self, value: float, name: Optional[str] = None):
        from ... import _lib as L

        store = EqParamStore.get()
        if store.used >= L.MAX_EPARAM:
            raise NotImplementedError(f"more than {L.MAX_EPARAM} learnable equation parameters")
        self.store, self.slot = store, store.used
        store.used += 1
        store.values[self.slot] = float(value)
        self.name = name or f"eq_param_{EqParam._count}"  # paddle names them create_parameter_<n>.w_0
        EqParam._count += 1
        self.shape = []
        self.stop_gradient = False

    def item(self) -> float:
        return float(self.store.values[self.slot])

    __float__ = item

    def numpy(self):
        import numpy as np

        return np.asarray(self.item(), dtype=np.float32)

    def set_value(self, v) -> None:
        self.store.values[self.slot] = float(v)

    @property
    def grad(self) -> float:
        return float(self.store.grad[self.slot])


class PDE:
    def __init__(self):
        self.equations: Dict[str, Union[Callable, sp.Basic]] = {}
        self.learnable_parameters: List = []
        self.detach_keys: Optional[Tuple[str, ...]] = None

    @staticmethod
    def create_symbols(symbol_str: str):
        return sp.symbols(symbol_str)

    def create_function(self, name: str, invars: Tuple[sp.Symbol, ...]) -> sp.Function:
        return sp.Function(name)(*invars)

    def _apply_detach(self):
        """base.py:91-151: wrap every sub-expression whose key is in detach_keys into detach(.),
        never wrapping twice and never wrapping the differentiated function of a Derivative."""
        if self.detach_keys is None:
            return
        from sympy.core.traversal import postorder_traversal

        from ...utils.symbolic import _cvt_to_key

        det = sp.Function(DETACH_FUNC_NAME)
        for name, expr in list(self.equations.items()):
            if not isinstance(expr, sp.Basic):
                continue
            new = expr
            for item in postorder_traversal(expr):
                if _cvt_to_key(item) not in self.detach_keys:
                    continue
                new = new.replace(item, det(item))
                new = new.replace(det(det(item)), det(item))
                for sub in list(postorder_traversal(new)):
                    if isinstance(sub, sp.Derivative) and getattr(sub.args[0], "name", None) == DETACH_FUNC_NAME:
                        new = new.replace(sub, sp.Derivative(sub.args[0].args[0], *sub.args[1:]))
            self.equations[name] = new

    def add_equation(self, name: str, equation: Callable):
        self.equations.update({name: equation})

    def parameters(self) -> List:
        return list(self.learnable_parameters)

    def state_dict(self) -> Dict[str, object]:
        """nn.ParameterList.state_dict(): keys "0", "1", ... (base.py:192-208)."""
        return {str(i): p.numpy() if hasattr(p, "numpy") else p for i, p in enumerate(self.learnable_parameters)}

    def set_state_dict(self, state_dict):
        missing = [str(i) for i in range(len(self.learnable_parameters)) if str(i) not in state_dict]
        unexpected = [k for k in state_dict if not (k.isdigit() and int(k) < len(self.learnable_parameters))]
        for i, p in enumerate(self.learnable_parameters):
            if str(i) in state_dict:
                p.set_value(float(state_dict[str(i)]))
        return missing, unexpected

    def __str__(self):
        return "\n".join([self.__class__.__name__] + [f"    {name}: {eq}" for name, eq in self.equations.items()])
