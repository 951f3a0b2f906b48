"""ppsci.equation.PDE (/root/reference/ppsci/equation/pde/base.py:31-243): a named collection of
residual definitions, each either a sympy expression or a Python callable on the data dict."""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple, Union

import sympy as sp

DETACH_FUNC_NAME = "detach"


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
        return {str(i): p for i, p in enumerate(self.learnable_parameters)}

    def set_state_dict(self, state_dict):
        return [], []

    def __str__(self):
        return "\n".join([self.__class__.__name__] + [f"    {name}: {eq}" for name, eq in self.equations.items()])
