"""ppsci.lambdify (/root/reference/ppsci/utils/symbolic.py:681-981): sympy expression -> callable on
the data dict.

The node list is built the way the reference builds it -- post-order traversal (symbolic.py:507-534),
`subs(1.0, 1)` (:791), input symbols dropped (:799-803), duplicates removed (:806) -- and every node
caches its value into `data_dict` under the `_cvt_to_key` string ("u__x__x", :111-137), Add / Mul
being left folds over the children in sympy argument order (:225-235).  The values are traced
expressions (graph.Sym): the whole list is later lowered to one epilogue program, so "derivative
fusion" (symbolic.py:336-403, 631-678), an optimisation of the reference's reverse sweeps, has no
counterpart here and `fuse_derivative` is accepted and ignored."""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import sympy as sp
import torch

from ..autodiff import hessian, jacobian
from ..equation.pde.base import DETACH_FUNC_NAME
from ..graph import Sym, apply

DATA_DICT = Dict[str, object]

_SYMPY_UNARY = {
    sp.sin: "sin", sp.cos: "cos", sp.exp: "exp", sp.tanh: "tanh", sp.log: "log", sp.Abs: "abs",
    sp.sinh: "sinh", sp.cosh: "cosh", sp.tan: "tan", sp.sign: "sign",
    sp.asin: "asin", sp.acos: "acos", sp.atan: "atan", sp.asinh: "asinh", sp.acosh: "acosh", sp.atanh: "atanh",
    sp.erf: "erf", sp.loggamma: "lgamma", sp.ceiling: "ceil", sp.floor: "floor",
}


def _cvt_to_key(expr: sp.Basic) -> str:
    if isinstance(expr, sp.Function) and str(expr.func) == DETACH_FUNC_NAME:
        return f"{_cvt_to_key(expr.args[0])}_{DETACH_FUNC_NAME}"
    if isinstance(expr, (sp.Symbol, sp.core.function.UndefinedFunction, sp.Function)):
        return expr.name if hasattr(expr, "name") else str(expr)
    if isinstance(expr, sp.Derivative):
        s = expr.args[0].name
        for symbol, order in expr.args[1:]:
            s += f"__{symbol}" * order
        return s
    return str(expr)


def _post_traverse(cur: sp.Basic, nodes: List[sp.Basic]) -> List[sp.Basic]:
    if isinstance(cur, sp.Function):
        for arg in cur.args:
            nodes = _post_traverse(arg, nodes)
        nodes.append(cur)
    elif isinstance(cur, sp.Derivative):
        nodes = _post_traverse(cur.args[0], nodes)
        nodes.append(cur)
    elif isinstance(cur, (sp.Symbol, sp.Number)):
        nodes.append(cur)
    else:
        for arg in cur.args:
            nodes = _post_traverse(arg, nodes)
        nodes.append(cur)
    return nodes


class ComposedNode:
    """symbolic.py:488-504: runs the node list in order and returns the value of the root."""

    def __init__(self, sympy_nodes: List[sp.Basic], models: Tuple, parameters: Tuple = ()):
        assert len(sympy_nodes)
        self.sympy_nodes = sympy_nodes
        self.models = models
        self.parameters = tuple(parameters)
        self.keys = [_cvt_to_key(n) for n in sympy_nodes]

    def _eval(self, node: sp.Basic, data: DATA_DICT):
        if isinstance(node, sp.Derivative):  # DerivativeNode symbolic.py:310-333
            val = data[_cvt_to_key(node.args[0])]
            for sym, order in node.args[1:]:
                order = int(order)
                x = data[_cvt_to_key(sym)]
                if order & 1:
                    val = jacobian(val, x)
                    order -= 1
                for _ in range(0, order, 2):
                    val = hessian(val, x)
            return val
        if node.func == sp.Add:
            val = data[_cvt_to_key(node.args[0])]
            for a in node.args[1:]:
                val = val + data[_cvt_to_key(a)]
            return val
        if node.func == sp.Mul:
            val = data[_cvt_to_key(node.args[0])]
            for a in node.args[1:]:
                val = val * data[_cvt_to_key(a)]
            return val
        if node.func == sp.Pow:
            return _apply_any("pow", _as_sym(data[_cvt_to_key(node.args[0])]), _as_sym(data[_cvt_to_key(node.args[1])]))
        if node.func in (sp.Max, sp.Min):
            op = "max" if node.func == sp.Max else "min"
            val = _apply_any(op, _as_sym(data[_cvt_to_key(node.args[0])]), _as_sym(data[_cvt_to_key(node.args[1])]))
            for a in node.args[2:]:
                val = _apply_any(op, val, _as_sym(data[_cvt_to_key(a)]))
            return val
        if node.func == sp.atan2:
            return _apply_any("atan2", _as_sym(data[_cvt_to_key(node.args[0])]), _as_sym(data[_cvt_to_key(node.args[1])]))
        if node.func == sp.Heaviside:
            return _apply_any("heaviside", _as_sym(data[_cvt_to_key(node.args[0])]))
        if isinstance(node, sp.Function) and str(node.func) == DETACH_FUNC_NAME:  # DetachNode :165-181
            return _as_sym(data[_cvt_to_key(node.args[0])]).detach()
        if isinstance(node, sp.Function) and node.func in _SYMPY_UNARY:
            return _apply_any(_SYMPY_UNARY[node.func], _as_sym(data[_cvt_to_key(node.args[0])]))
        if node.is_Number or node.is_NumberSymbol:  # ConstantNode :433-468
            if not (node.is_Float or node.is_Integer or node.is_Boolean or node.is_Rational):
                raise TypeError(f"expr({node}) should be Float/Integer/Boolean/Rational, but got {type(node)}")
            if any(isinstance(v, torch.Tensor) for v in data.values()):  # eager: an fp32 scalar (ConstantNode)
                return float(np.float32(float(node)))
            return Sym.const(float(node))
        raise NotImplementedError(f"The node {node} is not supported in lambdify.")

    def __call__(self, data_dict: DATA_DICT):
        for node, key in zip(self.sympy_nodes, self.keys):
            if key in data_dict:  # cache hit (also how precomputed 'sdf__x' inputs are used, :314-317)
                continue
            if isinstance(node, sp.Function) and node.func not in _SYMPY_UNARY and node.func not in (sp.Heaviside, sp.atan2) \
                    and str(node.func) != DETACH_FUNC_NAME and not isinstance(node, (sp.Max, sp.Min)):
                # LayerNode symbolic.py:406-430
                hit = [m for m in self.models if str(node.func) in m.output_keys]
                if len(hit) > 1:
                    raise ValueError(f"Name of function: '{node}' should be unique along given models")
                if hit:
                    data_dict.update(hit[0](data_dict))
                elif str(node.func) != "sdf":
                    raise ValueError(f"Node {node} can not match any model in given model(s).")
                continue
            if isinstance(node, sp.Symbol):  # ParameterNode symbolic.py:471-485: one scalar, broadcast over the points
                hit = [p for p in self.parameters if p.name == node.name]
                if any(isinstance(v, torch.Tensor) for v in data_dict.values()):
                    raise NotImplementedError("learnable equation parameters on the eager fallback path")
                data_dict[key] = Sym.param(hit[0].name, hit[0].slot)
                continue
            data_dict[key] = self._eval(node, data_dict)
        return data_dict[self.keys[-1]]

    forward = __call__


def _as_sym(v):
    if isinstance(v, (Sym, torch.Tensor)):
        return v
    return Sym.const(float(v))


_TORCH_FUNCS = {"pow": torch.pow, "max": torch.maximum, "min": torch.minimum, "atan2": torch.atan2,
                "heaviside": lambda x: torch.heaviside(x, torch.zeros_like(x)), "sin": torch.sin, "cos": torch.cos,
                "tanh": torch.tanh, "exp": torch.exp, "log": torch.log, "sqrt": torch.sqrt, "abs": torch.abs, "sinh": torch.sinh,
                "cosh": torch.cosh, "tan": torch.tan, "sign": torch.sign, "asin": torch.asin, "acos": torch.acos,
                "atan": torch.atan, "asinh": torch.asinh, "acosh": torch.acosh, "atanh": torch.atanh, "erf": torch.erf,
                "lgamma": torch.lgamma, "ceil": torch.ceil, "floor": torch.floor, "neg": torch.neg}


def _apply_any(op: str, *args):
    """graph.apply on traced values; the same operator on real tensors (eager fallback): constants arrive as fp32
    scalars like the reference's ConstantNode."""
    if any(isinstance(a, torch.Tensor) for a in args):
        ref = next(a for a in args if isinstance(a, torch.Tensor))
        ts = [a if isinstance(a, torch.Tensor) else torch.full_like(ref, float(a.value if isinstance(a, Sym) else a))
              for a in args]
        return _TORCH_FUNCS[op](*ts)
    return apply(op, *args)


def lambdify(
    expr: Union[sp.Basic, List[sp.Basic]],
    models=None,
    extra_parameters: Optional[Sequence] = None,
    graph_filename: Optional[str] = None,
    create_graph: bool = True,
    retain_graph: Optional[bool] = None,
    fuse_derivative: bool = False,
) -> Union[ComposedNode, List[ComposedNode]]:
    extra_parameters = tuple(extra_parameters or ())
    for prm in extra_parameters:
        if not hasattr(prm, "slot"):
            raise TypeError("extra_parameters must be equation parameters (ppsci.equation.pde.base.EqParam)")
    parameter_names = tuple(prm.name for prm in extra_parameters)
    if models is not None and hasattr(models, "model_list"):
        models = tuple(models.model_list)
    if not isinstance(models, (tuple, list)):
        models = (models,)

    def convert(single: sp.Basic) -> ComposedNode:
        single = single.subs(1.0, 1)
        nodes = _post_traverse(single, [])
        nodes = [n for n in nodes if (not n.is_Symbol) or (_cvt_to_key(n) in parameter_names)]  # symbolic.py:797-803
        nodes = list(dict.fromkeys(nodes))
        return ComposedNode(nodes, tuple(models), extra_parameters)

    if isinstance(expr, sp.Basic):
        return convert(expr)
    return [convert(e) for e in expr]
