"""ppsci.utils.expression.ExpressionSolver (/root/reference/ppsci/utils/expression.py:60-222) on the fused HIP path.

The reference's three entry points re-execute model forward + expressions + loss op by op on every call.  Here a
call compiles its (expressions, input keys, batch size) once -- compile.CompiledConstraint: trace on proxies, lower
to a stream set + epilogue program -- and later calls with the same signature only bind the new batch and launch the
fixed kernel sequence.  Same arguments, same return structure:

  train_forward(expr_dicts, input_dicts, model, constraint, label_dicts, weight_dicts)
      -> (losses_all: {term: 0-d tensor}, losses_constraint: {constraint: float})       (:61-131)
     The reference returns tensors with an autograd graph and the caller runs `total.backward()`; here the forward has
     already written the adjoint seeds, and `backward()` runs the reverse sweeps and returns dL/dparams of the SUM of
     all terms (mtl.Sum, the reference default) as one flat tensor in `model.parameters()` order.
  eval_forward(expr_dict, input_dict, model, validator, label_dict, weight_dict)
      -> (output_dict, validator_loss: {term: 0-d tensor})                              (:133-190)
  visu_forward(expr_dict, input_dict, model) -> output_dict                             (:192-222)
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import numpy as np
import torch

from ..compile import CompiledConstraint
from ..device import get_device

__all__ = ["ExpressionSolver"]


def _n_rows(d: Dict[str, object]) -> int:
    v = next(iter(d.values()))
    return int(v.shape[0]) if hasattr(v, "shape") else len(v)


class ExpressionSolver:
    """Expression computing helper, which computes named results according to the corresponding function and inputs.

    Examples:
        >>> import ppsci
        >>> model = ppsci.arch.MLP(("x", "y"), ("u", "v"), 5, 128)
        >>> expr_solver = ppsci.utils.expression.ExpressionSolver()
    """

    nvtx_flag: bool = False  # accepted for signature compatibility (rocprofv3 needs no in-process markers)

    def __init__(self):
        self._cache: Dict[tuple, CompiledConstraint] = {}
        self._last_train: Tuple[object, list] = (None, [])

    def forward(self, *args, **kwargs):
        raise NotImplementedError("Use train_forward/eval_forward/visu_forward instead of forward.")

    __call__ = forward

    # ------------------------------------------------------------------ helpers
    def _compiled(self, tag, model, exprs, input_dict, label_dict, weight_dict, loss, train: bool, values: bool):
        n = _n_rows(input_dict)
        in_keys = tuple(input_dict.keys())
        lab_keys = tuple((label_dict or {}).keys())
        w_keys = tuple(k for k in (weight_dict or {}).keys() if k in lab_keys)
        key = (tag, id(model), tuple((k, id(f)) for k, f in exprs.items()), in_keys, lab_keys, w_keys, id(loss), n, train)
        cc = self._cache.get(key)
        if cc is None:
            extra = [k for k in exprs if k not in lab_keys] if values else []
            cc = CompiledConstraint(str(tag), model, dict(exprs), list(in_keys), list(lab_keys), list(w_keys), loss, n, n,
                                    get_device(), train=train, want_values=values, extra_outputs=extra)
            self._cache[key] = cc
        cc.bind(input_dict, label_dict or {}, weight_dict or {})
        return cc

    # ------------------------------------------------------------------ reference surface
    def train_forward(self, expr_dicts, input_dicts, model, constraint, label_dicts, weight_dicts):
        losses_all: Dict[str, torch.Tensor] = {}
        losses_constraint: Dict[str, float] = {}
        params = model.materialize() if hasattr(model, "materialize") else model.flat_params
        compiled = []
        for i, cst_name in enumerate(constraint):
            cst_obj = constraint[cst_name]
            cc = self._compiled(("train", cst_name), model, expr_dicts[i], input_dicts[i], label_dicts[i], weight_dicts[i],
                                cst_obj.loss, True, False)
            cc.fused.forward(params, True)
            compiled.append(cc)
            terms = cc.fused.loss_terms
            losses_constraint[cst_name] = 0.0
            vals = cc.fused.losses()
            for j, key in enumerate(cc.label_keys):
                losses_constraint[cst_name] += vals[key]
                losses_all[key] = losses_all[key] + terms[j] if key in losses_all else terms[j].clone()
        self._last_train = (model, compiled)
        return losses_all, losses_constraint

    def backward(self) -> torch.Tensor:
        """Gradient of the sum of all loss terms of the last train_forward w.r.t. the model's flat parameter vector."""
        model, compiled = self._last_train
        if model is None:
            raise RuntimeError("backward() follows a train_forward()")
        params = model.materialize() if hasattr(model, "materialize") else model.flat_params
        grad = torch.zeros_like(params)
        for i, cc in enumerate(compiled):
            cc.fused.backward(params)
            cc.fused.reduce_grads(grad, i > 0 or len(cc.fused.nets) > 1)
        # factored / tied layers (weight_norm, random_weight, fourier): kernel-layout gradient -> trainable tensors
        return model.pull_back(grad) if getattr(model, "reparam", False) else grad

    def eval_forward(self, expr_dict, input_dict, model, validator, label_dict, weight_dict):
        cc = self._compiled(("eval", getattr(validator, "name", id(validator))), model, expr_dict, input_dict, label_dict,
                            weight_dict, validator.loss, False, True)
        params = model.materialize() if hasattr(model, "materialize") else model.flat_params
        cc.fused.forward(params, False)
        vals = cc.values()
        output_dict = {k: vals[k].clone() for k in vals}
        terms = cc.fused.loss_terms
        return output_dict, {k: terms[j].clone() for j, k in enumerate(cc.label_keys)}

    def visu_forward(self, expr_dict: Optional[Dict[str, Callable]], input_dict, model):
        exprs = dict(expr_dict) if expr_dict is not None else {k: (lambda out, k=k: out[k]) for k in model.output_keys}
        cc = self._compiled(("visu",), model, exprs, input_dict, None, None, None, False, True)
        params = model.materialize() if hasattr(model, "materialize") else model.flat_params
        cc.fused.forward(params, False)
        vals = cc.values()
        return {k: vals[k].clone() for k in exprs}
