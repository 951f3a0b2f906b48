"""Constraint / validator compilation: trace the user's `output_expr` once on proxy tensors, lower it
(graph.lower) and bind the device buffers (engine.FusedConstraint).

This is the counterpart of `Solver.__init__`'s `convert_expr` (/root/reference/ppsci/solver/solver.py:
496-535) + the per-iteration body of `ExpressionSolver.train_forward` (ppsci/utils/expression.py:89-126):
what the reference re-executes op by op every iteration is decided here once."""
from __future__ import annotations

import zlib
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import sympy as sp
import torch

from . import autodiff, graph
from .engine import FusedConstraint
from .graph import Sym

LABEL_PREFIX, WEIGHT_PREFIX, CAUSAL_PREFIX = "label:", "weight:", "causal:"


def trace_exprs(model, input_keys: Sequence[str], exprs: Dict[str, Callable],
                extra_parameters: Sequence = (), batch: Optional[Dict[str, object]] = None,
                concretized: Optional[List[str]] = None) -> Dict[str, Sym]:
    """expression.py:96-102 on proxies: model forward, then every named expression on the data dict.  `batch`: the input
    columns of the ONE batch this trace will ever run on (a static constraint) -- Python control flow on their values is
    then followed (graph.batch_values); what was asked is appended to `concretized`."""
    with graph.batch_values(batch) as st:
        out = _trace_exprs(model, input_keys, exprs, extra_parameters)
        if concretized is not None:
            concretized.extend(st.concretized)
    return out


def _trace_exprs(model, input_keys, exprs, extra_parameters) -> Dict[str, Sym]:
    data: Dict[str, object] = {}
    for k in input_keys:
        data[k] = Sym.input(k) if k in model.input_keys else Sym.aux(k)
    members = getattr(model, "model_list", [model])
    transformed = any(getattr(m, "_input_transform", None) is not None for m in members)
    missing = [] if transformed else [k for k in model.input_keys if k not in data]  # (features come from the transform)
    if missing:
        raise KeyError(f"model input(s) {missing} are not provided by the dataset (has {list(input_keys)})")
    output_dict = model(data)
    data.update(output_dict)
    out: Dict[str, Sym] = {}
    for name, ex in exprs.items():
        if isinstance(ex, sp.Basic):
            from .utils.symbolic import lambdify

            ex = lambdify(ex, model, extra_parameters)
        val = ex(data)
        if not isinstance(val, Sym):
            val = graph._lift(val)
        out[name] = val
    autodiff.clear()  # expression.py:109
    return out


def _to_dev(a, dev) -> torch.Tensor:
    if isinstance(a, torch.Tensor):
        return a.to(device=dev, dtype=torch.float32).contiguous().view(-1)
    return torch.as_tensor(np.ascontiguousarray(np.asarray(a, dtype=np.float32))).to(dev).contiguous().view(-1)


class CompiledConstraint:
    """A constraint (or validator) bound to the fused kernels for a fixed batch size."""

    def __init__(self, name: str, model, exprs: Dict[str, Callable], input_keys: Sequence[str],
                 label_keys: Sequence[str], weight_keys: Sequence[str], loss, batch_size: int, n_global: int,
                 device, train: bool = True, want_values: bool = False, extra_outputs: Sequence[str] = (),
                 extra_parameters: Sequence = (), static_batch: Optional[Dict[str, object]] = None):
        self.name, self.model, self.loss = name, model, loss
        # static_batch: the input columns of the one batch this constraint will ever be bound to; the value requests the
        # expressions made of it (`if float(d["x"][0]) == 0.0:` ...) -- non-empty: the program is specialised to that batch
        self.specialised_to: List[str] = []
        outputs = trace_exprs(model, input_keys, exprs, extra_parameters, static_batch, self.specialised_to)
        # Row slices `expr[a:a+1]` as whole outputs (examples/euler_beam/euler_beam.py:49-54).  The reference's loss then
        # broadcasts the [1, 1] value against the [n, 1] label / weight columns (mse.py:82-105):
        #     sum_p w_p (v_a - l_p)^2  =  W (v_a - lbar)^2 + const,   W = sum w_p,  lbar = sum w_p l_p / W,
        # i.e. a per-point loss on row a alone with label lbar and weight W -- which IS a per-point program: at bind time the
        # label column becomes lbar, the weight column a one-hot W at row a, and const is added to the reported term.
        self._row_slices: Dict[str, int] = {}
        for k, v in list(outputs.items()):
            if isinstance(v, Sym) and v.kind == "rows":
                a, b = v.comp
                b = batch_size if b is None else b
                if (a, b) == (0, batch_size):
                    outputs[k] = v.args[0]
                    continue
                simple = (b - a == 1 and 0 <= a < batch_size and k in label_keys and loss is not None
                          and getattr(loss, "term_kind", 0) == 0 and not getattr(loss, "causal", None)
                          and not getattr(loss, "periodic", False) and n_global == batch_size)
                if not simple:
                    raise NotImplementedError(f"row slice [{a}:{b}] of output {k!r}: only one-row slices under MSELoss on a "
                                              "single rank are lowered to the fused kernels")
                self._row_slices[k] = a
                outputs[k] = v.args[0]
        weight_keys = list(weight_keys) + [k for k in self._row_slices if k not in weight_keys]
        # Batch-coupled outputs (graph.couple: Volterra's `lhs[:N] - int_mat @ u`): the residual exists on the first R rows of
        # the batch only -- the label / weight columns of the reference have R rows (volterra_ide.py: the dataset transform grows
        # the INPUT to N + N Q quadrature points) and the loss is a mean over R.  Here: columns padded to the batch with a zero
        # weight behind row R (bind), the term scale taken for R samples.
        self._coupled: Dict[str, int] = {k: v.comp[0] for k, v in outputs.items() if isinstance(v, Sym) and v.kind == "couple"}
        for k, v in outputs.items():
            if k in self._coupled:
                if v.comp[1] != batch_size or n_global != batch_size:
                    raise NotImplementedError(f"batch-coupled output {k!r}: its matrix has {v.comp[1]} columns for a batch of "
                                              f"{batch_size} points" + (" (single rank only)" if n_global != batch_size else ""))
                if k not in label_keys or loss is None or getattr(loss, "term_kind", 0) != 0:
                    raise NotImplementedError(f"batch-coupled output {k!r} is lowered as an MSELoss term")
        weight_keys = weight_keys + [k for k in self._coupled if k not in weight_keys]
        for k in label_keys:
            if k not in outputs:
                # a label on a raw network output (expression.py: output_dict holds the model outputs too)
                if k in model.output_keys:
                    outputs[k] = Sym.net(model, model.output_keys.index(k))
                else:
                    raise KeyError(f"label key {k!r} is neither an expression nor a network output")
        losses = []
        for k in label_keys:
            losses.append(dict(key=k, label=LABEL_PREFIX + k, weight=(WEIGHT_PREFIX + k) if k in weight_keys else None,
                               area="area" if "area" in input_keys else None,
                               scale=loss.term_scale(k, self._coupled.get(k, n_global)) if loss is not None else 0.0,
                               kind=getattr(loss, "term_kind", 0) if loss is not None else 0,
                               causal=(CAUSAL_PREFIX + k) if getattr(loss, "causal", None) else None,
                               periodic=bool(getattr(loss, "periodic", False))))
        self._loss_rows = losses
        self.low = graph.lower(outputs, losses, extra_outputs, n_global=n_global)
        if getattr(loss, "periodic", False):
            if batch_size % 2:
                raise ValueError(f"Length of output({batch_size}) should be even.")  # mse.py:326-329
            if any(k in weight_keys for k in label_keys):
                raise ValueError("Periodic*Loss with per-point weights: the reference multiplies its [n] pair terms by the "
                                 "[2n] weight column, which does not broadcast (mse.py:335-336)")
        causal = getattr(loss, "causal", None)
        if causal and batch_size % int(causal["n_chunks"]) != 0:
            raise ValueError(f"CausalMSELoss: batch size {batch_size} is not a multiple of n_chunks "
                             f"{causal['n_chunks']} (loss.reshape([n_chunks, -1]), mse.py:168)")
        self.batch_size = batch_size
        self.label_keys = list(label_keys)
        dev = device
        # all per-point arrays of the batch are rows of ONE device block, so that a fresh batch
        # (ContinuousNamedArrayDataset: new points every iteration, array_dataset.py:208-228) is one
        # asynchronous H2D copy from a pinned staging block instead of one blocking copy per array
        n_in, n_aux = len(self.low.input_names), len(self.low.aux_names)
        self._block = torch.zeros((max(1, n_in + n_aux), batch_size), dtype=torch.float32, device=dev)
        inputs = [self._block[i] for i in range(n_in)]
        aux = [self._block[n_in + i] for i in range(n_aux)]
        self._stage: List[torch.Tensor] = []
        self._stage_done: List[Optional[torch.cuda.Event]] = []
        self._flip = 0
        nets = []
        for m, spec, _, idx, pre in self.low.nets:
            lay = m.layout
            if pre is not None:  # input transform: every network input is an [S, N] stream block
                import dataclasses

                from . import _lib as L_

                if any(e_ != L_.EMBED_NONE for e_ in (lay.embed or [])):
                    raise NotImplementedError("periods together with a registered input transform")
                lay = dataclasses.replace(lay, embed=[L_.EMBED_STREAMS] * lay.d_raw, omega=[0.0] * lay.d_raw)
            nets.append((lay, getattr(m, "_param_offset", 0), spec, idx, [(pg.build(), r0, nr) for pg, r0, nr in pre] if pre else None))
        if not nets:
            raise NotImplementedError("a constraint that evaluates no network has nothing to train")
        self.fused = FusedConstraint(name, nets, self.low.streams, self.low.program.build(), inputs, aux,
                                     self.low.loss_keys, want_residual=(want_values or bool(self.low.causal) or bool(self.low.periodic)
                                                                        or bool(self.low.couplings)))
        if self.low.param_slots:
            from .equation.pde.base import EqParamStore

            self.fused.set_eq_params(EqParamStore.get())
        if self.low.couplings:
            if self.low.periodic or self.low.causal or self._row_slices:
                raise NotImplementedError("batch couplings together with a periodic / causal loss or row-sliced outputs")
            cpl = self.low.couplings
            mats = [torch.as_tensor(graph._COUPLE_MATS[it["name"]]).to(dev) for it in cpl["items"]]
            self.fused.set_couplings(cpl["items"], cpl["pv"].build(), cpl["p3"].build(), mats)
        if self.low.reductions:
            if self.low.periodic or self.low.causal or self._row_slices:
                raise NotImplementedError("batch reductions together with a periodic / causal loss or row-sliced outputs")
            red = self.low.reductions
            self.fused.set_reductions(red["p1"].build(), red["p3"].build(), red["k"])
        if self.low.periodic:
            self.fused.set_periodic(self.low.periodic)
        if self.low.causal:
            self.fused.set_causal(self.low.causal, int(causal["n_chunks"]), float(causal["tol"]))
        self.train = train

    def bind(self, input: Dict[str, object], label: Optional[Dict[str, object]], weight: Optional[Dict[str, object]]):
        """Upload one batch (named [n,1] arrays) into the constraint's device buffers."""
        names = list(self.low.input_names) + list(self.low.aux_names)
        if self._row_slices:
            label, weight = self._bind_row_slices(input, dict(label or {}), dict(weight or {}))
        if self._coupled:
            label, weight = self._bind_coupled(dict(label or {}), dict(weight or {}))
        srcs: List[object] = []
        for i, name in enumerate(names):
            if i < len(self.low.input_names):
                srcs.append(input[name])
            elif name.startswith(LABEL_PREFIX):
                srcs.append(label[name[len(LABEL_PREFIX):]])
            elif name.startswith(WEIGHT_PREFIX):
                w = weight[name[len(WEIGHT_PREFIX):]]
                if name[len(WEIGHT_PREFIX):] in self._row_slices:
                    srcs.append(w)  # already the effective column
                    continue
                srcs.append(self.loss.batch_weight(w) if hasattr(self.loss, "batch_weight") else w)
            elif name.startswith(CAUSAL_PREFIX) or name.startswith((graph.COUPLE_RHS_PREFIX, graph.COUPLE_VBAR_PREFIX)):
                srcs.append(None)  # written on the device every step (engine.FusedConstraint.forward)
            else:
                srcs.append(input[name])
        if not self._block.is_cuda or any(isinstance(v, torch.Tensor) and v.is_cuda for v in srcs):
            for row, src in zip(self._block, srcs):
                if src is not None:
                    row.copy_(_to_dev(src, self._block.device))
            return
        if not self._stage:  # two pinned blocks: batch k+1 is staged while the copy of batch k may be in flight
            self._stage = [torch.zeros(self._block.shape, dtype=torch.float32).pin_memory() for _ in range(2)]
            self._stage_done = [None, None]
        k = self._flip
        self._flip ^= 1
        if self._stage_done[k] is not None:
            self._stage_done[k].synchronize()
        st = self._stage[k].numpy()
        for i, src in enumerate(srcs):
            if src is not None:
                a = src.detach().numpy() if isinstance(src, torch.Tensor) else np.asarray(src)
                np.copyto(st[i], a.reshape(-1), casting="same_kind" if a.dtype.kind == "f" else "unsafe")
        self._block.copy_(self._stage[k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._stage_done[k] = ev

    def _bind_coupled(self, label, weight):
        """Label / weight columns of batch-coupled outputs: R rows in the reference, padded to the batch with weight 0."""
        n = self.batch_size

        def host(a, rows):
            a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
            a = np.asarray(a, dtype=np.float32).reshape(-1)
            if a.size == 1:
                a = np.full(rows, a[0], np.float32)
            if a.size != rows:
                raise ValueError(f"a batch-coupled output has {rows} residual rows, its label / weight column {a.size}")
            return a

        for k, rows in self._coupled.items():
            lab = np.zeros(n, np.float32)
            if k in label:
                lab[:rows] = host(label[k], rows)
            w = np.zeros(n, np.float32)
            w[:rows] = host(weight[k], rows) if weight.get(k) is not None else 1.0
            label[k], weight[k] = lab.reshape(n, 1), w.reshape(n, 1)
        return label, weight

    def _bind_row_slices(self, input, label, weight):
        """Label / weight columns of the row-sliced outputs (see __init__) and the constant part of their loss terms."""
        def host(a):
            a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
            return a.astype(np.float64).reshape(-1)

        # same label / weight / area objects as the last bind (a static batch re-bound, or a loader that yields the same
        # arrays): the columns and the constant are unchanged -- no device-to-host copies on the step path
        # The key is identity AND content: a loader may refill the same buffers in place (tensors: the autograd version
        # counter, bumped by every in-place write; host arrays: a CRC of the bytes -- no device traffic either way)
        def stamp(a):
            if isinstance(a, torch.Tensor):
                return ("t", id(a), a._version)
            if isinstance(a, np.ndarray):
                return ("n", id(a), zlib.crc32(memoryview(np.ascontiguousarray(a)).cast("B")))
            return ("v", id(a), repr(a))

        key = tuple(stamp(label.get(k)) for k in self._row_slices) + tuple(stamp(weight.get(k)) for k in self._row_slices) + (stamp(input.get("area")),)
        sources = ([label.get(k) for k in self._row_slices], [weight.get(k) for k in self._row_slices], input.get("area"))
        cached = getattr(self, "_row_slice_cache", None)
        if cached is not None and cached[0] == key:
            for k in self._row_slices:
                label[k], weight[k] = cached[1][k]
            return label, weight
        n = self.batch_size
        area = host(input["area"]) if ("area" in input and "area" in self.low.input_names + self.low.aux_names) else np.ones(n)
        offsets = {}
        for k, a in self._row_slices.items():
            lab = np.broadcast_to(host(label[k]), (n,)) if k in label else np.zeros(n)
            w = weight.get(k)
            w = np.ones(n) if w is None else np.broadcast_to(host(self.loss.batch_weight(w) if hasattr(self.loss, "batch_weight")
                                                                  else w), (n,))
            we = w * area
            W = float(we.sum())
            lbar = float((we * lab).sum() / W) if W != 0.0 else 0.0
            const = float((we * lab * lab).sum() - W * lbar * lbar)
            col = np.zeros(n, np.float32)
            col[a] = W / area[a] if area[a] != 0.0 else 0.0
            label[k] = np.full((n, 1), lbar, np.float32)
            weight[k] = col.reshape(n, 1)
            scale = next(float(r["scale"]) for r in self._loss_rows if r["key"] == k)
            offsets[k] = scale * const
        self.fused.loss_offsets = offsets
        # (the cache holds the source objects too, so that their ids cannot be recycled while it is valid)
        self._row_slice_cache = (key, {k: (label[k], weight[k]) for k in self._row_slices}, sources)
        return label, weight

    def values(self) -> Dict[str, torch.Tensor]:
        """Per-point values of every loss key / extra output ([n,1] tensors), after a forward."""
        assert self.fused.resid is not None
        out = {k: self.fused.resid[i].view(-1, 1) for i, k in enumerate(self.low.loss_keys)}
        for k, a in self._row_slices.items():  # `expr[a:a+1]` outputs are [1, 1] in the reference (euler_beam.py:49-54)
            if k in out:
                out[k] = out[k][a:a + 1]
        return out
