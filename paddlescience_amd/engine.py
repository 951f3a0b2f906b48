"""Execution engine of the hot path: owns the flat parameter / gradient / Adam-moment buffers and
the per-constraint device buffers, and issues the five kernels of one training step.

One step == `train_epoch_func` body of the reference (/root/reference/ppsci/solver/train.py:82-184):
  for every constraint: model forward + expression forward + loss   (expression.py:89-126)
  total_loss.backward()                                             (train.py:158)
  fused_allreduce_gradients                                         (train.py:168-171)
  optimizer.step(); clear_grad()                                    (train.py:175-180)
Here: taylor_fwd -> epilogue -> taylor_bwd per constraint, one reduce_rows into the flat gradient,
one RCCL all-reduce (torch.distributed, SUM) of that flat buffer when world_size > 1, one fused Adam.
"""
from __future__ import annotations

import os
import weakref
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib as L
from . import hotpath as hp


_EPI_LOSSES = os.environ.get("PPSCI_EPI_LOSSES", "1") != "0"  # loss terms finished inside the epilogue launch


class FusedConstraint:
    """Device-resident state of one constraint batch: inputs, aux arrays, streams, stash, partials.

    `nets`: one entry per network the constraint evaluates -- (layout, offset of its parameters in the flat
    parameter buffer, StreamSpec in its own input order, indices of its inputs among `inputs`); a plain layout stands
    for a single network at offset 0 that takes all inputs.  The members' streams are consecutive row blocks of U."""

    def __init__(self, name: str, nets, streams: hp.StreamSpec, edesc: L.EpilogueDesc,
                 inputs: Sequence[torch.Tensor], aux: Sequence[torch.Tensor], loss_keys: Sequence[str],
                 want_residual: bool = False):
        self.name = name
        self.streams, self.edesc = streams, edesc
        self.inputs = [t.contiguous().view(-1) for t in inputs]
        self.aux = [t.contiguous().view(-1) for t in aux]
        self.n = self.inputs[0].numel()
        self.loss_keys = list(loss_keys)
        dev = self.inputs[0].device
        f32 = dict(dtype=torch.float32, device=dev)
        if isinstance(nets, hp.NetLayout):
            nets = [(nets, 0, streams, list(range(len(self.inputs))), None)]
        nets = [tuple(nt) + (None,) * (5 - len(nt)) for nt in nets]
        q = sum(nt[0].d_out for nt in nets) * streams.S
        self.U = torch.zeros((q, self.n), **f32)
        self.Ubar = torch.zeros((q, self.n), **f32)  # rows never loaded by the program stay 0
        self.nets = []
        row = 0
        for lay, off, spec, idx, pre in nets:
            if getattr(lay, "is_pirate", False):
                # PirateNet runs layer by layer (arch/piratenet.py): its executor stands in for taylor_fwd / taylor_bwd and
                # delivers ONE gradient row in the trainable layout
                if pre is not None:
                    raise NotImplementedError("PirateNet with a registered input transform")
                nr = lay.d_out * streams.S
                net_inputs = [self.inputs[j] for j in idx]
                self.nets.append(dict(layout=lay, off=off, desc=None, inputs=net_inputs, pre=None,
                                      exec=lay.make_exec(spec, self.n, net_inputs),
                                      U=self.U[row:row + nr], Ubar=self.Ubar[row:row + nr], stash=None, grad_rows=1,
                                      grad_partials=torch.empty((1, lay.n_params), **f32), workspace=None))
                row += nr
                continue
            desc = lay.desc(spec)
            rows = hp.bwd_partial_rows(desc, self.n)
            if rows <= 0:
                raise NotImplementedError(
                    f"network {lay.n_hidden}x{lay.width} with {spec.S} streams has no HIP reverse kernel")
            nr = lay.d_out * streams.S
            if pre is not None:
                # input transform: the epilogue VM computes every input feature and its derivative streams from the
                # raw variables ([d_in * S, N], <= MAX_RES rows per program); the kernels read [S, N] blocks
                feat = torch.zeros((lay.d_raw * streams.S, self.n), **f32)
                net_inputs = [feat[k * streams.S:(k + 1) * streams.S].reshape(-1) for k in range(lay.d_raw)]
                pre = [(ed, feat[r0:r0 + nr]) for ed, r0, nr in pre]
            else:
                net_inputs = [self.inputs[j] for j in idx]
            self.nets.append(dict(
                layout=lay, off=off, desc=desc, inputs=net_inputs, pre=pre,
                U=self.U[row:row + nr], Ubar=self.Ubar[row:row + nr],
                stash=torch.empty(hp.stash_bytes(desc, self.n) // 4, **f32), grad_rows=rows,
                grad_partials=torch.empty((rows, lay.n_params), **f32),
                workspace=torch.empty(max(4, hp.bwd_workspace_bytes(desc, self.n) // 4), **f32)))
            row += nr
        first = self.nets[0]  # single-network accessors (bench.py, tests, tools)
        self.layout, self.desc, self.stash = first["layout"], first["desc"], first["stash"]
        self.grad_rows, self.grad_partials, self.workspace = first["grad_rows"], first["grad_partials"], first["workspace"]
        self.work = sum(self.n * nt["layout"].n_params * streams.S for nt in self.nets)
        self.loss_rows = hp.epilogue_partial_rows(self.n)
        self._pre_partials = torch.zeros((self.loss_rows, L.MAX_RES), **f32)  # unused sums of the stream programs
        self.loss_partials = torch.zeros((self.loss_rows, max(1, edesc.n_res)), **f32)
        self.loss_terms = torch.zeros(max(1, edesc.n_res), **f32)
        self._loss_counter = torch.zeros(1, dtype=torch.int32, device=dev)  # ticket counter of ppsci_epilogue_losses
        self.resid = torch.zeros((max(1, edesc.n_res), self.n), **f32) if want_residual else None

    def set_inputs(self, inputs: Sequence[torch.Tensor], aux: Optional[Sequence[torch.Tensor]] = None) -> None:
        """New batch of the same size (ContinuousNamedArrayDataset, array_dataset.py:208-228)."""
        for dst, src in zip(self.inputs, inputs):
            dst.copy_(src.view(-1))
        if aux is not None:
            for dst, src in zip(self.aux, aux):
                dst.copy_(src.view(-1))

    def set_eq_params(self, store) -> None:
        """The epilogue reads learnable equation parameters (OP_LD_PARAM): their adjoints are summed per block into
        `eq_partials` and, in backward(), added into the store's gradient vector."""
        self.eq_store = store
        self.eq_partials = torch.zeros((self.loss_rows, L.MAX_EPARAM), dtype=torch.float32, device=self.U.device)

    def set_reductions(self, p1: L.EpilogueDesc, p3: L.EpilogueDesc, k: int) -> None:
        """Batch reductions in the expressions (graph.Sym.mean / .sum, graph.lower): `p1` sums the k summands (LINEAR terms),
        the residual program reads the sums from parameter slots 0..k-1, `p3` is the residual program + the k summands seeded
        with dL/dR_k.  red_values = [R_0..7 | dL/dR_0..7] on the device; nothing of it ever visits the host."""
        f32 = dict(dtype=torch.float32, device=self.U.device)
        self.reductions = dict(p1=p1, p3=p3, k=k)
        self.red_values = torch.zeros(2 * L.MAX_EPARAM, **f32)
        self.red_partials = torch.zeros((self.loss_rows, L.MAX_EPARAM), **f32)
        self._red_l1 = torch.zeros((self.loss_rows, k), **f32)
        self._red_l3 = torch.zeros((self.loss_rows, max(1, p3.n_res)), **f32)
        self._red_p3 = torch.zeros((self.loss_rows, L.MAX_EPARAM), **f32)  # (pass 3's own parameter adjoints: not used)

    def _forward_reductions(self, train: bool) -> None:
        """Three launches of the epilogue VM around two fixed-order row sums (see set_reductions); under data parallelism
        the sums and their adjoints are all-reduced (SUM): a mean is over the GLOBAL batch, as the loss is."""
        rd, k, n = self.reductions, self.reductions["k"], self.n
        dist = torch.distributed
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        hp.epilogue(rd["p1"], n, self.inputs, self.U, self.aux, None, None, self._red_l1)
        hp.reduce_rows(self._red_l1, self.loss_rows, k, self.red_values[:k], False)
        if multi:
            dist.all_reduce(self.red_values[:k], op=dist.ReduceOp.SUM)
        hp.epilogue(self.edesc, n, self.inputs, self.U, self.aux, self.resid, self.Ubar if train else None, self.loss_partials,
                    self.red_values, self.red_partials if train else None)
        hp.reduce_rows(self.loss_partials, self.loss_rows, max(1, self.edesc.n_res), self.loss_terms, False)
        if not train:
            return
        hp.reduce_rows(self.red_partials, self.loss_rows, L.MAX_EPARAM, self.red_values[L.MAX_EPARAM:], False)
        if multi:
            dist.all_reduce(self.red_values[L.MAX_EPARAM:], op=dist.ReduceOp.SUM)
        hp.epilogue(rd["p3"], n, self.inputs, self.U, self.aux, None, self.Ubar, self._red_l3, self.red_values, self._red_p3)

    def set_couplings(self, items: Sequence[dict], pv: L.EpilogueDesc, p3: L.EpilogueDesc, matrices: Sequence[torch.Tensor]) -> None:
        """Batch-coupled residuals lhs[:R] - M v (graph.couple; ppsci/equation/ide/volterra.py:66-77): `pv` evaluates every v into
        a row of `_cv`, `p3` is the residual program + one LINEAR term per coupling on v with the per-point weight column vbar."""
        f32 = dict(dtype=torch.float32, device=self.U.device)
        assert self.resid is not None
        self.couplings = dict(items=list(items), pv=pv, p3=p3, M=list(matrices))
        self._cv = torch.zeros((len(items), self.n), **f32)
        self._cl1 = torch.zeros((self.loss_rows, len(items)), **f32)
        self._cl3 = torch.zeros((self.loss_rows, max(1, p3.n_res)), **f32)

    def _forward_couplings(self, train: bool) -> None:
        """values of v -> rhs = M v (aux column) -> residual program (loss, dL/dU with rhs held fixed, residual rows) ->
        vbar = -M^T (2 scale w r) (aux column) -> residual program + the LINEAR terms on v: dL/dU complete."""
        cp, n = self.couplings, self.n
        hp.epilogue(cp["pv"], n, self.inputs, self.U, self.aux, self._cv, None, self._cl1)
        for j, (it, M) in enumerate(zip(cp["items"], cp["M"])):
            hp.dense_matvec(M, self._cv[j], self.aux[it["rhs_aux"]], 1.0, False)
        hp.epilogue(self.edesc, n, self.inputs, self.U, self.aux, self.resid, self.Ubar if train else None, self.loss_partials)
        hp.reduce_rows(self.loss_partials, self.loss_rows, max(1, self.edesc.n_res), self.loss_terms, False)
        if not train:
            return
        for it, M in zip(cp["items"], cp["M"]):
            scale = float(self.edesc.res[it["res_row"]].scale)
            hp.dense_matvec(M, self.resid[it["res_row"]], self.aux[it["vbar_aux"]], -2.0 * scale, True,
                            rowscale=self.aux[it["weight_aux"]])
        hp.epilogue(cp["p3"], n, self.inputs, self.U, self.aux, None, self.Ubar, self._cl3)

    def set_causal(self, rows: Sequence[tuple], n_chunks: int, tol: float) -> None:
        """CausalMSELoss (mse.py:109-189): rows = (residual row, label aux, weight aux, area aux, factor aux)."""
        self.causal, self.n_chunks, self.tol = list(rows), n_chunks, tol
        self.chunk_scratch = torch.zeros((len(self.causal), n_chunks), dtype=torch.float32, device=self.U.device)

    def set_periodic(self, rows: Sequence[tuple]) -> None:
        """Periodic*Loss (loss/periodic.py): rows = (residual row, label aux); batch = [points ; periodic images]."""
        self.periodic = list(rows)

    def forward(self, params: torch.Tensor, train: bool) -> None:
        for nt in self.nets:
            if nt["pre"] is not None:
                for ed, rows in nt["pre"]:
                    hp.epilogue(ed, self.n, self.inputs, None, self.aux, rows, None, self._pre_partials)
            if "exec" in nt:
                nt["exec"].forward(params[nt["off"]:nt["off"] + nt["layout"].n_params], nt["U"], train)
                continue
            hp.taylor_fwd(nt["desc"], params[nt["off"]:nt["off"] + nt["layout"].n_params], nt["inputs"], nt["U"],
                          nt["stash"] if train else None, self.n)
        if getattr(self, "reductions", None):
            return self._forward_reductions(train)
        if getattr(self, "couplings", None):
            return self._forward_couplings(train)
        if getattr(self, "causal", None):
            # first pass: the per-point values only; then the causal factor of every key from its window means
            # (constants for the reverse sweep: `.detach()`, mse.py:174); the pass below then weights with them
            hp.epilogue(self.edesc, self.n, self.inputs, self.U, self.aux, self.resid, None, self.loss_partials,
                        *self._eq_args(False))
            ax = lambda i: self.aux[i] if i >= 0 else None  # noqa: E731
            for j, (row, lab, w, ar, cw) in enumerate(self.causal):
                hp.causal_weights(self.n_chunks, self.tol, self.resid[row], ax(lab), ax(w), ax(ar),
                                  self.chunk_scratch[j], self.aux[cw])
        periodic = getattr(self, "periodic", None)
        if periodic:
            # first pass: values only; every point's label becomes its partner's value (a constant for the reverse
            # sweep), so the per-point loss below has the pair loss's gradient and twice its value
            hp.epilogue(self.edesc, self.n, self.inputs, self.U, self.aux, self.resid, None, self.loss_partials,
                        *self._eq_args(False))
            h = self.n // 2
            for row, lab in periodic:
                self.aux[lab][:h].copy_(self.resid[row][h:])
                self.aux[lab][h:].copy_(self.resid[row][:h])
        if not periodic and self.edesc.n_res >= 1 and _EPI_LOSSES:
            # the loss terms come out of the epilogue launch itself (the workgroup that finishes last sums the rows)
            hp.epilogue(self.edesc, self.n, self.inputs, self.U, self.aux, self.resid, self.Ubar if train else None,
                        self.loss_partials, *self._eq_args(train), loss_terms=self.loss_terms, counter=self._loss_counter)
            return
        hp.epilogue(self.edesc, self.n, self.inputs, self.U, self.aux, self.resid, self.Ubar if train else None,
                    self.loss_partials, *self._eq_args(train))

        hp.reduce_rows(self.loss_partials, self.loss_rows, max(1, self.edesc.n_res), self.loss_terms, False)
        if periodic:
            for row, _ in periodic:
                self.loss_terms[row:row + 1].mul_(0.5)

    def _eq_args(self, train: bool):
        st = getattr(self, "eq_store", None)
        return (None, None) if st is None else (st.values, self.eq_partials if train else None)

    def backward(self, params: torch.Tensor, out: Optional[torch.Tensor] = None) -> bool:
        """Reverse sweeps of every member network.  `out` (the flat gradient): a single-network constraint whose reverse
        sweep finishes with ONE gradient row writes it there directly -- no copy pass -- and True is returned."""
        direct = out is not None and len(self.nets) == 1 and self.nets[0]["grad_rows"] == 1
        for nt in self.nets:
            n = nt["layout"].n_params
            dst = out[nt["off"]:nt["off"] + n].view(1, n) if direct else nt["grad_partials"]
            if "exec" in nt:
                nt["exec"].backward(params[nt["off"]:nt["off"] + n], nt["Ubar"], dst)
                continue
            hp.taylor_bwd(nt["desc"], params[nt["off"]:nt["off"] + n], nt["inputs"], nt["Ubar"],
                          nt["stash"], nt["workspace"], dst, self.n)
        return direct

    def reduce_grads(self, grad: torch.Tensor, accumulate: bool) -> None:
        """grad[member's slice] (+)= this constraint's gradient of that member (fixed order)."""
        for nt in self.nets:
            n = nt["layout"].n_params
            hp.reduce_rows(nt["grad_partials"], nt["grad_rows"], n, grad[nt["off"]:nt["off"] + n], accumulate)

    def one_launch_ready(self) -> bool:
        """Can this constraint's whole step run as ONE launch (hp.taylor_step)?  A single plain network fed by all the
        constraint's inputs, a program without learnable equation parameters, none of the two-pass losses (causal,
        periodic), and a one-launch kernel for the net / stream set; the workspace is allocated on the first yes."""
        ok = getattr(self, "_one_launch", None)
        if ok is None:
            nt = self.nets[0]
            ok = (len(self.nets) == 1 and "exec" not in nt and nt["pre"] is None and nt["off"] == 0
                  and len(nt["inputs"]) == len(self.inputs)
                  and all(a.data_ptr() == b.data_ptr() for a, b in zip(nt["inputs"], self.inputs))
                  and self.edesc.n_res >= 1)
            self._step_kind = hp.STEP_NONE
            if ok:
                nbytes = hp.taylor_step_workspace_bytes(nt["desc"], self.edesc, self.n)
                ok = nbytes > 0
                if ok:
                    self._step_kind = hp.taylor_step_kind(nt["desc"], self.edesc, self.n)
                    self._step_ws = torch.zeros(nbytes // 4, dtype=torch.float32, device=self.U.device)
            self._one_launch = ok
        return (ok and not getattr(self, "causal", None) and not getattr(self, "periodic", None)
                and getattr(self, "eq_store", None) is None and not getattr(self, "reductions", None)
                and not getattr(self, "couplings", None))

    def one_launch_wins(self, max_points_single_wave: int) -> bool:
        """one_launch_ready() and the one-launch kernel is the faster path at this batch size: the fused tile kernel
        (padded width 64) at any size, the single-wave kernel (padded width 32) for small batches only."""
        return self.one_launch_ready() and (self._step_kind == hp.STEP_FUSED_TILE or self.n <= max_points_single_wave)

    def step_one_launch(self, params: torch.Tensor, grad: torch.Tensor, accumulate: bool, adam: Optional[dict]) -> None:
        plan = getattr(self, "_step_plan", None)
        if plan is None or plan.key != (params.data_ptr(), grad.data_ptr()):
            nt = self.nets[0]
            # the fused tile kernel keeps U, dL/dU and the stash on the chip; it writes U / dL/dU out only when asked to
            # (`step_outputs`: tests, tools) -- a training step does not read them
            outs = self._step_kind != hp.STEP_FUSED_TILE or getattr(self, "step_outputs", False)
            plan = self._step_plan = hp.StepPlan(nt["desc"], self.edesc, params, self.n, self.inputs, self.aux,
                                                 self.U if outs else None, self.Ubar if outs else None, self.resid,
                                                 nt["stash"] if self._step_kind != hp.STEP_FUSED_TILE else None, self._step_ws,
                                                 self.loss_terms, grad)
        plan.run(self.edesc, accumulate, adam)

    def losses(self) -> Dict[str, float]:
        vals = self.loss_terms.detach().cpu().tolist()  # one device->host sync, only when logging
        off = getattr(self, "loss_offsets", None) or {}  # constant parts of row-sliced terms (compile.CompiledConstraint)
        return {k: vals[i] + off.get(k, 0.0) for i, k in enumerate(self.loss_keys)}


class StepGraph:
    """Capture-once / replay of a fixed launch sequence (every argument a persistent device buffer) as a HIP graph,
    through torch.cuda.CUDAGraph on the launch stream.  First call with a key: eager (also the warm-up a capture
    needs); second: capture + replay; later: replay.  A failed capture is LOGGED (a silent fallback would hide a
    performance regression) and the step runs eagerly from then on.  At most `max_graphs` captured graphs are kept
    (least recently used first out): a key that changes every iteration must not grow the table without bound."""

    def __init__(self, enabled: bool, max_graphs: int = 16):
        self.enabled = enabled
        self.max_graphs = max_graphs
        self._graphs: Dict[tuple, object] = {}

    def run(self, key: tuple, eager) -> None:
        if not self.enabled:
            return eager()
        g = self._graphs.pop(key, None)
        if g is None:
            eager()
            self._graphs[key] = False
            self._evict()
            return
        self._graphs[key] = g  # most recently used last
        if g is False:
            try:
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, capture_error_mode="relaxed"):
                    eager()
                self._graphs[key] = graph
                graph.replay()
            except Exception as e:  # noqa: BLE001 -- capture is an optimisation, never a requirement
                from .utils import logger

                logger.warning(f"HIP-graph capture of the training step failed ({type(e).__name__}: {e}); the step is "
                               "launched kernel by kernel from now on (PPSCI_HIP_GRAPH=0 silences this)")
                self.enabled = False
                self._graphs.clear()
                torch.cuda.synchronize()
                eager()
            return
        g.replay()

    def _evict(self) -> None:
        while len(self._graphs) > self.max_graphs:
            self._graphs.pop(next(iter(self._graphs)))

    def clear(self) -> None:
        self._graphs.clear()


def step_with_adam(eng, constraints, opt, params: torch.Tensor, grad_scale: float = 1.0) -> None:
    """One training step of an engine whose backward pass ENDS in row reductions (spinn_engine.SpinnEngine: the gradient rows of
    the branch nets + the loss rows; operator_engine.OperatorEngine: the partials of the 1x1 convolutions' weight gradients),
    with a plain Adam (`opt`: optimizer._AdamState without clipping / decay) on one rank: forward + backward without those
    reductions, then ONE launch that sums the rows and applies the update (hp.reduce_rows_multi_adam) -- instead of a
    reduction launch followed by the optimizer's launch.  More reductions than one launch takes, or rows that accumulate over
    several constraints: the reductions are flushed and the optimizer steps as usual."""
    if os.environ.get("PPSCI_FUSED_REDUCE_ADAM", "1") == "0":  # (A/B measurements, tests)
        eng.forward_backward(constraints)
        opt.step(eng.grad, grad_scale)
        return
    segs = eng.forward_backward_deferred(constraints)
    if segs is None or len(segs) > 16:
        if segs is not None:
            eng.flush_deferred(segs)
        opt.step(eng.grad, grad_scale)
        return
    opt.t += 1
    hp.reduce_rows_multi_adam(segs, params, eng.grad, opt.m, opt.v, opt.get_lr(), opt.t, opt.beta1, opt.beta2, opt.epsilon,
                              grad_scale)


def run_on_streams(streams: List["torch.cuda.Stream"], jobs) -> None:
    """Run independent launch sequences concurrently, one HIP stream each, forked from and joined back into the
    current stream (inside a capture: parallel branches of the graph)."""
    cur = torch.cuda.current_stream()
    while len(streams) < len(jobs):
        streams.append(torch.cuda.Stream())
    for job, st in zip(jobs, streams):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            job()
    for st in streams[:len(jobs)]:
        cur.wait_stream(st)


_NATIVE_COMM = None


def native_comm_ready() -> bool:
    """PPSCI_NATIVE_ALLREDUCE=1 on a multi-rank GPU job: create (once) the RCCL communicator of the C ABI (csrc/comm.hip) -- rank
    0's 128-byte id travels over the existing torch.distributed group -- and use ppsci_allreduce_sum for the gradient all-reduce.
    Default (unset / 0): torch.distributed.all_reduce, which is the same RCCL underneath."""
    global _NATIVE_COMM
    if _NATIVE_COMM is None:
        dist = torch.distributed
        if (os.environ.get("PPSCI_NATIVE_ALLREDUCE", "0") != "1" or not dist.is_initialized() or dist.get_world_size() < 2
                or not torch.cuda.is_available() or L.is_emulated()):
            _NATIVE_COMM = False
        else:
            import ctypes as C

            lib = L.lib()
            buf = C.create_string_buffer(128)
            if dist.get_rank() == 0:
                L.check(lib.ppsci_comm_unique_id(buf))
            box = [buf.raw if dist.get_rank() == 0 else None]
            dist.broadcast_object_list(box, src=0)
            L.check(lib.ppsci_comm_init(dist.get_rank(), dist.get_world_size(), C.c_char_p(box[0])))
            _NATIVE_COMM = True
    return _NATIVE_COMM


def _release_fragments(ptr: int) -> None:
    try:
        L.lib().ppsci_release_fragments(ptr)
    except Exception:  # interpreter shutdown
        pass


class Engine:
    def __init__(self, layout: hp.NetLayout, params: torch.Tensor, beta1=0.9, beta2=0.999, eps=1e-8,
                 dp_reduce: str = "sum"):
        assert layout is None or params.numel() == layout.n_params  # None: several networks (ModelList)
        self.layout = layout
        self.params = params
        # the library keeps pre-split weight fragments per parameter buffer (feature-split kernels): released with the engine
        fin = weakref.finalize(self, _release_fragments, params.data_ptr())
        fin.atexit = False
        self.grad = torch.zeros_like(params)
        self.m = torch.zeros_like(params)
        self.v = torch.zeros_like(params)
        self.t = 0
        self.beta1, self.beta2, self.eps = beta1, beta2, eps
        self.dp_reduce = dp_reduce
        self.world = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
        # The forward / epilogue / reverse / reduction kernels of all constraints (7 launches per constraint,
        # every argument a fixed device buffer) are captured once into a HIP graph and replayed: small configs
        # (Laplace2D: 10 k points, 2 constraints) are launch-bound from Python otherwise.  PPSCI_HIP_GRAPH=0
        # turns it off; a failed capture falls back to eager launches for good.
        self.use_graph = params.is_cuda and os.environ.get("PPSCI_HIP_GRAPH", "1") != "0"
        self._step_graph = StepGraph(self.use_graph)
        self.graph_max_work = 1.0e9  # points x parameters x streams per step below which the step is launch-bound
        self.multi_stream = os.environ.get("PPSCI_MULTI_STREAM", "1") != "0"
        self.multi_stream_max_points = 16384  # a constraint this small cannot fill the chip on its own
        self._streams: List[torch.cuda.Stream] = []
        # One launch per constraint (forward -> epilogue -> reverse -> reduction [-> Adam] in one kernel) for steps whose
        # constraints are all small: 7 launches of 4-5 us dispatch + drain each become one.  PPSCI_ONE_LAUNCH=0 turns it off.
        # Steps of ONE constraint only by default: the launches of several constraints would run one after the other, while
        # the separate kernels of small constraints run concurrently as parallel branches of the captured graph (measured on
        # MI355X, Laplace2D example shape 10 201 + 400 points on a 5 x 20 net: 115 us against 97 us).
        self.one_launch = os.environ.get("PPSCI_ONE_LAUNCH", "1") != "0"
        self.one_launch_max_points = 16384
        self.one_launch_max_constraints = 1

    def one_launch_ready(self, constraints: Sequence[FusedConstraint]) -> bool:
        return (self.one_launch and self.layout is not None and 0 < len(constraints) <= self.one_launch_max_constraints
                and all(isinstance(c, FusedConstraint) and c.one_launch_wins(self.one_launch_max_points)
                        for c in constraints))

    def step_one_launch(self, constraints: Sequence[FusedConstraint], adam: Optional[dict] = None) -> None:
        """Gradient (and loss terms) of the step, constraint by constraint in order, one launch each; `adam` (see
        hp.taylor_step) makes the LAST launch apply the optimizer step as well -- single rank only."""
        last = len(constraints) - 1
        for i, c in enumerate(constraints):
            c.step_one_launch(self.params, self.grad, i > 0, adam if i == last else None)

    def _forward_backward_eager(self, constraints: Sequence[FusedConstraint]) -> None:
        # Constraints are independent until their gradients are summed.  A small one (a boundary or initial
        # condition with a few hundred points, or the reference's 4 096-point PDE batches) occupies a fraction of
        # the 256 CUs for the latency of one tile's layer chain, so such sets run concurrently, one HIP stream per
        # constraint, and join before the fixed-order gradient sum (also inside a captured graph: parallel
        # branches).  PPSCI_MULTI_STREAM=0 turns it off.
        several = self.layout is None  # ModelList: a constraint may touch only some members' slices of the gradient,
        if several:                    # so start from zero and always accumulate
            self.grad.zero_()
        if (self.multi_stream and len(constraints) > 1 and self.params.is_cuda
                and min(c.n for c in constraints) <= self.multi_stream_max_points):
            def job(c):
                if self.one_launch and c.one_launch_wins(self.one_launch_max_points):
                    # forward -> epilogue -> reverse -> reduction in ONE launch per constraint, the constraints' launches
                    # as parallel branches; the gradient row lands where the separate kernels would leave it
                    row = c.nets[0]["grad_partials"]
                    return lambda: c.step_one_launch(self.params, row.view(-1), False, None)
                return lambda: (c.forward(self.params, True), c.backward(self.params))

            run_on_streams(self._streams, [job(c) for c in constraints])
            for i, c in enumerate(constraints):
                c.reduce_grads(self.grad, several or i > 0)
        else:
            for i, c in enumerate(constraints):
                if (self.one_launch and not several and isinstance(c, FusedConstraint) and c.one_launch_ready()
                        and c._step_kind == hp.STEP_FUSED_TILE):
                    # padded width 64: the fused tile kernel (no stash round trip through HBM), gradient (+)= in place
                    c.step_one_launch(self.params, self.grad, i > 0, None)
                    continue
                c.forward(self.params, True)
                # the first constraint's gradient row goes straight into the flat gradient (no copy kernel)
                if not c.backward(self.params, self.grad if (i == 0 and not several) else None):
                    c.reduce_grads(self.grad, several or i > 0)
        # d loss / d (learnable equation parameter): per-block sums of every constraint that reads one, in order
        first = True
        for c in constraints:
            if getattr(c, "eq_store", None) is not None:
                hp.reduce_rows(c.eq_partials, c.loss_rows, L.MAX_EPARAM, c.eq_store.grad, not first)
                first = False

    def forward_backward(self, constraints: Sequence[FusedConstraint]) -> None:
        if self.one_launch_ready(constraints):
            return self.step_one_launch(constraints)
        if not self.use_graph:
            return self._forward_backward_eager(constraints)
        # launch-bound only: a replayed HIP graph adds ~1.5 us of dependency handling per kernel node, which costs
        # the GPU-bound 100 k-point Allen-Cahn step 2 % (0.458 -> 0.468 ms) while it makes Laplace2D 4x faster
        # ... and so does a step whose constraints are all small (the reference's 4 096-point batches on a 4 x 256
        # net: ~16 launches of 10-250 us, issued from Python in about the time the GPU needs for them)
        work = sum(c.work for c in constraints)
        if work > self.graph_max_work and max(c.n for c in constraints) > self.multi_stream_max_points:
            return self._forward_backward_eager(constraints)
        if self.world > 1 and any(getattr(c, "reductions", None) for c in constraints):
            # batch reductions all-reduce their sums BETWEEN the launches of a constraint: collectives stay out of captured graphs
            return self._forward_backward_eager(constraints)
        self._step_graph.enabled = self.use_graph
        self._step_graph.run(tuple(id(c) for c in constraints), lambda: self._forward_backward_eager(constraints))
        self.use_graph = self._step_graph.enabled

    def invalidate_graphs(self) -> None:
        """Call after changing anything a captured launch holds by value (residual scales of an epilogue)."""
        self._step_graph.clear()

    def allreduce(self) -> None:
        # ONE collective per step: the flat fp32 gradient, SUM, in place, through torch.distributed (backend "nccl" is RCCL
        # over xGMI on ROCm; "gloo" in the CPU tests).  Matches fused_allreduce_gradients (/root/reference/ppsci/solver/train.py:168-171).
        if self.world > 1:
            if native_comm_ready():  # PPSCI_NATIVE_ALLREDUCE=1: the C ABI's RCCL communicator (csrc/comm.hip)
                L.check(L.lib().ppsci_allreduce_sum(hp._p(self.grad), self.grad.numel(), hp._stream_ptr(self.grad)))
            else:
                torch.distributed.all_reduce(self.grad, op=torch.distributed.ReduceOp.SUM)

    def optimizer_step(self, lr: float) -> None:
        self.t += 1
        scale = (1.0 / self.world) if (self.dp_reduce == "mean" and self.world > 1) else 1.0
        hp.adam_step(self.params, self.grad, self.m, self.v, lr, self.t, self.beta1, self.beta2, self.eps, scale)

    def apply_adam_fused(self, constraints: Sequence[FusedConstraint], adam: dict) -> bool:
        """Data parallelism with the fused tile kernel (one constraint, padded width 64): forward_backward() was tile kernel +
        tail kernel (sums); behind the all-reduce -- enqueued on the launch stream, no host synchronisation -- ONE launch
        applies Adam from the finished gradient and leaves the bf16 fragments of the updated matrices for the next step
        (ppsci_taylor_step_plan_apply): the per-GPU step is the single-rank step + the collective.  False: nothing done (the
        caller runs its optimizer)."""
        c0 = constraints[0] if len(constraints) == 1 else None
        if (self.world > 1 and c0 is not None and isinstance(c0, FusedConstraint) and self.one_launch_ready(constraints)
                and c0._step_kind == hp.STEP_FUSED_TILE and getattr(c0, "_step_plan", None) is not None
                and c0._step_plan.key == (self.params.data_ptr(), self.grad.data_ptr())):
            c0._step_plan.apply_adam(adam)
            return True
        return False

    def train_step(self, constraints: Sequence[FusedConstraint], lr: float) -> None:
        if self.world == 1 and self.one_launch_ready(constraints):
            self.t += 1
            return self.step_one_launch(constraints, dict(m=self.m, v=self.v, lr=lr, beta1=self.beta1, beta2=self.beta2,
                                                          eps=self.eps, grad_scale=1.0, t=self.t))
        self.forward_backward(constraints)
        self.allreduce()
        scale = (1.0 / self.world) if (self.dp_reduce == "mean" and self.world > 1) else 1.0
        if self.apply_adam_fused(constraints, dict(m=self.m, v=self.v, lr=lr, beta1=self.beta1, beta2=self.beta2, eps=self.eps,
                                                   grad_scale=scale, t=self.t + 1)):
            self.t += 1
            return
        self.optimizer_step(lr)
