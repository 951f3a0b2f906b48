"""Training engine for separable PINNs (ppsci.arch.SPINN): per constraint
  3 x modmlp_fwd  ->  spinn_grid_fwd (residual + MSE + adjoint)  ->  spinn_grid_bwd  ->  3 x modmlp_bwd,
then one fixed-order reduction per branch into the flat gradient, one all-reduce, one fused Adam.
Counterpart of the generic engine.py for BASELINE config 5 (examples/spinn/helmholtz3d.py of the
reference: one PDE constraint on the nc^3 grid + six boundary faces).  Data parallelism shards the
points of one axis rank-strided (each rank owns an [nx/W, ny, nz] slab of the interior grid); branch nets are replicated."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib as L
from . import hotpath as hp
import os

from .engine import StepGraph, run_on_streams
from .hotpath import _p, _stream_ptr


class SpinnConstraint:
    def __init__(self, name: str, model, coeffs: np.ndarray, label_key: str, scale_fn, device, world: int = 1,
                 rank: int = 0):
        self.name, self.model, self.label_key = name, model, label_key
        self.coeffs = np.asarray(coeffs, dtype=np.float64)
        self.scale_fn = scale_fn  # total_points -> loss scale
        self.device, self.world, self.rank = device, world, rank
        self.shape = None
        self._last_ids = [None] * 4
        self._uploaded = [None] * 4
        self._version = 0  # bumped whenever the device buffers are re-allocated (captured graphs hold addresses)

    def _alloc(self, shape):
        m, dev = self.model, self.device
        self.shape = tuple(shape)
        self._version += 1
        self._last_ids, self._uploaded = [None] * 4, [None] * 4
        nx, ny, nz = shape
        R, P = m.spec.R, m.branch_params
        f32 = dict(dtype=torch.float32, device=dev)
        self.x = [torch.zeros(n, **f32) for n in shape]
        self.F = [torch.zeros((3, n, R), **f32) for n in shape]
        self.Fbar = [torch.zeros((3, n, R), **f32) for n in shape]
        self.stash = [torch.zeros(int(L.lib().ppsci_modmlp_stash_floats(C.byref(m.spec.desc), n)), **f32) for n in shape]
        # per-point gradient rows of the three branch nets: with equal point counts they interleave as [n][3][P], so that one
        # reduce_rows(n, 3P) call sums all three into the flat gradient (branch b's parameters are grad[bP:(b+1)P])
        # (rows: one per 16-point tile on the MFMA tile kernel of the reverse sweep, one per point otherwise)
        grows = [int(L.lib().ppsci_modmlp_bwd_rows(C.byref(m.spec.desc), n)) for n in shape]
        self.gjoint = len(set(shape)) == 1
        if self.gjoint:
            self.gpart_all = torch.zeros((grows[0], 3 * P), **f32)
            self.gpart = [self.gpart_all.view(-1)[b * P:] for b in range(3)]  # row 0 of branch b; rows are 3P apart
        else:
            self.gpart = [torch.zeros((r, P), **f32) for r in grows]
        total = nx * ny * nz
        self.label = torch.zeros(total, **f32)
        self.gadj = torch.zeros(total, **f32)
        self.desc = L.SpinnGridDesc()
        self.desc.n[0], self.desc.n[1], self.desc.n[2] = nx, ny, nz
        self.desc.rank = R
        self.desc.cu, self.desc.cxx, self.desc.cyy, self.desc.czz = (float(c) for c in self.coeffs)
        self.lrows = int(L.lib().ppsci_spinn_grid_partial_rows(C.byref(self.desc)))
        self.lpart = torch.zeros(self.lrows, **f32)
        self.bscratch = torch.zeros(max(4, int(L.lib().ppsci_spinn_grid_bwd_scratch_floats(C.byref(self.desc)))), **f32)
        self.loss_term = torch.zeros(1, **f32)

    def bind(self, input: Dict[str, np.ndarray], label: Dict[str, np.ndarray]):
        keys = self.model.input_keys
        arrs = [np.asarray(input[k], dtype=np.float32).reshape(-1) for k in keys]
        lab = np.asarray(label[self.label_key], dtype=np.float32)
        gshape = tuple(a.shape[0] for a in arrs)
        rep = 1.0
        if self.world > 1:
            # rank-strided slab along the first axis that has at least `world` points (x for the interior grid,
            # y or z for the boundary faces whose x-axis is a single point); a grid smaller than that on every
            # axis is evaluated by all ranks and weighted 1/world so that the SUM all-reduce stays exact
            ax = next((i for i, n in enumerate(gshape) if n >= self.world), None)
            if ax is None:
                rep = 1.0 / self.world
            else:
                arrs[ax] = arrs[ax][self.rank::self.world]
                idx = [slice(None)] * 3
                idx[ax] = slice(self.rank, None, self.world)
                lab = lab.reshape(gshape)[tuple(idx)]
        shape = tuple(a.shape[0] for a in arrs)
        if self.shape != shape:
            self._alloc(shape)
        # The reference re-uploads every iteration.  Here an array is uploaded only when it changed: the same
        # object with the same sampled values is taken as unchanged (in-place edits that keep first / middle /
        # last are not seen); a new object is compared in full with a private copy of what was uploaded last.
        def _fp(a):
            f = np.asarray(a).reshape(-1)
            n = f.shape[0]
            return (id(a), n, float(f[0]), float(f[n // 2]), float(f[n - 1])) if n else (id(a), 0)

        srcs = [input[k] for k in keys] + [label[self.label_key]]
        vals = arrs + [lab.reshape(-1)]
        dsts = list(self.x) + [self.label]
        for j, (src, val, dst) in enumerate(zip(srcs, vals, dsts)):
            fp = _fp(src)
            if fp == self._last_ids[j]:
                continue
            self._last_ids[j] = fp
            val = np.ascontiguousarray(val)
            old = self._uploaded[j]
            if old is not None and old.shape == val.shape and np.array_equal(old, val):
                continue
            self._uploaded[j] = val.copy()
            dst.copy_(torch.from_numpy(val))
        total_global = gshape[0] * gshape[1] * gshape[2]
        self.desc.scale = float(self.scale_fn(total_global)) * rep

    def forward(self, train: bool, reduce_loss: bool = True):
        """reduce_loss=False: the per-workgroup loss rows stay in `lpart`; the caller sums them (SpinnEngine: together with the
        gradient rows, one launch)."""
        m, lib = self.model, L.lib()
        vp = lambda ts: (C.c_void_p * 3)(*[t.data_ptr() for t in ts])  # noqa: E731
        L.check(lib.ppsci_modmlp_fwd_batch(C.byref(m.spec.desc), 3, vp([m.branch(b) for b in range(3)]),
                                           (C.c_int64 * 3)(*[t.numel() for t in self.x]), vp(self.x), vp(self.F),
                                           vp(self.stash) if train else None, _stream_ptr(self.x[0])))
        L.check(lib.ppsci_spinn_grid_fwd(C.byref(self.desc), _p(self.F[0]), _p(self.F[1]), _p(self.F[2]), _p(self.label), None,
                                         _p(self.gadj) if train else None, _p(self.lpart), _stream_ptr(self.label)))
        if reduce_loss:
            hp.reduce_rows(self.lpart, self.lrows, 1, self.loss_term, False)

    def backward(self):
        m, lib = self.model, L.lib()
        vp = lambda ts: (C.c_void_p * 3)(*[t.data_ptr() for t in ts])  # noqa: E731
        if (os.environ.get("PPSCI_SPINN_PARTS", "0") == "1"
                and lib.ppsci_modmlp_bwd_parts_supported(C.byref(m.spec.desc), C.byref(self.desc))):
            # the grid kernel leaves its per-group partials of dL/dF in the scratch; the branch nets' tile kernel sums them on load.
            # OFF by default -- measured on MI355X (128^3, 3 x 128 points): 0.0983 ms per step with it, 0.0914 without: the 24
            # tile workgroups read 128 KB of partials each in front of their latency chain, which costs more than the ~5 us launch
            # of spinn_fbar_sum_kernel (384 workgroups) it saves.  Kept as a tested option for larger point counts.
            L.check(lib.ppsci_spinn_grid_bwd(C.byref(self.desc), _p(self.F[0]), _p(self.F[1]), _p(self.F[2]), _p(self.gadj),
                                             _p(self.bscratch), None, None, None, _stream_ptr(self.gadj)))
            L.check(lib.ppsci_modmlp_bwd_batch_parts(C.byref(m.spec.desc), C.byref(self.desc), vp([m.branch(b) for b in range(3)]),
                                                     vp(self.x), _p(self.bscratch), vp(self.stash), vp(self.gpart),
                                                     3 * m.branch_params if self.gjoint else 0, _stream_ptr(self.x[0])))
            return
        L.check(lib.ppsci_spinn_grid_bwd(C.byref(self.desc), _p(self.F[0]), _p(self.F[1]), _p(self.F[2]), _p(self.gadj),
                                         _p(self.bscratch), _p(self.Fbar[0]), _p(self.Fbar[1]), _p(self.Fbar[2]),
                                         _stream_ptr(self.gadj)))
        L.check(lib.ppsci_modmlp_bwd_batch(C.byref(m.spec.desc), 3, vp([m.branch(b) for b in range(3)]),
                                           (C.c_int64 * 3)(*[t.numel() for t in self.x]), vp(self.x), vp(self.Fbar),
                                           vp(self.stash), vp(self.gpart), 3 * m.branch_params if self.gjoint else 0,
                                           _stream_ptr(self.x[0])))

    def loss(self) -> float:
        return float(self.loss_term.cpu()[0])


class SpinnEngine:
    def __init__(self, model):
        self.model = model
        self.grad = torch.zeros_like(model.flat_params)
        self.world = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
        self.dp_reduce = "sum"
        self._step_graph = StepGraph(self.grad.is_cuda and os.environ.get("PPSCI_HIP_GRAPH", "1") != "0")
        self.multi_stream = os.environ.get("PPSCI_MULTI_STREAM", "1") != "0"
        self._streams: list = []

    def _segments(self, constraints: Sequence[SpinnConstraint]):
        """The row reductions that end the step of ONE constraint, as (source, destination, rows, cols) pointers: the loss
        rows and the gradient rows of the three branch nets."""
        P = self.model.branch_params
        c0 = constraints[0]
        segs = [(c0.lpart.data_ptr(), c0.loss_term.data_ptr(), c0.lrows, 1)]
        if c0.gjoint:
            segs.append((c0.gpart_all.data_ptr(), self.grad.data_ptr(), c0.gpart_all.shape[0], 3 * P))
        else:
            segs += [(c0.gpart[b].data_ptr(), self.grad[b * P:(b + 1) * P].data_ptr(), c0.gpart[b].shape[0], P) for b in range(3)]
        return segs

    def forward_backward_deferred(self, constraints: Sequence[SpinnConstraint]):
        """forward_backward WITHOUT the reduction launch behind it, for a caller that sums the rows and applies Adam in one
        launch (hp.reduce_rows_multi_adam; solver.Solver._reduce_and_adam_in_one_launch): returns the pending reductions.
        None: several constraints share the gradient (their rows accumulate: the reductions stay separate launches) -- the
        step was run in full."""
        if len(constraints) != 1:
            self.forward_backward(constraints)
            return None
        key = tuple((id(c), c._version, float(c.desc.scale)) for c in constraints) + ("deferred",)
        self._step_graph.run(key, lambda: self._forward_backward_eager(constraints, reduce=False))
        return self._segments(constraints)

    def flush_deferred(self, segs) -> None:
        arr = (L.ReduceSeg * len(segs))()
        for k, (src, dst, rows, cols) in enumerate(segs):
            arr[k].partials, arr[k].out, arr[k].rows, arr[k].cols, arr[k].accumulate = src, dst, rows, cols, 0
        L.check(L.lib().ppsci_reduce_rows_multi(len(segs), arr, _stream_ptr(self.grad)))

    def _forward_backward_eager(self, constraints: Sequence[SpinnConstraint], reduce: bool = True):
        P = self.model.branch_params
        if self.multi_stream and len(constraints) > 1 and self.grad.is_cuda:
            # the PDE grid and the six boundary faces are independent until their gradients are summed
            run_on_streams(self._streams, [(lambda c=c: (c.forward(True, False), c.backward())) for c in constraints])
        else:
            for c in constraints:
                c.forward(True, False)
                c.backward()
        if not reduce:
            return
        # ONE launch (ppsci_reduce_rows_multi) sums every constraint's loss rows and the FIRST constraint's gradient rows -- segments
        # with different destinations; the other constraints' gradient rows are added behind it, one launch each, in order (they
        # share a destination).  Helmholtz3D's single PDE constraint: two launches (this one + Adam) instead of three.
        segs = [(c.lpart, c.loss_term, c.lrows, 1) for c in constraints]
        c0 = constraints[0]
        if c0.gjoint:
            segs.append((c0.gpart_all, self.grad[:3 * P], c0.gpart_all.shape[0], 3 * P))
        else:
            segs += [(c0.gpart[b], self.grad[b * P:(b + 1) * P], c0.gpart[b].shape[0], P) for b in range(3)]
        for i0 in range(0, len(segs), 16):
            batch = segs[i0:i0 + 16]
            arr = (L.ReduceSeg * len(batch))()
            for k, (src, dst, rows, cols) in enumerate(batch):
                arr[k].partials, arr[k].out, arr[k].rows, arr[k].cols, arr[k].accumulate = src.data_ptr(), dst.data_ptr(), rows, cols, 0
            L.check(L.lib().ppsci_reduce_rows_multi(len(batch), arr, _stream_ptr(self.grad)))
        for c in constraints[1:]:
            if c.gjoint:
                hp.reduce_rows(c.gpart_all, c.gpart_all.shape[0], 3 * P, self.grad[:3 * P], True)
                continue
            for b in range(3):
                hp.reduce_rows(c.gpart[b], c.gpart[b].shape[0], P, self.grad[b * P:(b + 1) * P], True)

    def forward_backward(self, constraints: Sequence[SpinnConstraint]):
        # A step is ~12 launches per constraint of a few microseconds each (seven constraints in the reference's
        # Helmholtz3D example): launch-bound from Python, so the whole sequence is one replayed HIP graph.  The
        # residual scale and the grid shape are captured by value, hence part of the key.
        key = tuple((id(c), c._version, float(c.desc.scale)) for c in constraints)
        self._step_graph.run(key, lambda: self._forward_backward_eager(constraints))

    def invalidate_graphs(self) -> None:
        self._step_graph.clear()

    def allreduce(self):
        if self.world > 1:
            torch.distributed.all_reduce(self.grad, op=torch.distributed.ReduceOp.SUM)
