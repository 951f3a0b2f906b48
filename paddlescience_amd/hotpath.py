"""Thin host wrappers over the C ABI (include/ppsci_hip.h).  torch tensors are used ONLY as
device-memory handles (`.data_ptr()`), never for arithmetic on the hot path.

The functions here are what `ExpressionSolver.train_forward` / `Solver` (paddlescience_amd/solver)
call instead of the reference's chain of eager paddle ops
(/root/reference/ppsci/utils/expression.py:60-131, ppsci/solver/train.py:112-184).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib as L


def _stream_ptr(t: torch.Tensor):
    """The HIP stream kernels are launched on: torch's current stream of the tensor's device (raw handle; the
    torch.cuda.Stream wrapper costs ~4 us per call, which adds up over ~85 launches of a SPINN step)."""
    if t.is_cuda:
        return C.c_void_p(torch._C._cuda_getCurrentRawStream(t.device.index))
    return None


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _chk_f32(*ts):
    for t in ts:
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous()):
            raise ValueError("hot-path buffers must be contiguous float32 tensors")


def _require_device(t: torch.Tensor):
    """The product path runs on the GPU only; CPU tensors are accepted solely under the emulator."""
    if not t.is_cuda and not L.is_emulated():
        raise RuntimeError("paddlescience_amd hot path needs CUDA(HIP) tensors; there is no CPU fallback")


@dataclass
class StreamSpec:
    """Which derivative streams the kernels carry: first-order along dirs[i] (vectors in raw-input
    space) and pure second-order along dirs[:n2]."""

    dirs: List[List[float]]
    n2: int
    n3: int = 0  # third-order streams along dirs[:n3]  (n3 <= n2)
    n4: int = 0  # fourth-order streams along dirs[:n4] (n4 <= n3)

    @property
    def S(self) -> int:
        return 1 + len(self.dirs) + self.n2 + self.n3 + self.n4


@dataclass
class NetLayout:
    """Shape of a ppsci.arch.MLP as the kernels see it (mlp.py:179-279)."""

    d_raw: int
    n_hidden: int
    width: int
    d_out: int
    activation: str = "tanh"
    skip_connection: bool = False
    embed: Optional[List[int]] = None
    omega: Optional[List[float]] = None
    fourier_half: int = 0  # > 0: hidden layer 0 is the FourierEmbedding as the kernels run it (counted in n_hidden)

    def desc(self, streams: StreamSpec) -> L.MlpDesc:
        return L.make_mlp_desc(self.d_raw, self.n_hidden, self.width, self.d_out, self.activation,
                               self.skip_connection, streams.dirs, streams.n2, self.embed, self.omega,
                               self.fourier_half, getattr(streams, "n3", 0), getattr(streams, "n4", 0))

    @property
    def d0(self) -> int:
        return self.d_raw + (sum(1 for e in self.embed if e == L.EMBED_PERIOD) if self.embed else 0)

    def param_shapes(self) -> List[Tuple[str, Tuple[int, ...]]]:
        out, fin = [], self.d0
        for l in range(self.n_hidden):
            out += [(f"linears.{l}.weight", (fin, self.width)), (f"linears.{l}.bias", (self.width,))]
            fin = self.width
        out += [("last_fc.weight", (fin, self.d_out)), ("last_fc.bias", (self.d_out,))]
        if self.activation in L.PARAM_ACTS:  # kernel layout: the activation parameters follow the last bias
            out += [(f"acts.{l}.param", (self.width,)) for l in range(self.n_hidden)]
        return out

    @property
    def n_params(self) -> int:
        n = 0
        for _, shp in self.param_shapes():
            k = 1
            for s in shp:
                k *= s
            n += k
        return n


def stash_bytes(desc: L.MlpDesc, n: int) -> int:
    return int(L.lib().ppsci_stash_bytes(C.byref(desc), n))


def bwd_partial_rows(desc: L.MlpDesc, n: int) -> int:
    return int(L.lib().ppsci_bwd_partial_rows(C.byref(desc), n))


def bwd_workspace_bytes(desc: L.MlpDesc, n: int) -> int:
    return int(L.lib().ppsci_bwd_workspace_bytes(C.byref(desc), n))


def epilogue_partial_rows(n: int) -> int:
    return int(L.lib().ppsci_epilogue_partial_rows(n))


def taylor_fwd(desc: L.MlpDesc, params: torch.Tensor, inputs: Sequence[torch.Tensor], U: torch.Tensor,
               stash: Optional[torch.Tensor], n: Optional[int] = None) -> None:
    """inputs[j]: [N] array of raw input j -- or, for desc.embed[j] == EMBED_STREAMS, its [S, N] stream block
    (then pass the number of points `n`)."""
    n = inputs[0].numel() if n is None else n
    _require_device(params)
    _chk_f32(params, U, *inputs)
    assert U.numel() == desc.d_out * (1 + desc.n1 + desc.n2 + desc.n3 + desc.n4) * n
    ptrs = L.ptr_array([t.data_ptr() for t in inputs])
    L.check(L.lib().ppsci_taylor_fwd(C.byref(desc), _p(params), n, ptrs, _p(U), _p(stash), _stream_ptr(params)))


def epilogue(edesc: L.EpilogueDesc, n: int, inputs: Sequence[torch.Tensor], U: Optional[torch.Tensor],
             aux: Sequence[torch.Tensor], resid: Optional[torch.Tensor], Ubar: Optional[torch.Tensor],
             loss_partials: torch.Tensor, eq_params: Optional[torch.Tensor] = None,
             eq_param_partials: Optional[torch.Tensor] = None, loss_terms: Optional[torch.Tensor] = None,
             counter: Optional[torch.Tensor] = None) -> None:
    """eq_params / eq_param_partials: learnable equation parameters ([MAX_EPARAM]) and the per-block sums of their
    adjoints ([rows, MAX_EPARAM]) for programs with OP_LD_PARAM (ppsci_epilogue_params).  loss_terms + counter (a zeroed
    int32): the kernel finishes the loss reduction itself (ppsci_epilogue_losses)."""
    _require_device(loss_partials)
    _chk_f32(U, resid, Ubar, loss_partials, eq_params, eq_param_partials, loss_terms, *inputs, *aux)
    ip = L.ptr_array([t.data_ptr() for t in inputs])
    ap = L.ptr_array([t.data_ptr() for t in aux])
    if loss_terms is not None:
        L.check(L.lib().ppsci_epilogue_losses(C.byref(edesc), n, ip, _p(U), ap, _p(resid), _p(Ubar), _p(loss_partials),
                                              _p(eq_params), _p(eq_param_partials), _p(loss_terms), _p(counter),
                                              _stream_ptr(loss_partials)))
        return
    if eq_params is None:
        L.check(L.lib().ppsci_epilogue(C.byref(edesc), n, ip, _p(U), ap, _p(resid), _p(Ubar), _p(loss_partials),
                                       _stream_ptr(loss_partials)))
    else:
        L.check(L.lib().ppsci_epilogue_params(C.byref(edesc), n, ip, _p(U), ap, _p(resid), _p(Ubar),
                                              _p(loss_partials), _p(eq_params), _p(eq_param_partials),
                                              _stream_ptr(loss_partials)))


def taylor_bwd(desc: L.MlpDesc, params: torch.Tensor, inputs: Sequence[torch.Tensor], Ubar: torch.Tensor,
               stash: torch.Tensor, workspace: torch.Tensor, grad_partials: torch.Tensor,
               n: Optional[int] = None) -> None:
    n = inputs[0].numel() if n is None else n
    _require_device(params)
    _chk_f32(params, Ubar, grad_partials, *inputs)
    ptrs = L.ptr_array([t.data_ptr() for t in inputs])
    # (the checked form: the workspace was sized under the kernel-choice knobs of THAT moment, csrc/taylor_api.hip)
    L.check(L.lib().ppsci_taylor_bwd_ws(C.byref(desc), _p(params), n, ptrs, _p(Ubar), _p(stash), _p(workspace),
                                        workspace.numel() * workspace.element_size(), _p(grad_partials), _stream_ptr(params)))


def taylor_step_workspace_bytes(desc: L.MlpDesc, edesc: L.EpilogueDesc, n: int) -> int:
    """0: this network / stream set / program has no one-launch step kernel (ppsci_taylor_step_workspace_bytes)."""
    return int(L.lib().ppsci_taylor_step_workspace_bytes(C.byref(desc), C.byref(edesc), n))


STEP_NONE, STEP_SINGLE_WAVE, STEP_FUSED_TILE = 0, 1, 2


def taylor_step_kind(desc: L.MlpDesc, edesc: L.EpilogueDesc, n: int) -> int:
    """Which kernel ppsci_taylor_step runs for this network / stream set / program: STEP_SINGLE_WAVE (padded width 32; wins
    for batches of a few thousand points only) or STEP_FUSED_TILE (padded width 64: forward -> program -> reverse per tile
    with the stash on the chip; any batch size), STEP_NONE: the separate launches."""
    return int(L.lib().ppsci_taylor_step_kind(C.byref(desc), C.byref(edesc), n))


class StepPlan:
    """ppsci_taylor_step_plan / _run: forward -> epilogue -> reverse -> fixed-order reduction (-> Adam) of one constraint
    in one launch, with the argument block prepared once (every buffer is persistent).  adam: dict(m, v, lr, beta1,
    beta2, eps, grad_scale, t) or None."""

    def __init__(self, desc: L.MlpDesc, edesc: L.EpilogueDesc, params: torch.Tensor, n: int, inputs: Sequence[torch.Tensor],
                 aux: Sequence[torch.Tensor], U: Optional[torch.Tensor], Ubar: Optional[torch.Tensor],
                 resid: Optional[torch.Tensor], stash: Optional[torch.Tensor], workspace: torch.Tensor,
                 loss_terms: torch.Tensor, grad: torch.Tensor):
        _require_device(params)
        _chk_f32(params, U, Ubar, resid, loss_terms, grad, workspace, *inputs, *aux)
        ip = L.ptr_array([t.data_ptr() for t in inputs])
        ap = L.ptr_array([t.data_ptr() for t in aux]) if aux else None
        self._free = L.lib().ppsci_taylor_step_plan_free
        self._run = L.lib().ppsci_taylor_step_run_ex
        self._params = params
        self._frag_token = None  # (global parameter-write count, torch version of `params`) right after this plan's last run
        self.handle = L.lib().ppsci_taylor_step_plan(C.byref(desc), C.byref(edesc), _p(params), n, ip, ap, _p(U), _p(Ubar),
                                                     _p(resid), _p(stash), _p(workspace), workspace.numel() * 4,
                                                     _p(loss_terms), _p(grad))
        if not self.handle:
            raise RuntimeError("ppsci_taylor_step_plan: " + L.lib().ppsci_last_error().decode())
        self.key = (params.data_ptr(), grad.data_ptr())
        self._keep = (params, inputs, aux, U, Ubar, resid, stash, workspace, loss_terms, grad)  # the plan holds raw pointers
        self._dev = params
        self.scales = self._scales(edesc)

    @staticmethod
    def _scales(edesc: L.EpilogueDesc):
        return tuple(edesc.res[k].scale for k in range(edesc.n_res))

    def run(self, edesc: L.EpilogueDesc, accumulate: bool, adam: Optional[dict] = None) -> None:
        sc = self._scales(edesc)
        if sc != self.scales:  # loss re-weighting (GradNorm / NTK, Solver._apply_loss_weights) between steps
            L.check(L.lib().ppsci_taylor_step_plan_set_scales(self.handle, C.byref(edesc)))
            self.scales = sc
        aa = None
        if adam is not None:
            aa = C.byref(L.AdamArgs(adam["m"].data_ptr(), adam["v"].data_ptr(), adam["lr"], adam["beta1"], adam["beta2"],
                                    adam["eps"], adam.get("grad_scale", 1.0), adam["t"]))
        # The tail kernel of a fused-tile step leaves the bf16 fragments of the updated hidden matrices behind; the next run
        # skips the weight-split launch when NOTHING has written the parameters in between: no kernel of this module
        # (param_writes counts adam_step / optim_step / the re-parametrisation kernels / other plans' fused Adam) and no torch
        # operation (the tensor's version counter).  Anything else makes the library split again.
        # Never while a HIP graph is being captured: a captured sequence is replayed after optimizer steps this host code
        # does not see, so the split launch must be part of every captured step.
        capturing = self._params.is_cuda and torch.cuda.is_current_stream_capturing()
        keep = (not capturing and self._frag_token is not None
                and self._frag_token == (_PARAM_WRITES[0], self._params._version))
        L.check(self._run(self.handle, 1 if accumulate else 0, aa, _stream_ptr(self._dev), L.STEP_KEEP_FRAGMENTS if keep else 0))
        if adam is not None:
            note_param_write()
        # a captured launch has not run: whatever replays it later leaves no trace here, so the next eager run splits again
        self._frag_token = None if capturing else (_PARAM_WRITES[0], self._params._version)

    def apply_adam(self, adam: dict) -> None:
        """Data parallelism: Adam from the finished (all-reduced) gradient of the plan + the fragments of the updated hidden
        matrices, one launch (ppsci_taylor_step_plan_apply)."""
        aa = L.AdamArgs(adam["m"].data_ptr(), adam["v"].data_ptr(), adam["lr"], adam["beta1"], adam["beta2"], adam["eps"],
                        adam.get("grad_scale", 1.0), adam["t"])
        L.check(L.lib().ppsci_taylor_step_plan_apply(self.handle, C.byref(aa), _stream_ptr(self._dev)))
        note_param_write()
        capturing = self._params.is_cuda and torch.cuda.is_current_stream_capturing()
        self._frag_token = None if capturing else (_PARAM_WRITES[0], self._params._version)

    @property
    def static_program(self) -> str:
        """Name of the compile-time table (csrc/epi_static_programs.h) this plan's residual program runs as; "" = the VM."""
        nm = C.c_char_p()
        return (nm.value or b"").decode() if L.lib().ppsci_taylor_step_plan_static(self.handle, C.byref(nm)) > 0 else ""

    def run_main(self) -> None:
        """Measurement: the main kernel of the step alone (ppsci_taylor_step_run_main)."""
        L.check(L.lib().ppsci_taylor_step_run_main(self.handle, _stream_ptr(self._dev)))

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            self._free(h)


_PARAM_WRITES = [0]


def note_param_write() -> None:
    """Called by everything in this module that lets a kernel write a parameter buffer (torch operations are seen through
    the tensors' version counters): StepPlan.run compares the count to decide whether its weight fragments are current."""
    _PARAM_WRITES[0] += 1


def dense_matvec(M: torch.Tensor, x: torch.Tensor, y: torch.Tensor, alpha: float = 1.0, transpose: bool = False,
                 rowscale: Optional[torch.Tensor] = None) -> None:
    """ppsci_dense_matvec: y[:rows] = alpha M x (transpose=False) or y[:cols] = alpha M^T (x * rowscale) (transpose=True)."""
    _require_device(y)
    _chk_f32(M, x, y, rowscale)
    rows, cols = M.shape
    assert x.numel() >= (rows if transpose else cols) and y.numel() >= (cols if transpose else rows)
    L.check(L.lib().ppsci_dense_matvec(rows, cols, _p(M), _p(x), _p(rowscale), float(alpha), 1 if transpose else 0, _p(y),
                                       _stream_ptr(y)))


def reduce_rows(partials: torch.Tensor, rows: int, cols: int, out: torch.Tensor, accumulate: bool) -> None:
    _require_device(out)
    _chk_f32(partials, out)
    L.check(L.lib().ppsci_reduce_rows(_p(partials), rows, cols, _p(out), 1 if accumulate else 0, _stream_ptr(out)))


def reduce_rows_multi_adam(segs: Sequence[tuple], params: torch.Tensor, grad: torch.Tensor, m: torch.Tensor, v: torch.Tensor,
                           lr: float, step_t: int, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8,
                           grad_scale: float = 1.0) -> None:
    """ppsci_reduce_rows_multi_adam: segs = (source pointer, destination pointer, rows, cols) -- the row reductions that end a
    backward pass and the Adam update of the flat parameter buffer in ONE launch (<= 16 segments)."""
    _require_device(params)
    _chk_f32(params, grad, m, v)
    note_param_write()
    arr = (L.ReduceSeg * len(segs))()
    for k, (src, dst, rows, cols) in enumerate(segs):
        arr[k].partials, arr[k].out, arr[k].rows, arr[k].cols, arr[k].accumulate = src, dst, rows, cols, 0
    L.check(L.lib().ppsci_reduce_rows_multi_adam(len(segs), arr, params.numel(), _p(params), _p(grad), _p(m), _p(v), lr, beta1,
                                                 beta2, eps, step_t, grad_scale, _stream_ptr(params)))


def adam_step(params: torch.Tensor, grad: torch.Tensor, m: torch.Tensor, v: torch.Tensor, lr: float, step_t: int,
              beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8, grad_scale: float = 1.0) -> None:
    _require_device(params)
    _chk_f32(params, grad, m, v)
    note_param_write()
    L.check(L.lib().ppsci_adam_step(params.numel(), _p(params), _p(grad), _p(m), _p(v), lr, beta1, beta2, eps,
                                    step_t, grad_scale, _stream_ptr(params)))


OPT_SGD, OPT_MOMENTUM, OPT_RMSPROP, OPT_ADAMW = 0, 1, 2, 3
LOSS_MSE, LOSS_ABS, LOSS_SQRTABS, LOSS_ABSREL, LOSS_LINEAR = 0, 1, 2, 3, 4


def optim_step(kind: int, params: torch.Tensor, grad: torch.Tensor, states: Sequence[Optional[torch.Tensor]],
               hyper: Sequence[float], flag: bool = False) -> None:
    """ppsci_optim_step: hyper = [lr, grad_scale, l2, a, b, c, d] (include/ppsci_hip.h)."""
    _require_device(params)
    _chk_f32(params, grad, *[t for t in states if t is not None])
    note_param_write()
    st = list(states) + [None] * (3 - len(states))
    hy = (C.c_float * 7)(*[float(v) for v in list(hyper) + [0.0] * (7 - len(hyper))])
    L.check(L.lib().ppsci_optim_step(kind, params.numel(), _p(params), _p(grad), _p(st[0]), _p(st[1]), _p(st[2]), hy,
                                     1 if flag else 0, _stream_ptr(params)))


def causal_weights(n_chunks: int, tol: float, value: torch.Tensor, label: Optional[torch.Tensor],
                   weight: Optional[torch.Tensor], area: Optional[torch.Tensor], chunk_scratch: torch.Tensor,
                   cw: torch.Tensor) -> None:
    """ppsci_causal_weights: per-point causal factor of CausalMSELoss for one loss key (mse.py:158-177)."""
    _require_device(cw)
    _chk_f32(*[t for t in (value, label, weight, area, chunk_scratch, cw) if t is not None])
    L.check(L.lib().ppsci_causal_weights(value.numel(), n_chunks, float(tol), _p(value), _p(label), _p(weight),
                                         _p(area), _p(chunk_scratch), _p(cw), _stream_ptr(cw)))


def linear_materialize(kind: int, fin: int, fout: int, v: torch.Tensor, g: Optional[torch.Tensor],
                       b: Optional[torch.Tensor], W: torch.Tensor, b_out: Optional[torch.Tensor]) -> None:
    """ppsci_linear_materialize: trainable tensors of one layer -> its slice of the kernel parameter buffer."""
    _require_device(W)
    _chk_f32(*[t for t in (v, g, b, W, b_out) if t is not None])
    note_param_write()
    L.check(L.lib().ppsci_linear_materialize(kind, fin, fout, _p(v), _p(g), _p(b), _p(W), _p(b_out), _stream_ptr(W)))


def linear_multi(jobs, back: bool, like: torch.Tensor) -> None:
    """ppsci_linear_multi: the materialize (back = False) or pullback (True) of several layers, 16 per launch.  jobs: the
    argument tuples of linear_materialize (kind, fin, fout, v, g, b, W, b_out) / linear_pullback (kind, fin, fout, v, g, gW,
    gb, gv, gg, gb_out)."""
    _require_device(like)
    if not back:
        note_param_write()
    ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    for i0 in range(0, len(jobs), 16):
        batch = jobs[i0:i0 + 16]
        arr = (L.LinearJob * len(batch))()
        for k, j in enumerate(batch):
            _chk_f32(*[t for t in j[3:] if t is not None])
            arr[k].kind, arr[k].fin, arr[k].fout = j[0], j[1], j[2]
            if back:
                arr[k].v, arr[k].g, arr[k].gW, arr[k].gb, arr[k].gv, arr[k].gg, arr[k].gb_out = [ptr(t) for t in j[3:10]]
            else:
                arr[k].v, arr[k].g, arr[k].b, arr[k].W, arr[k].b_out = [ptr(t) for t in j[3:8]]
        L.check(L.lib().ppsci_linear_multi(len(batch), arr, 1 if back else 0, _stream_ptr(like)))


def linear_pad(src_dims, dst_dims, v: torch.Tensor, b: torch.Tensor, W: torch.Tensor, b_out: torch.Tensor) -> None:
    """ppsci_linear_pad: trainable [fin_s, fout_s] block (+ bias) -> zero-filled kernel-layout [fin_d, fout_d] slice."""
    _require_device(W)
    _chk_f32(v, b, W, b_out)
    note_param_write()
    L.check(L.lib().ppsci_linear_pad(src_dims[0], src_dims[1], dst_dims[0], dst_dims[1], _p(v), _p(b), _p(W), _p(b_out),
                                     _stream_ptr(W)))


def linear_unpad(src_dims, dst_dims, gW: torch.Tensor, gb: torch.Tensor, gv: torch.Tensor, gb_out: torch.Tensor) -> None:
    """ppsci_linear_unpad: the trainable block of the kernel-layout gradient."""
    _require_device(gW)
    _chk_f32(gW, gb, gv, gb_out)
    L.check(L.lib().ppsci_linear_unpad(src_dims[0], src_dims[1], dst_dims[0], dst_dims[1], _p(gW), _p(gb), _p(gv),
                                       _p(gb_out), _stream_ptr(gW)))


def linear_pullback(kind: int, fin: int, fout: int, v: Optional[torch.Tensor], g: Optional[torch.Tensor],
                    gW: torch.Tensor, gb: Optional[torch.Tensor], gv: torch.Tensor, gg: Optional[torch.Tensor],
                    gb_out: Optional[torch.Tensor]) -> None:
    """ppsci_linear_pullback: gradient of the kernel-layout slice -> gradients of the trainable tensors."""
    _require_device(gW)
    _chk_f32(*[t for t in (v, g, gW, gb, gv, gg, gb_out) if t is not None])
    L.check(L.lib().ppsci_linear_pullback(kind, fin, fout, _p(v), _p(g), _p(gW), _p(gb), _p(gv), _p(gg), _p(gb_out),
                                          _stream_ptr(gW)))


# ----------------------------------------------------------------------------- epilogue builder
class Program:
    """Builds a ppsci_epilogue_desc in SSA form with common-subexpression reuse for loads/consts."""

    def __init__(self, n_streams: int, n_in: int):
        self.instrs: List[Tuple[int, int, int, float]] = []
        self.res: List[Tuple[int, int, int, int, float]] = []
        self.n_streams, self.n_in = n_streams, n_in
        self.n_aux = 0
        self._memo: Dict[Tuple, int] = {}

    def _emit(self, op, a=0, b=0, c=0.0, memo=True) -> int:
        key = (op, a, b, float(c))
        if memo and key in self._memo:
            return self._memo[key]
        if len(self.instrs) >= L.MAX_PROG:
            raise NotImplementedError(f"epilogue program longer than {L.MAX_PROG} instructions")
        self.instrs.append((op, a, b, float(c)))
        idx = len(self.instrs) - 1
        if memo:
            self._memo[key] = idx
        return idx

    def ld_in(self, j: int) -> int:
        return self._emit(L.OP_LD_IN, j)

    def ld_u(self, q: int) -> int:
        return self._emit(L.OP_LD_U, q)

    def ld_aux(self, k: int) -> int:
        self.n_aux = max(self.n_aux, k + 1)
        return self._emit(L.OP_LD_AUX, k)

    def ld_param(self, slot: int) -> int:
        return self._emit(L.OP_LD_PARAM, slot)

    def const(self, c: float) -> int:
        import numpy as np

        return self._emit(L.OP_CONST, 0, 0, float(np.float32(c)))

    def op(self, op: int, a: int, b: int = 0) -> int:
        return self._emit(op, a, b, 0.0)

    def residual(self, value: int, label: int = -1, weight: int = -1, area: int = -1, scale: float = 1.0,
                 kind: int = 0, scale_param: int = 0) -> int:
        if len(self.res) >= L.MAX_RES:
            raise NotImplementedError(f"more than {L.MAX_RES} loss terms in one epilogue")
        for k in (label, weight, area):
            if k >= 0:
                self.n_aux = max(self.n_aux, k + 1)
        self.res.append((value, label, weight, area, float(scale), int(kind), int(scale_param)))
        return len(self.res) - 1

    def build(self) -> L.EpilogueDesc:
        e = L.EpilogueDesc()
        e.n_instr, e.n_res = len(self.instrs), len(self.res)
        e.n_streams, e.n_in, e.n_aux = self.n_streams, self.n_in, self.n_aux
        for i, (op, a, b, c) in enumerate(self.instrs):
            e.prog[i].op, e.prog[i].a, e.prog[i].b, e.prog[i].c = op, a, b, c
        for k, (v, lab, w, ar, sc, kind, sp) in enumerate(self.res):
            r = e.res[k]
            r.value, r.label, r.weight, r.area, r.scale, r.kind, r.scale_param = v, lab, w, ar, sc, kind, sp
        return e
