"""Native forward + backward of ppsci.arch.FNONet / TFNO2dNet: every non-FFT operation of a training step is one of
this framework's HIP kernels (csrc/fno.hip, csrc/spectral_conv.hip) and the backward pass is written out by hand -- no
autograd graph, no rocBLAS / MIOpen kernels.  What the reference runs per step (tfnonet.py:179-193):

    x = lifting(x)                                   fno_block.MLP: conv1x1 -> GELU -> conv1x1        (fno_block.py:263-320)
    for l: x = act( norm(SpectralConv_l(x)) + skip_l(x) )   forward_with_postactivation          (fno_block.py:1191-1220)
    y = projection(x)                                fno_block.MLP

    SpectralConv (fno_block.py:707-796): rfftn -> per-mode complex channel contraction on the kept modes -> irfftn, + bias

The FFTs are raw hipFFT executions on the launch stream (ppsci_fft2d_r2c / ppsci_fft2d_c2r, csrc/fft.hip: unscaled, no
clones, no normalisation kernels; they are linear, so their adjoints are FFTs again, see `ppsci_spectral_conv2d_bwd_real`);
everything else is ppsci_pw_conv / ppsci_pw_conv_wgrad / ppsci_fno_tail_* / ppsci_spectral_conv2d_* / ppsci_reduce_rows.
The 1/(H*W) of the rfftn / irfftn pair -- the same for every `fft_norm` -- is folded into the spectral contraction.  Gradients are written straight into the views of `model.flat_grad`.

Covered configuration: 2-D, dense spectral weights, GELU, post-activation blocks (optionally with the tanh stabilizer),
linear or identity skip, GroupNorm(1 group) or no norm, DomainPadding, two-layer (or one-layer) lifting and a two-layer
projection.  This executor is the ONLY implementation of the network: training, evaluation and prediction all run through
it (arch/fno.py holds parameters, not operations); what it does not cover is refused when the model is built."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional

import torch

from . import _lib as L
from . import hotpath as hp
from .arch import fno as fno_arch
from .hotpath import _p, _stream_ptr


def supports(model) -> Optional[str]:
    """None when the kernels cover `model`, else the reason they do not (the constructors of arch/fno.py already refuse
    the options without a kernel: other norms / activations / skips, preactivation, use_mlp, 1-D / 3-D)."""
    from .arch import fno

    if not isinstance(model, fno.FNONet):
        return "not an FNONet"
    if model.lifting.n_layers not in (1, 2) or model.projection.n_layers != 2:
        return "lifting / projection depth"
    return None


def _pw_conv(B, cin, cout, P, x, W, out, bias=None, zmul=None, act=None, transpose=False, accumulate=False, xv=None, zv=None):
    """xv / zv: L.PwVirtual -- x / zmul as functions of a stored tensor, evaluated on load (include/ppsci_hip.h)."""
    L.check(L.lib().ppsci_pw_conv_v(B, cin, cout, P, _p(x), C.byref(xv) if xv is not None else None, _p(W),
                                    1 if transpose else 0, _p(bias), _p(zmul), C.byref(zv) if zv is not None else None,
                                    1 if accumulate else 0, _p(out), _p(act), _stream_ptr(out)))


def _virtual(mode, x0=None, W0=None, b0=None):
    v = L.PwVirtual()
    v.mode = mode
    if mode == 2:
        v.K0 = W0.shape[1]
        v.x0, v.W0, v.b0 = x0.data_ptr(), W0.data_ptr(), (b0.data_ptr() if b0 is not None else None)
    return v


class FnoNative:
    def __init__(self, model):
        why = supports(model)
        if why is not None:
            raise NotImplementedError(f"native FNO path: {why}")
        self.m = model
        self.shape = None
        # One buffer set per input shape (batch, H, W), kept alive: training, evaluation and prediction share this executor
        # (arch/fno.py), the reference's TFNO config trains at 16 x 16 and evaluates at 32 x 32 with eval_during_train, a ragged
        # last eval batch changes the batch size -- and the operator engine REPLAYS a captured HIP graph of the training step
        # that holds the training buffers' addresses.  Handing those back to the caching allocator when another shape comes by
        # would leave the graph writing through stale pointers.  At most `max_sets` sets are kept (least recently used out);
        # dropping one bumps `generation`, which is part of the engine's graph key, so no captured step outlives its buffers.
        self._sets = {}
        self.max_sets = 8
        self.generation = 0
        # A TFNO step is ~57 kernels of 5-17 us, most of them a dependent chain -- but not all: a block's skip convolution does
        # not depend on its spectral branch (forward), and no weight gradient is needed before the end of the backward pass.
        # PPSCI_FNO_SIDE_STREAM=1 puts those launches onto a second HIP stream, forked from and joined back into the launch stream
        # (inside the engine's captured graph: parallel branches).  Measured on MI355X (round 5, batch 16, 64 x 64): 0.766 ms
        # per step against 0.677 ms in one stream -- the ~15 extra fork / join edges of the replayed graph cost more than the
        # overlapped 5-12 us kernels give back -- so it is OFF by default (kept as a tested knob).
        self.use_side = os.environ.get("PPSCI_FNO_SIDE_STREAM", "0") == "1" and model.flat_params.is_cuda
        self._side = None

    # ------------------------------------------------------------------ second stream
    def _fork(self):
        """Context manager: launches inside go to the side stream, ordered behind everything issued so far on the launch stream."""
        import contextlib

        if not self.use_side:
            return contextlib.nullcontext()
        if self._side is None:
            self._side = torch.cuda.Stream()
        self._side.wait_stream(torch.cuda.current_stream())
        return torch.cuda.stream(self._side)

    def _join(self) -> None:
        """The launch stream waits for the side stream's work."""
        if self.use_side and self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)

    # ------------------------------------------------------------------ buffers
    def _switch(self, B: int, H: int, W: int) -> None:
        keep = ("m", "shape", "_sets", "max_sets", "generation", "use_side", "_side", "defer_wgrad_sums")
        return self._switch_(B, H, W, keep)

    def _switch_(self, B: int, H: int, W: int, keep) -> None:
        """Make the buffer set of input shape (B, H, W) the current one (allocating it on first use); the previous set stays
        alive under its own key."""
        if self.shape is not None:
            self._sets[self.shape] = {k: v for k, v in self.__dict__.items() if k not in keep}
        self.shape = None  # no current set until the new one is complete: a failed allocation leaves the executor usable
        for k in [k for k in self.__dict__ if k not in keep]:
            del self.__dict__[k]
        key = (B, H, W)
        if key in self._sets:
            self.__dict__.update(self._sets.pop(key))  # (re-inserted when it is switched away from: most recently used last)
            self.shape = key
            return
        while len(self._sets) >= self.max_sets:
            self._sets.pop(next(iter(self._sets)))
            self.generation += 1
        self._alloc(B, H, W)

    def _alloc(self, B: int, H: int, W: int) -> None:
        m = self.m
        dev = m.flat_params.device
        f = dict(dtype=torch.float32, device=dev)
        Ch, nl = m.hidden_channels, m.n_layers
        # DomainPadding (fno_block.py:19-140): the blocks run on the padded [Hp, Wp] planes, lifting / projection on [H0, W0]
        H0, W0 = H, W
        ah, aw, self.oh, self.ow = m.padding_of(H0, W0) if hasattr(m, "padding_of") else (0, 0, 0, 0)
        H, W = H0 + ah, W0 + aw
        self.padded = bool(ah or aw)
        P, P0 = H * W, H0 * W0
        self.shape = (B, H0, W0)
        self.hw, self.hw0 = (H, W), (H0, W0)
        self.P, self.P0 = P, P0
        lift, proj = m.lifting.fcs, m.projection.fcs
        self.c_lift = lift[0].out_channels if len(lift) == 2 else 0
        self.c_proj = proj[0].out_channels
        # The hidden tensors of the two channel MLPs are 8 x a block tensor (256 channels): the lifting layer's
        # GELU(W1 x + b1) is recomputed from its <= 4 input channels wherever it is an operand (never stored); of the
        # projection's only the pre-activation z2 is stored, GELU(z2) is applied on load
        self.lift_virtual = bool(self.c_lift) and m.in_channels <= 4
        if self.c_lift and not self.lift_virtual:
            self.z1, self.a1 = torch.empty((B, self.c_lift, P0), **f), torch.empty((B, self.c_lift, P0), **f)
        if self.padded:
            self.x0u = torch.empty((B, Ch, P0), **f)   # lifting output before padding
            self.xou = torch.empty((B, Ch, P0), **f)   # blocks' output after unpadding
            self.gpad = torch.empty((B, Ch, P), **f)   # padded gradient of the blocks' output
        self.x = [torch.empty((B, Ch, P), **f) for _ in range(nl + 1)]          # block inputs, x[nl] = blocks' output
        self.stab = m.fno_blocks.stabilizer == "tanh"
        if self.stab:
            self.xs = [torch.empty((B, Ch, P), **f) for _ in range(nl)]        # tanh(x_l): the spectral branch's input
        self.s = torch.empty((B, Ch, P), **f)                                    # skip branch of the current block
        self.t = [torch.empty((B, Ch, P), **f) for _ in range(nl)]              # pre-activations
        Wf = W // 2 + 1
        self.v = [torch.empty((B, Ch, P), **f) for _ in range(nl)]               # spectral outputs (irfftn results)
        mx, my = m.fno_blocks.convs[0].n_modes
        # SFNONet: the spherical-harmonic pair instead of the FFT pair (csrc/sht.hip; tables per plane shape from arch/sht_tables.py)
        self.sht = getattr(m, "spectral", "fft") == "sht"
        if self.sht:
            from .arch import sht_tables

            if not L.lib().ppsci_sht_supported(H, W, mx, my):
                raise NotImplementedError(f"SFNONet: a {H} x {W} plane with {mx} x {my} coefficients does not fit the transform kernels")
            tw, ta, tb = sht_tables.tables(H, W, mx, my, m.sht_grid, m.sht_norm)
            self.sht_tw = torch.tensor(tw, **f)
            # (each table in the storage order of the kernel that reads it: forward AND as the other transform's adjoint)
            (self.sht_a_an, self.sht_a_sy), (self.sht_b_an, self.sht_b_sy) = (
                tuple(torch.tensor(t, **f) for t in sht_tables.kernel_layouts(T)) for T in (ta, tb))
        # the transforms on the kept modes only (two small DFTs per plane in LDS, spectra of mx x my numbers) when a plane
        # fits LDS; hipFFT on the full spectrum otherwise.  PPSCI_FNO_FULL_FFT=1 forces the library path (tests, timing)
        self.kept = (not self.sht and bool(L.lib().ppsci_dft2_kept_supported(H, W, mx, my))
                     and os.environ.get("PPSCI_FNO_FULL_FFT", "0") != "1")
        # (the tanh stabilizer puts a pointwise function between a block's output and the next transform's input)
        self.fuse_dft = self.kept and m.fno_blocks.stabilizer != "tanh"
        # the forward contraction inside the inverse transform's launch (PPSCI_FNO_FUSE_CONTRACT=0: two launches; A/B, tests)
        self.fuse_contract = self.kept and os.environ.get("PPSCI_FNO_FUSE_CONTRACT", "1") != "0"
        sp = (B, Ch, mx, my, 2) if (self.kept or self.sht) else (B, Ch, H, Wf, 2)
        self.xft = [torch.empty(sp, **f) for _ in range(nl)]      # unscaled rfftn(x_l): kept for dL/dw
        self.out_ft = torch.empty(sp, **f)  # (full spectrum: cleared + kept modes written every time, C2R destroys it)
        self.gx_ft = torch.empty(sp, **f)
        self.ghat = torch.empty(sp, **f)
        self.gsp = torch.empty((B, Ch, P), **f)
        self.rows = torch.empty(B * Ch * 4, **f)
        self.stats = [torch.empty(4 * B, **f) for _ in range(nl)]
        self.z2 = torch.empty((B, self.c_proj, P0), **f)
        self.gelu_on_load = _virtual(1)
        self.y = torch.empty((B, m.out_channels, P0), **f)
        # backward scratch
        cmax = max(Ch, self.c_lift, self.c_proj)
        self.ga = torch.empty((B, cmax, P), **f)
        self.gb = torch.empty((B, cmax, P), **f)
        if self.use_side or os.environ.get("PPSCI_FNO_OWN_BUFFERS", "0") == "1":  # (the second knob: A/B of the working set alone)
            # one per block: the weight gradients that read them run on the side stream, next to the following blocks' tails
            self.gts = [torch.empty((B, Ch, P), **f) for _ in range(nl)]
            self.gz2 = torch.empty((B, self.c_proj, P0), **f)
        else:
            # one stream: every reader of a block's gradient has been enqueued before the next block overwrites it -- ONE buffer
            # for all blocks, and the projection's hidden gradient in `ga` (the working set stays where round 4 had it)
            gt = torch.empty((B, Ch, P), **f)
            self.gts = [gt] * nl
            self.gz2 = self.ga.view(-1)[:B * self.c_proj * P0].view(B, self.c_proj, P0)
        self.gv = torch.empty((B, Ch, P), **f)
        self._wbufs: List[torch.Tensor] = []  # per-chunk partials of the weight gradients, one buffer per _wgrad call of a pass
        self._wcall, self._wsegs = 0, []
        self.desc = L.SpectralDesc()
        d = self.desc
        d.batch, d.c_in, d.c_out, d.h, d.wf, d.modes_x, d.modes_y = B, Ch, Ch, H, Wf, mx, my
        self.inv_n = 1.0 / float(H * W)

    def _pad(self, src, dst, unpad: bool) -> None:
        B, Ch = self.shape[0], self.m.hidden_channels
        (Hp, Wp), (H0, W0) = self.hw, self.hw0
        L.check(L.lib().ppsci_pad2d(B * Ch, H0, W0, Hp, Wp, self.oh, self.ow, 1 if unpad else 0, _p(src), _p(dst),
                                    _stream_ptr(dst)))

    # ------------------------------------------------------------------ forward
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x: [B, C_in, H, W] on the device -> y [B, C_out, H, W] (a buffer owned by the engine)."""
        m = self.m
        B, _, H0, W0 = x.shape
        if self.shape != (B, H0, W0):
            self._switch(B, H0, W0)
        P, P0, Ch, nl = self.P, self.P0, m.hidden_channels, m.n_layers
        H, W = self.hw
        self.x_in = x.contiguous().view(B, m.in_channels, P0)
        lift, proj, fb = m.lifting.fcs, m.projection.fcs, m.fno_blocks
        x0 = self.x0u if self.padded else self.x[0]
        if self.lift_virtual:
            self.a1_virtual = _virtual(2, self.x_in, lift[0].weight, lift[0].bias)
            _pw_conv(B, self.c_lift, Ch, P0, None, lift[1].weight, x0, bias=lift[1].bias, xv=self.a1_virtual)
        elif self.c_lift:
            _pw_conv(B, m.in_channels, self.c_lift, P0, self.x_in, lift[0].weight, self.z1, bias=lift[0].bias, act=self.a1)
            _pw_conv(B, self.c_lift, Ch, P0, self.a1, lift[1].weight, x0, bias=lift[1].bias)
        else:
            _pw_conv(B, m.in_channels, Ch, P0, self.x_in, lift[0].weight, x0, bias=lift[0].bias)
        if self.padded:
            self._pad(x0, self.x[0], False)
        st = _stream_ptr(self.y)
        for l in range(nl):
            xl = self.x[l]
            conv = fb.convs[l]
            skip = fb.fno_skips[l]
            if isinstance(skip, fno_arch.Conv1x1):
                with self._fork():  # next to the spectral branch; the tail below joins
                    _pw_conv(B, Ch, Ch, P, xl, skip.weight, self.s)
                sk = self.s
            else:
                sk = xl
            xft, v = self.xft[l], self.v[l]
            if self.stab:  # fno_block.py:1199: x = tanh(x) in front of the spectral convolution (the skip sees x)
                L.check(L.lib().ppsci_tanh_fwd(B * Ch * P, _p(xl), _p(self.xs[l]), st))
                xl = self.xs[l]
            nrm = fb.norm[l] if fb.norm is not None else None
            last = l == nl - 1
            if self.kept:
                # The tail's apply kernel of block l - 1 has already left the kept modes of x_l in xft[l] (its workgroups
                # hold whole planes); the inverse transform leaves the row sums of the tail's statistics pass behind
                mx, my = self.desc.modes_x, self.desc.modes_y
                if not (self.fuse_dft and l > 0):
                    L.check(L.lib().ppsci_dft2_kept_fwd(B * Ch, H, W, mx, my, 0, _p(xl), _p(xft), st))
                if self.fuse_contract:
                    # contraction + inverse transform (+ the tail's row sums) in one launch: a workgroup contracts the kept modes
                    # of the plane it then transforms (csrc/spectral_conv.hip spectral_inv_kernel)
                    L.check(L.lib().ppsci_spectral_conv2d_inv_kept(
                        C.byref(self.desc), H, W, 1, _p(xft), _p(conv.weight_real), _p(conv.weight_imag), self.inv_n, _p(v),
                        _p(conv.bias) if nrm is not None else None, _p(self.rows) if nrm is not None else None, st))
                else:
                    L.check(L.lib().ppsci_spectral_conv2d_fwd_kept(C.byref(self.desc), _p(xft), _p(conv.weight_real),
                                                                   _p(conv.weight_imag), _p(self.out_ft), self.inv_n, st))
                    if nrm is not None:
                        L.check(L.lib().ppsci_dft2_kept_inv_stats(B * Ch, H, W, mx, my, 1, _p(self.out_ft), _p(v), _p(conv.bias),
                                                                  Ch, _p(self.rows), st))
                    else:
                        L.check(L.lib().ppsci_dft2_kept_inv(B * Ch, H, W, mx, my, 1, _p(self.out_ft), _p(v), st))
                self._join()
                L.check(L.lib().ppsci_fno_tail_fwd_ex(
                    B, Ch, P, 1 if nrm is not None else 0, 0 if last else 1, float(nrm.eps) if nrm is not None else 0.0, _p(v),
                    _p(conv.bias), _p(nrm.weight) if nrm is not None else None, _p(nrm.bias) if nrm is not None else None,
                    _p(sk), _p(self.rows), _p(self.stats[l]), _p(self.t[l]), None if last else _p(self.x[l + 1]),
                    1 if nrm is not None else 0, H, W, mx, my,
                    _p(self.xft[l + 1]) if (self.fuse_dft and not last) else None, st))
                continue
            elif self.sht:  # sfnonet.py:333-354: sht -> weights per degree -> isht
                mx, my = self.desc.modes_x, self.desc.modes_y
                L.check(L.lib().ppsci_sht_analysis(B * Ch, H, W, mx, my, _p(self.sht_tw), _p(self.sht_a_an), _p(xl), _p(xft), st))
                # PPSCI_SHT_FUSE_CONTRACT=1: the contraction inside the synthesis launch (bit-identical).  Measured on MI355X at the
                # reference's shape: the synthesis grows 13.7 -> 23.6 us (128 plane-workgroups, each walking the 32 channels itself) for
                # a 6.0 us launch saved -- 0.542 against 0.513 ms per step -- so it is OFF by default (kept as a tested knob)
                if os.environ.get("PPSCI_SHT_FUSE_CONTRACT", "0") == "1":
                    L.check(L.lib().ppsci_sht_synthesis_contract(B, Ch, Ch, 0, H, W, mx, my, _p(self.sht_tw), _p(self.sht_b_sy), _p(xft),
                                                                 _p(conv.weight_real), _p(conv.weight_imag), _p(v), st))
                else:
                    L.check(L.lib().ppsci_sht_contract(B, Ch, Ch, mx, my, _p(xft), _p(conv.weight_real), _p(conv.weight_imag), 0,
                                                       _p(self.out_ft), st))
                    L.check(L.lib().ppsci_sht_synthesis(B * Ch, H, W, mx, my, _p(self.sht_tw), _p(self.sht_b_sy), _p(self.out_ft), _p(v), st))
            else:
                L.check(L.lib().ppsci_fft2d_r2c(B * Ch, H, W, _p(xl), _p(xft), st))
                L.check(L.lib().ppsci_spectral_conv2d_fwd_scaled(C.byref(self.desc), _p(xft), _p(conv.weight_real),
                                                                 _p(conv.weight_imag), _p(self.out_ft), self.inv_n, 1, st))
                L.check(L.lib().ppsci_fft2d_c2r(B * Ch, H, W, _p(self.out_ft), _p(v), st))
            self._join()
            L.check(L.lib().ppsci_fno_tail_fwd(
                B, Ch, P, 1 if nrm is not None else 0, 0 if last else 1, float(nrm.eps) if nrm is not None else 0.0, _p(v),
                _p(conv.bias), _p(nrm.weight) if nrm is not None else None, _p(nrm.bias) if nrm is not None else None,
                _p(sk), _p(self.rows), _p(self.stats[l]), _p(self.t[l]), None if last else _p(self.x[l + 1]), st))
        xo = self.t[nl - 1]  # no activation behind the last block
        if self.padded:
            self._pad(xo, self.xou, True)
            xo = self.xou
        self.xo = xo
        _pw_conv(B, Ch, self.c_proj, P0, xo, proj[0].weight, self.z2, bias=proj[0].bias)
        _pw_conv(B, self.c_proj, m.out_channels, P0, self.z2, proj[1].weight, self.y, bias=proj[1].bias, xv=self.gelu_on_load)
        return self.y.view(B, m.out_channels, H0, W0)

    # ------------------------------------------------------------------ backward
    def _partials(self, n: int) -> torch.Tensor:
        """The next per-chunk partial buffer of this backward pass (every weight gradient keeps its own until the one
        reduction launch at the end: self._flush_wgrads)."""
        i = self._wcall
        self._wcall += 1
        if i == len(self._wbufs):
            self._wbufs.append(torch.empty(n, dtype=torch.float32, device=self.y.device))
        assert self._wbufs[i].numel() >= n
        return self._wbufs[i]

    def _wgrad(self, B, ci, co, P, x, gy, w_param, b_param, xv=None) -> None:
        """Weight (+ bias) gradient of a 1x1 convolution: on the side stream -- its operands are complete on the launch stream
        at this point and are not rewritten before backward() joins (per-block `gts`, an own `gz2`) -- next to the chain."""
        with self._fork():
            self._wgrad_(B, ci, co, P, x, gy, w_param, b_param, xv)

    def _wgrad_(self, B, ci, co, P, x, gy, w_param, b_param, xv=None) -> None:
        chunks = int(L.lib().ppsci_pw_conv_wgrad_chunks(B, P))  # (P differs between the padded blocks and lifting / projection)
        wg = w_param.grad.view(-1)
        if b_param is not None and b_param.grad.data_ptr() == wg.data_ptr() + 4 * co * ci:
            # weight and bias gradients are neighbours in the flat buffer: partial rows [Co*Ci | Co], ONE fixed-order sum
            ld = co * ci + co
            part = self._partials(chunks * ld)
            L.check(L.lib().ppsci_pw_conv_wgrad_v(B, ci, co, P, _p(x), C.byref(xv) if xv is not None else None, _p(gy),
                                                  _p(part), C.c_void_p(part.data_ptr() + 4 * co * ci), ld, _stream_ptr(gy)))
            self._wsegs.append((part.data_ptr(), wg.data_ptr(), chunks, ld))
            return
        part = self._partials(chunks * co * ci)
        part_b = self._partials(chunks * co) if b_param is not None else None
        L.check(L.lib().ppsci_pw_conv_wgrad_v(B, ci, co, P, _p(x), C.byref(xv) if xv is not None else None, _p(gy),
                                              _p(part), _p(part_b), 0, _stream_ptr(gy)))
        self._wsegs.append((part.data_ptr(), wg.data_ptr(), chunks, co * ci))
        if b_param is not None:
            self._wsegs.append((part_b.data_ptr(), b_param.grad.view(-1).data_ptr(), chunks, co))

    def _lift_fused_both(self, Ch) -> bool:
        m = self.m
        lift = m.lifting.fcs
        w0, b0, w1, b1 = lift[0].weight, lift[0].bias, lift[1].weight, lift[1].bias
        K0, C1 = m.in_channels, self.c_lift
        return (self.lift_virtual and os.environ.get("PPSCI_FNO_LIFT0_FUSED", "1") != "0" and K0 <= 4 and Ch <= 32 and Ch % 4 == 0
                and b0 is not None and b1 is not None and os.environ.get("PPSCI_FNO_LIFT1_FUSED", "1") != "0"
                and b0.grad.data_ptr() == w0.grad.view(-1).data_ptr() + 4 * C1 * K0
                and b1.grad.data_ptr() == w1.grad.view(-1).data_ptr() + 4 * Ch * C1)

    def _lift_takes_addend(self, Ch) -> bool:
        """The lifting kernel is the ONLY consumer of dL/dx_0 (both lifting gradients from it, no padding in between)."""
        return bool(self.c_lift) and not self.padded and self._lift_fused_both(Ch)

    def _lift0_fused(self, B, Ch, P0, gx, lift, gx_add=None) -> bool:
        """Lifting MLP backward with the hidden tensor virtual: the second layer's weight gradient as before, the first layer's by
        ppsci_fno_lift0_wgrad -- GELU'(W0 x + b0) * (W1^T gx) is formed in registers and reduced against the input channels at
        once (was: 67 MB written by one launch and read back by the next at batch 16, 64 x 64).  False: shape outside the kernel's
        envelope or PPSCI_FNO_LIFT0_FUSED=0 -- the caller takes the two-launch path."""
        m = self.m
        K0, C1 = m.in_channels, self.c_lift
        w0, b0 = lift[0].weight, lift[0].bias
        if (os.environ.get("PPSCI_FNO_LIFT0_FUSED", "1") == "0" or K0 > 4 or Ch > 64 or Ch % 4 != 0 or b0 is None
                or b0.grad.data_ptr() != w0.grad.view(-1).data_ptr() + 4 * C1 * K0):
            return False
        w1, b1 = lift[1].weight, lift[1].bias
        # the second layer's gradient from the same pass (GELU(z1) is at hand there) when its weight and bias gradients are neighbours
        both = (Ch <= 32 and b1 is not None and b1.grad.data_ptr() == w1.grad.view(-1).data_ptr() + 4 * Ch * C1
                and os.environ.get("PPSCI_FNO_LIFT1_FUSED", "1") != "0")
        if not both:
            self._wgrad(B, C1, Ch, P0, None, gx, w1, b1, xv=self.a1_virtual)
        with self._fork():
            chunks = int(L.lib().ppsci_fno_lift0_wgrad_chunks(B, P0))
            ld = C1 * K0 + C1
            part = self._partials(chunks * ld)
            ld1 = Ch * C1 + Ch
            part1 = self._partials(chunks * ld1) if both else None
            assert gx_add is None or both
            L.check(L.lib().ppsci_fno_lift0_wgrad(B, K0, C1, Ch, P0, _p(self.x_in), _p(w0), _p(b0), _p(w1), _p(gx), _p(gx_add),
                                                  _p(part), C.c_void_p(part.data_ptr() + 4 * C1 * K0), ld,
                                                  _p(part1) if both else None, ld1 if both else 0, _stream_ptr(gx)))
            self._wsegs.append((part.data_ptr(), w0.grad.view(-1).data_ptr(), chunks, ld))
            if both:
                self._wsegs.append((part1.data_ptr(), w1.grad.view(-1).data_ptr(), chunks, ld1))
        return True

    defer_wgrad_sums = False  # backward() leaves the partials of the weight gradients unsummed (self._wsegs) when set

    def _flush_wgrads(self) -> None:
        """ONE launch sums the per-chunk partials of every weight gradient of the pass (ppsci_reduce_rows_multi; up to 16
        segments per launch): eight reductions of ~5 us each were launch latency, not work."""
        st = _stream_ptr(self.y)
        for i0 in range(0, len(self._wsegs), 16):
            batch = self._wsegs[i0:i0 + 16]
            arr = (L.ReduceSeg * len(batch))()
            for k, (src, dst, rows, cols) in enumerate(batch):
                arr[k].partials, arr[k].out, arr[k].rows, arr[k].cols, arr[k].accumulate = src, dst, rows, cols, 0
            L.check(L.lib().ppsci_reduce_rows_multi(len(batch), arr, st))
        self._wsegs = []

    def backward(self, gy: torch.Tensor) -> None:
        """gy = dL/dy [B, C_out, H, W]; writes dL/d(parameter) into every parameter's `.grad` (views of flat_grad)."""
        m = self.m
        B = self.shape[0]
        H, W = self.hw
        P, P0, Ch, nl = self.P, self.P0, m.hidden_channels, m.n_layers
        lift, proj, fb = m.lifting.fcs, m.projection.fcs, m.fno_blocks
        gy = gy.contiguous().view(B, m.out_channels, P0)
        st = _stream_ptr(self.y)
        self._wcall, self._wsegs = 0, []
        # projection: y = W2 gelu(z2) + b2, z2 = W1 x_out + b1
        self._wgrad(B, self.c_proj, m.out_channels, P0, self.z2, gy, proj[1].weight, proj[1].bias, xv=self.gelu_on_load)
        gz2 = self.gz2
        if (m.out_channels <= 4 and P0 % 4 == 0 and os.environ.get("PPSCI_FNO_PROJ_STREAMED", "1") != "0"
                and (gy.data_ptr() | gz2.data_ptr() | self.z2.data_ptr()) % 16 == 0):
            # <= 4 output channels: the hidden gradient is GELU'(z2) times a rank-m factor -- streamed (fno.hip), not a K = m GEMM
            L.check(L.lib().ppsci_fno_proj_hidden_grad(B, self.c_proj, m.out_channels, P0, _p(self.z2), _p(proj[1].weight), _p(gy),
                                                       _p(gz2), st))
        else:
            _pw_conv(B, m.out_channels, self.c_proj, P0, gy, proj[1].weight, gz2, zmul=self.z2, transpose=True)
        self._wgrad(B, Ch, self.c_proj, P0, self.xo, gz2, proj[0].weight, proj[0].bias)
        gx = self.gb.view(-1)[:B * Ch * P].view(B, Ch, P)  # dL/d(block output), ping-pongs with `gnext`
        gnext = self.ga.view(-1)[:B * Ch * P].view(B, Ch, P)
        if self.padded:  # the gradient of unpad is pad: zeros outside the window
            gxu = self.x0u  # (free during the backward pass; NOT xou: the weight gradient above still reads it on the side stream)
            _pw_conv(B, self.c_proj, Ch, P0, gz2, proj[0].weight, gxu, transpose=True)
            self._pad(gxu, self.gpad, False)
            gx = self.gpad
            gnext = self.gb.view(-1)[:B * Ch * P].view(B, Ch, P)
        else:
            _pw_conv(B, self.c_proj, Ch, P, gz2, proj[0].weight, gx, transpose=True)
        gx2 = None  # a second addend of dL/d(block output): the spectral branch's share, added by the consumer on load
        gx_add = None  # ... of dL/dx_0, for the lifting kernel
        gx2_modes = None  # ... or its kept modes: the consumer evaluates the inverse transform itself (fuse_dft)
        for l in range(nl - 1, -1, -1):
            conv, skip = fb.convs[l], fb.fno_skips[l]
            gt = self.gts[l]
            nrm = fb.norm[l] if fb.norm is not None else None
            last = l == nl - 1
            if self.kept:  # dL/dv only feeds the spectral branch: its kept modes come out of the tail's second pass, gv is not stored
                L.check(L.lib().ppsci_fno_tail_bwd_ex(
                    B, Ch, P, 1 if nrm is not None else 0, 0 if last else 1, _p(self.v[l]), _p(conv.bias),
                    _p(nrm.weight) if nrm is not None else None, _p(self.t[l]), _p(gx), _p(gx2), _p(self.rows),
                    _p(self.stats[l]), _p(gt), None, _p(nrm.weight.grad) if nrm is not None else None,
                    _p(nrm.bias.grad) if nrm is not None else None, _p(conv.bias.grad), H, W, self.desc.modes_x,
                    self.desc.modes_y, _p(self.ghat), _p(gx2_modes), st))
            else:
                L.check(L.lib().ppsci_fno_tail_bwd(
                    B, Ch, P, 1 if nrm is not None else 0, 0 if last else 1, _p(self.v[l]), _p(conv.bias),
                    _p(nrm.weight) if nrm is not None else None, _p(self.t[l]), _p(gx), _p(gx2), _p(self.rows),
                    _p(self.stats[l]), _p(gt), _p(self.gv), _p(nrm.weight.grad) if nrm is not None else None,
                    _p(nrm.bias.grad) if nrm is not None else None, _p(conv.bias.grad), st))
            # skip branch: s = Wskip x_l  (identity: the gradient passes straight through)
            if isinstance(skip, fno_arch.Conv1x1):
                self._wgrad(B, Ch, Ch, P, self.x[l], gt, skip.weight, None)
                _pw_conv(B, Ch, Ch, P, gt, skip.weight, gnext, transpose=True)
            else:
                hp.reduce_rows(gt.view(1, -1), 1, B * Ch * P, gnext.view(-1), False)
            # spectral branch: dL/dx_l += irfftn( rfftn(gv) . conj(w)^T ), weight gradients from x_ft and rfftn(gv)
            if self.kept:  # (the adjoint reads dL/dy's spectrum at the OUTPUT rows and writes the input rows)
                mx, my = self.desc.modes_x, self.desc.modes_y
                L.check(L.lib().ppsci_spectral_conv2d_bwd_kept(
                    C.byref(self.desc), _p(self.xft[l]), _p(conv.weight_real), _p(conv.weight_imag), _p(self.ghat),
                    _p(self.gx_ft), _p(conv.weight_real.grad), _p(conv.weight_imag.grad), self.inv_n, W, self.inv_n, st))
                if self.fuse_dft and l > 0:
                    # the tail of block l - 1 inverse-transforms these modes plane by plane in LDS; gx_ft is rewritten only
                    # by that block's own spectral backward, which runs behind its tail
                    gx2, gx2_modes = None, self.gx_ft
                    gx, gnext = gnext, gx
                    continue
                gx2_modes = None
                L.check(L.lib().ppsci_dft2_kept_inv(B * Ch, H, W, mx, my, 0, _p(self.gx_ft), _p(self.gsp), st))
            elif self.sht:  # each transform's adjoint is the other kernel on its own table (csrc/sht.hip)
                mx, my = self.desc.modes_x, self.desc.modes_y
                L.check(L.lib().ppsci_sht_analysis(B * Ch, H, W, mx, my, _p(self.sht_tw), _p(self.sht_b_an), _p(self.gv), _p(self.ghat), st))
                L.check(L.lib().ppsci_sht_contract_wgrad(B, Ch, Ch, mx, my, _p(self.xft[l]), _p(self.ghat), _p(conv.weight_real.grad),
                                                         _p(conv.weight_imag.grad), st))
                if os.environ.get("PPSCI_SHT_FUSE_CONTRACT", "0") == "1":
                    L.check(L.lib().ppsci_sht_synthesis_contract(B, Ch, Ch, 1, H, W, mx, my, _p(self.sht_tw), _p(self.sht_a_sy),
                                                                 _p(self.ghat), _p(conv.weight_real), _p(conv.weight_imag), _p(self.gsp), st))
                else:
                    L.check(L.lib().ppsci_sht_contract(B, Ch, Ch, mx, my, _p(self.ghat), _p(conv.weight_real), _p(conv.weight_imag), 1,
                                                       _p(self.gx_ft), st))
                    L.check(L.lib().ppsci_sht_synthesis(B * Ch, H, W, mx, my, _p(self.sht_tw), _p(self.sht_a_sy), _p(self.gx_ft), _p(self.gsp), st))
            else:
                L.check(L.lib().ppsci_fft2d_r2c(B * Ch, H, W, _p(self.gv), _p(self.ghat), st))
                L.check(L.lib().ppsci_spectral_conv2d_bwd_real_scaled(
                    C.byref(self.desc), _p(self.xft[l]), _p(conv.weight_real), _p(conv.weight_imag), _p(self.ghat),
                    _p(self.gx_ft), _p(conv.weight_real.grad), _p(conv.weight_imag.grad), self.inv_n, W, self.inv_n, 1, st))
                L.check(L.lib().ppsci_fft2d_c2r(B * Ch, H, W, _p(self.gx_ft), _p(self.gsp), st))
            if self.stab:  # gnext += gsp * (1 - tanh(x_l)^2)
                L.check(L.lib().ppsci_tanh_bwd(B * Ch * P, _p(self.xs[l]), _p(self.gsp), _p(gnext), 1, st))
                gx2 = None
            elif l > 0:
                gx2 = self.gsp  # dL/dx_l = gnext + gsp: the next block tail adds them on load (gsp is rewritten after it)
            elif self._lift_takes_addend(Ch):
                gx_add = self.gsp  # the lifting kernel adds it while it stages dL/dx_0 (one launch and 25 MB of traffic less)
            else:
                hp.reduce_rows(self.gsp.view(1, -1), 1, B * Ch * P, gnext.view(-1), True)  # gnext += gsp
            gx, gnext = gnext, gx
        # lifting (on the unpadded planes: the gradient of pad is unpad)
        if self.padded:
            self._pad(gx, self.x0u, True)
            gx = self.x0u
        if self.c_lift:
            gz1 = self.gb if gx.data_ptr() != self.gb.data_ptr() else self.ga
            gz1 = gz1.view(-1)[:B * self.c_lift * P0].view(B, self.c_lift, P0)
            if self.lift_virtual and self._lift0_fused(B, Ch, P0, gx, lift, gx_add):
                pass  # (both weight gradients done: the first layer's without its hidden gradient in memory)
            elif self.lift_virtual:
                self._wgrad(B, self.c_lift, Ch, P0, None, gx, lift[1].weight, lift[1].bias, xv=self.a1_virtual)
                _pw_conv(B, Ch, self.c_lift, P0, gx, lift[1].weight, gz1, transpose=True, zv=self.a1_virtual)
                self._wgrad(B, m.in_channels, self.c_lift, P0, self.x_in, gz1, lift[0].weight, lift[0].bias)
            else:
                self._wgrad(B, self.c_lift, Ch, P0, self.a1, gx, lift[1].weight, lift[1].bias)
                _pw_conv(B, Ch, self.c_lift, P0, gx, lift[1].weight, gz1, zmul=self.z1, transpose=True)
                self._wgrad(B, m.in_channels, self.c_lift, P0, self.x_in, gz1, lift[0].weight, lift[0].bias)
        else:
            self._wgrad(B, m.in_channels, Ch, P0, self.x_in, gx, lift[0].weight, lift[0].bias)
        self._join()  # every weight gradient's partial rows are complete
        if not self.defer_wgrad_sums:
            self._flush_wgrads()  # (deferred: the caller sums them together with its Adam update, operator_engine)
