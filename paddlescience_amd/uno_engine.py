"""Native forward + backward of ppsci.arch.UNONet (/root/reference/ppsci/arch/unonet.py:246-289): the FNO executor's kernels
(fno_engine.FnoNative: MFMA 1x1 convolutions, raw hipFFT executions, the per-mode complex contraction, the fused
bias + GroupNorm + skip block tail) plus the two operations of csrc/uno.hip that let a block change resolution, composed by hand
in both directions -- no autograd graph, no library operator.

Per Fourier layer i, input grid (H, W) -> output grid (H2, W2) = round(size * uno_scalings[i]) (the last layer: the end-to-end
grid, unonet.py:251-254, :273-274), channels Cin -> Cout:

    spectral   X = R2C(x)                       ppsci_fft2d_r2c                 rfftn              fno_block.py:718
               Z = s * X . w on the kept modes  ppsci_spectral_conv2d_fwd_scaled                   fno_block.py:721-777
               Z2 = rows [0,H2), cols [0,W2/2]  ppsci_spectrum_resize            irfftn(.., s=)    fno_block.py:779-793
               v = C2R(Z2) at (H2, W2)          ppsci_fft2d_c2r
    skip       S = bicubic(Wskip x)             ppsci_pw_conv_v, ppsci_resample2d                  fno_block.py:1192-1193, :466-498
    tail       t = norm(v + bias) + S           ppsci_fno_tail_fwd (no activation: n_layers = 1)   fno_block.py:1203-1210
    U skips    h_i = Whs_i t_i;  x_j = concat(t_{j-1}, bicubic(h_i))                               unonet.py:259-272, :276-277

s = 1 / (H2 W2): the reference's blocks transform with the norm "backward" whatever `fft_norm` is set to (see _alloc).  Backward: every step above is linear except the tail; the adjoint of the crop is
the pad (and the other way round), of the bicubic matrix pair its transposes, of a real transform the other real transform with the
Hermitian weights c(j) (1 on the DC / Nyquist column, 2 elsewhere) moved across -- which cancels inside one grid (fno_engine) and
leaves the factor c_{W2}(j) / c_W(j) on the columns where the two grids disagree (applied by ppsci_spectrum_resize on the way back).

Covered: 2-D, dense weights, GroupNorm(1 group) or no norm, linear or identity block skips, linear horizontal skips, DomainPadding
(with an end-to-end scaling of 1, as the reference's unpad needs)."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib as L
from . import hotpath as hp
from .arch import fno as fno_arch
from .fno_engine import FnoNative, _pw_conv, _virtual
from .hotpath import _p, _stream_ptr


def bicubic_matrix(n_in: int, n_out: int) -> np.ndarray:
    """[n_out, n_in]: one axis of F.interpolate(mode="bicubic", align_corners=True) -- the cubic convolution kernel with
    A = -0.75 on the four neighbours of o * (n_in - 1) / (n_out - 1), neighbour indices clamped to the plane."""
    A = -0.75
    M = np.zeros((n_out, n_in), dtype=np.float64)
    scale = (n_in - 1) / (n_out - 1) if n_out > 1 else 0.0
    for o in range(n_out):
        src = scale * o
        i0 = int(np.floor(src))
        t = src - i0
        w = (((A * (t + 1) - 5 * A) * (t + 1) + 8 * A) * (t + 1) - 4 * A,
             ((A + 2) * t - (A + 3)) * t * t + 1,
             ((A + 2) * (1 - t) - (A + 3)) * (1 - t) * (1 - t) + 1,
             ((A * (2 - t) - 5 * A) * (2 - t) + 8 * A) * (2 - t) - 4 * A)
        for k in range(4):
            M[o, min(max(i0 - 1 + k, 0), n_in - 1)] += w[k]
    return M


def supports(model) -> Optional[str]:
    from .arch import uno

    if not isinstance(model, uno.UNONet):
        return "not a UNONet"
    return None


class _Resampler:
    """The matrix pairs of one (H, W) -> (H2, W2) bicubic map on the device, forward and transposed."""

    def __init__(self, H, W, H2, W2, device):
        self.src, self.dst = (H, W), (H2, W2)
        if not L.lib().ppsci_resample2d_supported(H, W, H2, W2) or not L.lib().ppsci_resample2d_supported(H2, W2, H, W):
            raise NotImplementedError(f"bicubic resampling {H} x {W} -> {H2} x {W2}: a plane does not fit LDS")
        ah, aw = bicubic_matrix(H, H2), bicubic_matrix(W, W2)
        f = dict(dtype=torch.float32, device=device)
        self.ah, self.aw = torch.tensor(ah, **f), torch.tensor(aw, **f)
        self.aht, self.awt = torch.tensor(ah.T.copy(), **f), torch.tensor(aw.T.copy(), **f)

    def apply(self, n, x, y, accumulate=False):
        (H, W), (H2, W2) = self.src, self.dst
        L.check(L.lib().ppsci_resample2d(n, H, W, H2, W2, _p(x), _p(self.ah), _p(self.aw), _p(y), 1 if accumulate else 0,
                                         _stream_ptr(y)))

    def adjoint(self, n, gy, gx, accumulate=False):
        (H, W), (H2, W2) = self.src, self.dst
        L.check(L.lib().ppsci_resample2d(n, H2, W2, H, W, _p(gy), _p(self.aht), _p(self.awt), _p(gx), 1 if accumulate else 0,
                                         _stream_ptr(gx)))


class UnoNative(FnoNative):
    """Same contract as fno_engine.FnoNative (`forward`, `backward`, `generation`, deferred weight-gradient sums): the operator
    engine, the solver's eval / predict paths and the HIP-graph capture of the step do not tell the two apart."""

    def __init__(self, model):
        why = supports(model)
        if why is not None:
            raise NotImplementedError(f"native UNO path: {why}")
        self.m = model
        self.shape = None
        self._sets = {}
        self.max_sets = 8
        self.generation = 0
        self.use_side = False
        self._side = None

    def _switch(self, B: int, H: int, W: int) -> None:
        keep = ("m", "shape", "_sets", "max_sets", "generation", "use_side", "_side", "defer_wgrad_sums")
        return self._switch_(B, H, W, keep)

    # ------------------------------------------------------------------ buffers
    def _alloc(self, B: int, H0: int, W0: int) -> None:
        m = self.m
        dev = m.flat_params.device
        f = dict(dtype=torch.float32, device=dev)
        nl = m.n_layers
        ah, aw, self.oh, self.ow = m.padding_of(H0, W0)
        Hp, Wp = H0 + ah, W0 + aw
        self.padded = bool(ah or aw)
        final = (int(round(Hp * m.end_to_end_scaling_factor[0])), int(round(Wp * m.end_to_end_scaling_factor[1])))
        if self.padded and final != (Hp, Wp):
            # (the reference's DomainPadding.unpad looks its slices up under the padded INPUT shape, fno_block.py:124-140)
            raise NotImplementedError("UNONet: domain_padding with an end-to-end scaling other than 1")
        self.hw0, self.hwp = (H0, W0), (Hp, Wp)
        # geometry of the layers
        self.geo = []
        cur = (Hp, Wp)
        for i, blk in enumerate(m.fno_blocks):
            out = final if i == nl - 1 else blk.out_shape(*cur)
            if min(out) < 1:
                raise ValueError(f"UNONet layer {i}: the scaled grid {out} is empty")
            mx, my = blk.convs[0].n_modes
            if mx > cur[0] or my > cur[0 + 1] // 2 + 1:
                raise NotImplementedError(f"UNONet layer {i}: {mx} x {my} kept modes on a {cur[0]} x {cur[1]} grid (the reference "
                                          "slices its weights then; not built)")
            self.geo.append((cur, out))
            cur = out
        self.hw_out = (H0, W0) if self.padded else final
        P0, Pout = H0 * W0, self.hw_out[0] * self.hw_out[1]
        self.P0, self.Pout = P0, Pout
        self.shape = (B, H0, W0)
        lift, proj = m.lifting.fcs, m.projection.fcs
        self.c_lift, self.c_proj = lift[0].out_channels, proj[0].out_channels
        Ch = m.hidden_channels
        self.z1, self.a1 = torch.empty((B, self.c_lift, P0), **f), torch.empty((B, self.c_lift, P0), **f)
        self.x0u = torch.empty((B, Ch, P0), **f) if self.padded else None
        self.x0 = torch.empty((B, Ch, Hp * Wp), **f)
        rs: Dict[tuple, _Resampler] = {}

        def resampler(src, dst):
            if src == dst:
                return None
            if (src, dst) not in rs:
                rs[(src, dst)] = _Resampler(*src, *dst, dev)
            return rs[(src, dst)]

        self.blk = []
        cmax = max(Ch, self.c_lift, self.c_proj)
        pmax = max(P0, Pout, Hp * Wp)
        for i, blk in enumerate(m.fno_blocks):
            (H, W), (H2, W2) = self.geo[i]
            ci, co = blk.in_channels, blk.out_channels
            P, P2 = H * W, H2 * W2
            Wf, Wf2 = W // 2 + 1, W2 // 2 + 1
            d = L.SpectralDesc()
            d.batch, d.c_in, d.c_out, d.h, d.wf = B, ci, co, H, Wf
            d.modes_x, d.modes_y = blk.convs[0].n_modes
            # FNOBlocks does not hand `fft_norm` on to its SpectralConv (fno_block.py:1099-1111): the transforms ALWAYS run with
            # the default norm "backward" -- rfftn unscaled, irfftn / (H2 W2) -- whatever UNONet(fft_norm=...) says.  On one grid
            # every norm gives the same product (fno_engine); between two grids they differ, and "backward" is what the reference does.
            scale = 1.0 / float(H2 * W2)
            src = m.horizontal_skips_map.get(i)
            # the transforms on the kept modes only (two small DFTs per plane in LDS, csrc/spectral_conv.hip) when the planes of both
            # grids fit; hipFFT on full spectra + ppsci_spectrum_resize otherwise (PPSCI_FNO_FULL_FFT=1 forces it: tests, timing)
            kept = (bool(L.lib().ppsci_dft2_kept_supported(H, W, d.modes_x, d.modes_y))
                    and bool(L.lib().ppsci_dft2_kept_from_supported(H2, W2, d.modes_x, d.modes_y))
                    and os.environ.get("PPSCI_FNO_FULL_FFT", "0") != "1")
            nk = (d.modes_x, d.modes_y, 2)
            e = dict(
                kept=kept, xk=torch.empty((B, ci) + nk, **f) if kept else None,
                desc=d, scale=scale, hw=(H, W), hw2=(H2, W2), ci=ci, co=co, src=src, resized=(H, W) != (H2, W2),
                xin=torch.empty((B, ci, P), **f) if src is not None else None,  # concat(previous output, U skip)
                xft=torch.empty((B, ci, H, Wf, 2), **f) if not kept else None,
                out_ft=torch.empty((B, co, H, Wf, 2), **f) if not kept else None,
                out_ft2=torch.empty((B, co, H2, Wf2, 2), **f) if (H, W) != (H2, W2) and not kept else None,
                v=torch.empty((B, co, P2), **f), t=torch.empty((B, co, P2), **f), stats=torch.empty(4 * B, **f),
                s_low=torch.empty((B, co, P), **f) if isinstance(blk.fno_skips[0], fno_arch.Conv1x1) else None,
                s=torch.empty((B, co, P2), **f) if (H, W) != (H2, W2) else None,
                rs=resampler((H, W), (H2, W2)),
                hs=None, ghs=None, rs_h=None)
            if src is not None:
                e["rs_h"] = resampler(self.geo[src][1], (H, W))
                e["hup"] = torch.empty((B, m.uno_out_channels[src], P), **f) if e["rs_h"] is not None else None
            if i in m.horizontal_skips_map.values():
                e["hs"], e["ghs"] = torch.empty((B, co, P2), **f), torch.empty((B, co, P2), **f)
            cmax, pmax = max(cmax, ci, co), max(pmax, P, P2)
            self.blk.append(e)
        self.rows = torch.empty(B * cmax * 4, **f)
        self.xou = torch.empty((B, m.uno_out_channels[-1], Pout), **f) if self.padded else None
        self.z2 = torch.empty((B, self.c_proj, Pout), **f)
        self.gelu_on_load = _virtual(1)
        self.y = torch.empty((B, m.out_channels, Pout), **f)
        # backward scratch: every buffer big enough for any layer's [B, C, P]
        n = B * cmax * pmax
        self.g = [torch.empty(n, **f) for _ in range(6)]
        nf = max(B * max(e["ci"], e["co"]) * max(e["hw"][0] * (e["hw"][1] // 2 + 1), e["hw2"][0] * (e["hw2"][1] // 2 + 1))
                 for e in self.blk) * 2
        self.gf = [torch.empty(nf, **f) for _ in range(3)]
        self._wbufs: List[torch.Tensor] = []
        self._wcall, self._wsegs = 0, []

    @staticmethod
    def _view(buf, *shape):
        n = int(np.prod(shape))
        return buf[:n].view(*shape)

    def _pad_planes(self, n, src, dst, unpad: bool) -> None:
        (H0, W0), (Hp, Wp) = self.hw0, self.hwp
        L.check(L.lib().ppsci_pad2d(n, H0, W0, Hp, Wp, self.oh, self.ow, 1 if unpad else 0, _p(src), _p(dst), _stream_ptr(dst)))

    # ------------------------------------------------------------------ forward
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        m = self.m
        B, _, H0, W0 = x.shape
        if self.shape != (B, H0, W0):
            self._switch(B, H0, W0)
        P0, Ch, nl = self.P0, m.hidden_channels, m.n_layers
        lift, proj = m.lifting.fcs, m.projection.fcs
        self.x_in = x.contiguous().view(B, m.in_channels, P0)
        st = _stream_ptr(self.y)
        x0 = self.x0u if self.padded else self.x0
        _pw_conv(B, m.in_channels, self.c_lift, P0, self.x_in, lift[0].weight, self.z1, bias=lift[0].bias, act=self.a1)
        _pw_conv(B, self.c_lift, Ch, P0, self.a1, lift[1].weight, x0, bias=lift[1].bias)
        if self.padded:
            self._pad_planes(B * Ch, x0, self.x0, False)
        cur = self.x0
        for i, (blk, e) in enumerate(zip(m.fno_blocks, self.blk)):
            (H, W), (H2, W2), ci, co = e["hw"], e["hw2"], e["ci"], e["co"]
            P, P2 = H * W, H2 * W2
            if e["src"] is not None:  # unonet.py:259-272: concat(x, bicubic(h_src -> this grid)) along the channels
                hsrc = self.blk[e["src"]]["hs"]
                if e["rs_h"] is not None:
                    e["rs_h"].apply(B * hsrc.shape[1], hsrc, e["hup"])
                    hsrc = e["hup"]
                torch.cat([cur, hsrc], dim=1, out=e["xin"])
                cur = e["xin"]
            e["x"] = cur
            conv, skip = blk.convs[0], blk.fno_skips[0]
            if isinstance(skip, fno_arch.Conv1x1):
                _pw_conv(B, ci, co, P, cur, skip.weight, e["s_low"])
                sk = e["s_low"]
            else:
                sk = cur
            if e["rs"] is not None:
                e["rs"].apply(B * co, sk, e["s"])
                sk = e["s"]
            mx, my = e["desc"].modes_x, e["desc"].modes_y
            if e["kept"]:  # the kept modes of x, then contraction + inverse transform (onto the other grid, if it changes) in one launch
                L.check(L.lib().ppsci_dft2_kept_fwd(B * ci, H, W, mx, my, 0, _p(cur), _p(e["xk"]), st))
                hs, ws = (H, W) if e["resized"] else (0, 0)
                L.check(L.lib().ppsci_spectral_conv2d_inv_kept_ex(C.byref(e["desc"]), H2, W2, hs, ws, 0, 1, _p(e["xk"]),
                                                                  _p(conv.weight_real), _p(conv.weight_imag), e["scale"], _p(e["v"]),
                                                                  None, None, st))
            else:
                L.check(L.lib().ppsci_fft2d_r2c(B * ci, H, W, _p(cur), _p(e["xft"]), st))
                L.check(L.lib().ppsci_spectral_conv2d_fwd_scaled(C.byref(e["desc"]), _p(e["xft"]), _p(conv.weight_real),
                                                                 _p(conv.weight_imag), _p(e["out_ft"]), e["scale"], 1, st))
                zf = e["out_ft"]
                if e["resized"]:
                    L.check(L.lib().ppsci_spectrum_resize(B * co, H, W // 2 + 1, H2, W2 // 2 + 1, 0, 0, _p(zf), _p(e["out_ft2"]), st))
                    zf = e["out_ft2"]
                L.check(L.lib().ppsci_fft2d_c2r(B * co, H2, W2, _p(zf), _p(e["v"]), st))
            nrm = blk.norm[0] if blk.norm is not None else None
            L.check(L.lib().ppsci_fno_tail_fwd(
                B, co, P2, 1 if nrm is not None else 0, 0, float(nrm.eps) if nrm is not None else 0.0, _p(e["v"]), _p(conv.bias),
                _p(nrm.weight) if nrm is not None else None, _p(nrm.bias) if nrm is not None else None, _p(sk), _p(self.rows),
                _p(e["stats"]), _p(e["t"]), None, st))
            cur = e["t"]
            if e["hs"] is not None:
                _pw_conv(B, co, co, P2, cur, m.horizontal_skips[str(i)].weight, e["hs"])
        co = m.uno_out_channels[-1]
        if self.padded:
            self._pad_planes(B * co, cur, self.xou, True)
            cur = self.xou
        self.xo = cur
        Pout = self.Pout
        _pw_conv(B, co, self.c_proj, Pout, cur, proj[0].weight, self.z2, bias=proj[0].bias)
        _pw_conv(B, self.c_proj, m.out_channels, Pout, self.z2, proj[1].weight, self.y, bias=proj[1].bias, xv=self.gelu_on_load)
        return self.y.view(B, m.out_channels, *self.hw_out)

    # ------------------------------------------------------------------ backward
    def backward(self, gy: torch.Tensor) -> None:
        m = self.m
        B = self.shape[0]
        P0, Pout, Ch, nl = self.P0, self.Pout, m.hidden_channels, m.n_layers
        lift, proj = m.lifting.fcs, m.projection.fcs
        gy = gy.contiguous().view(B, m.out_channels, Pout)
        st = _stream_ptr(self.y)
        self._wcall, self._wsegs = 0, []
        V = self._view
        co = m.uno_out_channels[-1]
        # projection
        self._wgrad(B, self.c_proj, m.out_channels, Pout, self.z2, gy, proj[1].weight, proj[1].bias, xv=self.gelu_on_load)
        gz2 = V(self.g[0], B, self.c_proj, Pout)
        _pw_conv(B, m.out_channels, self.c_proj, Pout, gy, proj[1].weight, gz2, zmul=self.z2, transpose=True)
        self._wgrad(B, co, self.c_proj, Pout, self.xo, gz2, proj[0].weight, proj[0].bias)
        # gx: dL/d(output of layer i); buffers g[1] / g[2] alternate, g[0] / g[3] / g[4] / g[5] are per-layer scratch
        gx = V(self.g[1], B, co, self.geo[-1][1][0] * self.geo[-1][1][1])
        if self.padded:
            gxu = V(self.g[3], B, co, Pout)
            _pw_conv(B, self.c_proj, co, Pout, gz2, proj[0].weight, gxu, transpose=True)
            self._pad_planes(B * co, gxu, gx, False)
        else:
            _pw_conv(B, self.c_proj, co, Pout, gz2, proj[0].weight, gx, transpose=True)
        cur, other = 1, 2
        for i in range(nl - 1, -1, -1):
            blk, e = m.fno_blocks[i], self.blk[i]
            (H, W), (H2, W2), ci, co = e["hw"], e["hw2"], e["ci"], e["co"]
            P, P2 = H * W, H2 * W2
            conv, skip = blk.convs[0], blk.fno_skips[0]
            nrm = blk.norm[0] if blk.norm is not None else None
            gx2 = None
            if e["hs"] is not None:  # this layer's output also fed a U skip: h = Whs t
                hw = m.horizontal_skips[str(i)].weight
                self._wgrad(B, co, co, P2, e["t"], e["ghs"], hw, None)
                gx2 = V(self.g[3], B, co, P2)
                _pw_conv(B, co, co, P2, e["ghs"], hw, gx2, transpose=True)
            gt, gv = V(self.g[4], B, co, P2), V(self.g[5], B, co, P2)
            L.check(L.lib().ppsci_fno_tail_bwd(
                B, co, P2, 1 if nrm is not None else 0, 0, _p(e["v"]), _p(conv.bias), _p(nrm.weight) if nrm is not None else None,
                _p(e["t"]), _p(gx), _p(gx2), _p(self.rows), _p(e["stats"]), _p(gt), _p(gv),
                _p(nrm.weight.grad) if nrm is not None else None, _p(nrm.bias.grad) if nrm is not None else None,
                _p(conv.bias.grad), st))
            # skip branch: S = bicubic(Wskip x)
            gs = gt
            if e["rs"] is not None:
                gs = V(self.g[3], B, co, P)
                e["rs"].adjoint(B * co, gt, gs)
            gxin = V(self.g[other], B, ci, P)
            if isinstance(skip, fno_arch.Conv1x1):
                self._wgrad(B, ci, co, P, e["x"], gs, skip.weight, None)
                _pw_conv(B, co, ci, P, gs, skip.weight, gxin, transpose=True)
            else:
                hp.reduce_rows(gs.reshape(1, -1), 1, B * ci * P, gxin.view(-1), False)
            # spectral branch
            Wf, Wf2 = W // 2 + 1, W2 // 2 + 1
            gsp = V(self.g[0], B, ci, P)
            if e["kept"]:
                mx, my = e["desc"].modes_x, e["desc"].modes_y
                ghat = V(self.gf[0], B, co, mx, my, 2)
                L.check(L.lib().ppsci_dft2_kept_fwd_from(B * co, H2, W2, mx, my, H, W, _p(gv), _p(ghat), st))
                # weight gradients alone (threads along the modes), then the data gradient's contraction inside its inverse transform
                L.check(L.lib().ppsci_spectral_conv2d_bwd_kept(
                    C.byref(e["desc"]), _p(e["xk"]), _p(conv.weight_real), _p(conv.weight_imag), _p(ghat), None,
                    _p(conv.weight_real.grad), _p(conv.weight_imag.grad), e["scale"], W, e["scale"], st))
                # (3 = conjugate + accumulate: gxin holds the skip branch's share already)
                L.check(L.lib().ppsci_spectral_conv2d_inv_kept_ex(C.byref(e["desc"]), H, W, 0, 0, 3, 0, _p(ghat), _p(conv.weight_real),
                                                                  _p(conv.weight_imag), e["scale"], _p(gxin), None, None, st))
            else:
                ghat = V(self.gf[0], B, co, H2, Wf2, 2)
                L.check(L.lib().ppsci_fft2d_r2c(B * co, H2, W2, _p(gv), _p(ghat), st))
                G = ghat
                if e["resized"]:
                    G = V(self.gf[1], B, co, H, Wf, 2)
                    L.check(L.lib().ppsci_spectrum_resize(B * co, H2, Wf2, H, Wf, W2, W, _p(ghat), _p(G), st))
                gx_ft = V(self.gf[2], B, ci, H, Wf, 2)
                L.check(L.lib().ppsci_spectral_conv2d_bwd_real_scaled(
                    C.byref(e["desc"]), _p(e["xft"]), _p(conv.weight_real), _p(conv.weight_imag), _p(G), _p(gx_ft),
                    _p(conv.weight_real.grad), _p(conv.weight_imag.grad), e["scale"], W, e["scale"], 1, st))
                L.check(L.lib().ppsci_fft2d_c2r(B * ci, H, W, _p(gx_ft), _p(gsp), st))
                hp.reduce_rows(gsp.view(1, -1), 1, B * ci * P, gxin.view(-1), True)  # gxin += gsp
            # the input was concat(previous output, U skip): split the gradient
            if e["src"] is not None:
                se = self.blk[e["src"]]
                cs = se["co"]
                cp = ci - cs
                gprev = V(self.g[cur], B, cp, P)
                gprev.copy_(gxin[:, :cp])
                gskip = gxin[:, cp:]
                if e["rs_h"] is not None:
                    gsk = V(self.g[3], B, cs, P)
                    gsk.copy_(gskip)
                    e["rs_h"].adjoint(B * cs, gsk, se["ghs"])
                else:
                    se["ghs"].copy_(gskip)
                gx = gprev
            else:
                gx = gxin
                cur, other = other, cur
        # lifting
        if self.padded:
            gxu = V(self.g[3], B, Ch, P0)
            self._pad_planes(B * Ch, gx, gxu, True)
            gx = gxu
        gz1 = V(self.g[4], B, self.c_lift, P0)
        self._wgrad(B, self.c_lift, Ch, P0, self.a1, gx, lift[1].weight, lift[1].bias)
        _pw_conv(B, Ch, self.c_lift, P0, gx, lift[1].weight, gz1, zmul=self.z1, transpose=True)
        self._wgrad(B, m.in_channels, self.c_lift, P0, self.x_in, gz1, lift[0].weight, lift[0].bias)
        if not self.defer_wgrad_sums:
            self._flush_wgrads()
