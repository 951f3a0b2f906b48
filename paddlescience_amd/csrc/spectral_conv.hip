// spectral_conv.hip -- FNO spectral convolution: per-mode complex channel contraction on MFMA.
//
// Replaces the four real einsums "abcd,becd->aecd" of
//   _contract_dense_trick           /root/reference/ppsci/arch/fno_block.py:346-372
// together with the mode slicing / fftshift bookkeeping of
//   FactorizedSpectralConv.forward  /root/reference/ppsci/arch/fno_block.py:707-796
// (rfftn -> fftshift(dim -2) -> centre-crop n_modes[0] rows x first n_modes[1]//2+1 columns -> contraction
//  -> write into a zero spectrum -> fftshift -> irfftn).  The FFTs themselves stay in hipFFT (torch.fft).
//
// The shift + crop is folded into index arithmetic on the UNSHIFTED spectrum: weight row m of the kept block
// sits at shifted row c0 + m, i.e. unshifted row (c0 + m - H//2) mod H, and the reference's second fftshift puts the
// result to row (c0 + m + H//2) mod H: the same row for even H, the row before it for odd H (spec_row_in / _out).
//
// Per kept mode the complex product  out[b,o] = sum_i x[b,i] * w[i,o]  is one real GEMM
//   [B x 2Ci] . [[wr, wi], [-wi, wr]]  ->  [B x 2Co]
// done with v_mfma_f32_16x16x4_f32: one wave per (mode, 16-row batch tile, 16-column block of [out_r | out_i]).
// At the BASELINE size (B=16, Ci=Co=32, 84 modes) this is 11 MFLOP against ~1.7 MB of operands: the kernel is
// HBM/latency-bound; MFMA is used because the shape is a true dense contraction (north-star), not for speed.
#include "ppsci_common.h"

#ifndef PPSCI_EMU
#include <hip/hip_runtime.h>
#endif
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <tuple>
#include <vector>

extern "C" void ppsci_set_error(const char* fmt, ...);

struct SpecWArgs {
  ppsci_spectral_desc d;
  const float* x;
  const float* g;
  float* gwr;
  float* gwi;
  int c0;
  long long total;
  float wscale;  // see ppsci_spectral_conv2d_bwd_real
  int w_full;    // > 0: g is rfftn(dL/dy): weight gradients get wscale * c(my), c = 1 on the DC / Nyquist columns, else 2
  int compact;   // x / g hold the kept modes only: [B, C, modes_x, modes_y, 2] (ppsci_dft2_kept_*)
};

struct SpecArgs {
  ppsci_spectral_desc d;
  const float* x;   // [B, Cin, H, Wf, 2]   (Cin = c_in for fwd, c_out for bwd_x)
  const float* wr;  // [Ci, Co, Mx, My]
  const float* wi;
  float* out;       // [B, Cout, H, Wf, 2]
  int conj_t;       // 0: out = x . w            (forward)
                    // 1: out = x . conj(w)^T    (gradient w.r.t. the input spectrum)
  int c0, ntile_b, nblk_n, cin, cout;
  float scale;      // applied to the result (1/(H*W) when the FFTs around the contraction are unscaled hipFFT calls)
  int zero_fill;    // > 0: that many EXTRA workgroups (behind the mode workgroups) clear every position of `out` that
                    // no kept mode writes
  int nmode_wg;     // modes x batch tiles
  int compact;      // x / out hold the kept modes only: [B, C, modes_x, modes_y, 2] (ppsci_dft2_kept_*): no rows to map
  int nw_wg;        // > 0: that many further workgroups compute the weight gradient `w` (the backward's second half:
  SpecWArgs w;      // independent of the data gradient, same operands -- one launch instead of two)
  int w_lds;        // weights of the mode staged in LDS
};

// Kept mode m sits in row c0 + m of the fftshift-ed spectrum (c0 = (H - modes_x) // 2).  The reference shifts the input
// spectrum by H // 2, writes the products into the same rows of a cleared buffer and shifts that by H // 2 AGAIN
// (fno_block.py:721, :791 -- fftshift both times, not ifftshift): input row (c0 + m - H // 2) mod H, output row
// (c0 + m + H // 2) mod H.  The same row when H is even; one row apart when H is odd (DomainPadding sizes like 69).
__device__ __forceinline__ int spec_row_in(const ppsci_spectral_desc& d, int c0, int m) {
  return (c0 + m + d.h - d.h / 2) % d.h;
}
__device__ __forceinline__ int spec_row_out(const ppsci_spectral_desc& d, int c0, int m) {
  return (c0 + m + d.h / 2) % d.h;
}

// real-expanded weight element Wexp[kk][jj], kk in [0, 2*cin), jj in [0, 2*cout)
__device__ __forceinline__ float spec_w(const SpecArgs& a, int kk, int jj, int mode_off) {
  const int ci = a.d.c_in, co = a.d.c_out;
  const long long ms = (long long)a.d.modes_x * a.d.modes_y;
  if (!a.conj_t) {
    // rows: [xr(i) | xi(i)], cols: [out_r(o) | out_i(o)]
    const int i = kk < ci ? kk : kk - ci, o = jj < co ? jj : jj - co;
    const long long idx = ((long long)i * co + o) * ms + mode_off;
    if (kk < ci) return jj < co ? a.wr[idx] : a.wi[idx];
    return jj < co ? -a.wi[idx] : a.wr[idx];
  }
  // rows: [gr(o) | gi(o)], cols: [gx_r(i) | gx_i(i)];  gx = g . conj(w)^T
  const int o = kk < co ? kk : kk - co, i = jj < ci ? jj : jj - ci;
  const long long idx = ((long long)i * co + o) * ms + mode_off;
  if (kk < co) return jj < ci ? a.wr[idx] : -a.wi[idx];
  return jj < ci ? a.wi[idx] : a.wr[idx];
}

// gw[i,o,mode] = sum_b conj(x[b,i]) * g[b,o]:  gwr = sum xr*gr + xi*gi,  gwi = sum xr*gi - xi*gr

__device__ __forceinline__ void spec_wgrad_body(const SpecWArgs& a, long long t) {
  if (t >= a.total) return;
  const int ms = a.d.modes_x * a.d.modes_y;
  const int mode = (int)(t % ms);
  const long long io = t / ms;
  const int o = (int)(io % a.d.c_out), i = (int)(io / a.d.c_out);
  const int mx = mode / a.d.modes_y, my = mode - mx * a.d.modes_y;
  const long long plane = a.compact ? (long long)ms * 2 : (long long)a.d.h * a.d.wf * 2;
  const long long pix_x = a.compact ? (long long)mode * 2 : ((long long)spec_row_in(a.d, a.c0, mx) * a.d.wf + my) * 2;
  const long long pix_g = a.compact ? (long long)mode * 2 : ((long long)spec_row_out(a.d, a.c0, mx) * a.d.wf + my) * 2;
  float sr = 0.f, si = 0.f;
#pragma unroll 8
  for (int b = 0; b < a.d.batch; ++b) {
    const float* xp = a.x + ((long long)b * a.d.c_in + i) * plane + pix_x;
    const float* gp = a.g + ((long long)b * a.d.c_out + o) * plane + pix_g;
    const float xr = xp[0], xi = xp[1], gr = gp[0], gi = gp[1];
    sr += xr * gr + xi * gi;
    si += xr * gi - xi * gr;
  }
  if (a.w_full > 0) {
    const float cm = a.wscale * ((my == 0 || 2 * my == a.w_full) ? 1.f : 2.f);
    sr *= cm;
    si *= cm;
  }
  a.gwr[t] = sr;
  a.gwi[t] = si;
}

__global__ void __launch_bounds__(256) spectral_wgrad_kernel(SpecWArgs a) {
  spec_wgrad_body(a, (long long)blockIdx.x * 256 + threadIdx.x);
}

// ONE workgroup (4 waves) per (mode, 16-row batch tile): the mode's operands -- 16 x Cin complex inputs and the Ci x Co
// complex weights, all of them 4 / 8-byte gathers with strides of a whole plane / the mode count -- are fetched by 256
// threads at once into LDS (one memory latency; round 3's one-wave-per-column-block kernel paid one per k-step: 12.8 us
// at the BASELINE shape B 16, 32 -> 32 channels, 84 modes), then wave w runs the MFMA chain of column blocks w, w + 4, ...
// Weights that do not fit LDS (`w_lds` == 0: more than ~ 90 x 90 channels) are read from global memory per k-step.
// With `zero_fill` extra workgroups of the same launch clear every position of the output spectrum that no kept mode
// writes (disjoint from the product stores: no ordering needed) -- no separate clearing launch in front of the
// contraction, and the clearing overlaps the operand gathers of the mode workgroups.
// LDS: xs [16][2 Cin + 1] | wrs [Ci][Co + 1] | wis [Ci][Co + 1]   (odd row strides: conflict-free both ways round)
__device__ __forceinline__ float spec_w_lds(const SpecArgs& a, const float* wrs, const float* wis, int kk, int jj) {
  const int ci = a.d.c_in, co = a.d.c_out, ld = co + 1;
  if (!a.conj_t) {
    const int i = kk < ci ? kk : kk - ci, o = jj < co ? jj : jj - co;
    const int idx = i * ld + o;
    if (kk < ci) return jj < co ? wrs[idx] : wis[idx];
    return jj < co ? -wis[idx] : wrs[idx];
  }
  const int o = kk < co ? kk : kk - co, i = jj < ci ? jj : jj - ci;
  const int idx = i * ld + o;
  if (kk < co) return jj < ci ? wrs[idx] : -wis[idx];
  return jj < ci ? wis[idx] : wrs[idx];
}

__global__ void __launch_bounds__(256) spectral_mode_kernel(SpecArgs a) {
  PPSCI_DYN_SMEM(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
  if ((int)blockIdx.x >= a.nmode_wg) {
    // clearing workgroups: one [h, wf] plane at a time.  Rows that some kept mode writes: ((r - c0 - shift) mod h) <
    // modes_x with the shift of r_dst below; columns my < modes_y.  Disjoint from the product stores: no ordering needed.
    const int shift = a.conj_t ? a.d.h - a.d.h / 2 : a.d.h / 2;
    const int nplane = a.d.batch * a.cout, per = a.d.h * a.d.wf;
    if ((int)blockIdx.x >= a.nmode_wg + a.zero_fill) {
      spec_wgrad_body(a.w, (long long)((int)blockIdx.x - a.nmode_wg - a.zero_fill) * 256 + tid);
      return;
    }
    for (int pl = (int)blockIdx.x - a.nmode_wg; pl < nplane; pl += a.zero_fill) {
      float* o = a.out + (long long)pl * per * 2;
      for (int e = tid; e < per; e += 256) {
        const int r = e / a.d.wf, col = e - r * a.d.wf;
        const int rel = ((r - a.c0 - shift) % a.d.h + a.d.h) % a.d.h;
        if (col < a.d.modes_y && rel < a.d.modes_x) continue;
        o[2 * e] = 0.f;
        o[2 * e + 1] = 0.f;
      }
    }
    return;
  }
  const int tb = (int)blockIdx.x % a.ntile_b, mode = (int)blockIdx.x / a.ntile_b;
  const int mx = mode / a.d.modes_y, my = mode - mx * a.d.modes_y;
  const int r_src = a.conj_t ? spec_row_out(a.d, a.c0, mx) : spec_row_in(a.d, a.c0, mx);
  const int r_dst = a.conj_t ? spec_row_in(a.d, a.c0, mx) : spec_row_out(a.d, a.c0, mx);
  const long long plane = a.compact ? (long long)a.d.modes_x * a.d.modes_y * 2 : (long long)a.d.h * a.d.wf * 2;
  const long long pix = a.compact ? (long long)mode * 2 : ((long long)r_src * a.d.wf + my) * 2;
  const long long pix_dst = a.compact ? (long long)mode * 2 : ((long long)r_dst * a.d.wf + my) * 2;
  const int K = 2 * a.cin, ldx = K + 1, ci = a.d.c_in, co = a.d.c_out, ldw = co + 1;
  float* xs = smem;
  float* wrs = xs + 16 * ldx;
  float* wis = wrs + ci * ldw;
  for (int idx = tid; idx < 16 * a.cin; idx += 256) {
    const int row = idx / a.cin, ch = idx - row * a.cin, b = tb * 16 + row;
    float re = 0.f, im = 0.f;
    if (b < a.d.batch) {
      const float* src = &a.x[((long long)b * a.cin + ch) * plane + pix];
      re = src[0];
      im = src[1];
    }
    xs[row * ldx + ch] = re;
    xs[row * ldx + a.cin + ch] = im;
  }
  const long long ms = (long long)a.d.modes_x * a.d.modes_y;
  for (int idx = tid; a.w_lds && idx < ci * co; idx += 256) {
    const int i = idx / co, o = idx - i * co;
    wrs[i * ldw + o] = a.wr[(long long)idx * ms + mode];
    wis[i * ldw + o] = a.wi[(long long)idx * ms + mode];
  }
  __syncthreads();
  for (int nb = wave; nb < a.nblk_n; nb += 4) {
    const int jj = nb * 16 + c;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 4) {
      const int kk = k0 + g;
      float av = 0.f, bv = 0.f;
      if (kk < K) {
        av = xs[c * ldx + kk];
        if (jj < 2 * a.cout) bv = a.w_lds ? spec_w_lds(a, wrs, wis, kk, jj) : spec_w(a, kk, jj, mode);
      }
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
    }
    if (jj < 2 * a.cout) {
      const int ch = jj < a.cout ? jj : jj - a.cout, part = jj < a.cout ? 0 : 1;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int b = tb * 16 + 4 * g + rr;
        if (b < a.d.batch) a.out[((long long)b * a.cout + ch) * plane + pix_dst + part] = acc[rr] * a.scale;
      }
    }
  }
}

static int spec_check(const ppsci_spectral_desc* d, int* c0) {
  if (!d || d->batch < 1 || d->c_in < 1 || d->c_out < 1 || d->h < 2 || d->wf < 1 || d->modes_x < 1 || d->modes_y < 1 ||
      d->modes_x > d->h || d->modes_y > d->wf) {
    ppsci_set_error("spectral_conv: invalid descriptor");
    return PPSCI_E_INVALID;
  }
  *c0 = (d->h - d->modes_x) / 2;  // `start // 2` of the reference's slice (fno_block.py:749-752), any parity
  return PPSCI_OK;
}

static int launch_contract(const ppsci_spectral_desc* d, const float* x, const float* wr, const float* wi, float* out,
                           int conj_t, void* stream, float scale = 1.f, int zero_fill = 0, const SpecWArgs* wg = nullptr,
                           int compact = 0) {
  SpecArgs a;
  memset(&a, 0, sizeof(a));
  int rc = spec_check(d, &a.c0);
  if (rc != PPSCI_OK) return rc;
  a.d = *d;
  a.x = x;
  a.wr = wr;
  a.wi = wi;
  a.out = out;
  a.conj_t = conj_t;
  a.scale = scale;
  a.cin = conj_t ? d->c_out : d->c_in;
  a.cout = conj_t ? d->c_in : d->c_out;
  a.ntile_b = (d->batch + 15) / 16;
  a.nblk_n = (2 * a.cout + 15) / 16;
  a.nmode_wg = d->modes_x * d->modes_y * a.ntile_b;
  a.compact = compact;
  if (zero_fill && !compact) {
    const long long nplane = (long long)d->batch * a.cout;
    a.zero_fill = (int)(nplane < 4096 ? nplane : 4096);
  }
  const long long lds_x = 16LL * (2 * a.cin + 1) * 4, lds_w = 2LL * d->c_in * (d->c_out + 1) * 4;
  if (lds_x > 64 * 1024) {
    ppsci_set_error("spectral_conv: %d channels: one batch tile of a mode does not fit LDS", a.cin);
    return PPSCI_E_UNSUPPORTED;
  }
  a.w_lds = lds_x + lds_w <= 64 * 1024 ? 1 : 0;
  const long long lds = lds_x + (a.w_lds ? lds_w : 0);
  if (wg) {
    a.w = *wg;
    a.nw_wg = (int)((wg->total + 255) / 256);
  }
  const int grid = a.nmode_wg + a.zero_fill + a.nw_wg;
  if (PPSCI_SET_MAX_LDS(spectral_mode_kernel, (int)lds) != 0) {
    ppsci_set_error("spectral_conv: cannot raise dynamic LDS to %lld B", lds);
    return PPSCI_E_LAUNCH;
  }
  PPSCI_LAUNCH(spectral_mode_kernel, SpecArgs, grid, 256, (int)lds, stream, a);
  int e = PPSCI_LAST_LAUNCH_ERROR();
  if (e != 0) {
    ppsci_set_error("spectral_conv: launch failed (hip error %d)", e);
    return PPSCI_E_LAUNCH;
  }
  return PPSCI_OK;
}

extern "C" int ppsci_spectral_conv2d_fwd(const ppsci_spectral_desc* d, const float* x_ft, const float* w_re,
                                         const float* w_im, float* out_ft, void* stream) {
  if (!x_ft || !w_re || !w_im || !out_ft) {
    ppsci_set_error("spectral_conv2d_fwd: null pointer");
    return PPSCI_E_INVALID;
  }
  return launch_contract(d, x_ft, w_re, w_im, out_ft, 0, stream);
}

static int spectral_bwd(const ppsci_spectral_desc* d, const float* x_ft, const float* w_re, const float* w_im,
                        const float* gout_ft, float* gx_ft, float* gw_re, float* gw_im, float wscale, int w_full,
                        void* stream, float xscale = 1.f, int zero_fill = 0, int compact = 0);

// out_ft = scale * (x_ft . w) on the kept modes; `zero_fill` != 0: every other position of the output spectrum is cleared
// by the same launch -- for callers whose inverse transform destroys its input (hipFFT C2R, ppsci_fft2d_c2r) or that hand
// over uninitialised memory.  (Cleared by kernel stores, not hipMemsetAsync: as a node of a captured HIP graph the
// runtime's fill path made every replay of the TFNO step host-bound at 3.5 ms once the process had returned memory to
// the driver -- measured on MI355X / ROCm 7.0.2.)
extern "C" int ppsci_spectral_conv2d_fwd_scaled(const ppsci_spectral_desc* d, const float* x_ft, const float* w_re,
                                                const float* w_im, float* out_ft, float scale, int zero_fill,
                                                void* stream) {
  if (!d || !x_ft || !w_re || !w_im || !out_ft) {
    ppsci_set_error("spectral_conv2d_fwd_scaled: null pointer");
    return PPSCI_E_INVALID;
  }
  return launch_contract(d, x_ft, w_re, w_im, out_ft, 0, stream, scale, zero_fill ? 1 : 0);
}

// ppsci_spectral_conv2d_bwd_real with the input-spectrum gradient scaled by `xscale` and its buffer cleared first
extern "C" int ppsci_spectral_conv2d_bwd_real_scaled(const ppsci_spectral_desc* d, const float* x_ft, const float* w_re,
                                                     const float* w_im, const float* ghat_ft, float* gx_ft, float* gw_re,
                                                     float* gw_im, float wscale, int w_full, float xscale, int zero_fill,
                                                     void* stream) {
  if (!d || w_full < 1 || !gx_ft) {
    ppsci_set_error("spectral_conv2d_bwd_real_scaled: invalid argument");
    return PPSCI_E_INVALID;
  }
  return spectral_bwd(d, x_ft, w_re, w_im, ghat_ft, gx_ft, gw_re, gw_im, wscale, w_full, stream, xscale, zero_fill ? 1 : 0);
}

extern "C" int ppsci_spectral_conv2d_bwd(const ppsci_spectral_desc* d, const float* x_ft, const float* w_re,
                                         const float* w_im, const float* gout_ft, float* gx_ft, float* gw_re,
                                         float* gw_im, void* stream) {
  return spectral_bwd(d, x_ft, w_re, w_im, gout_ft, gx_ft, gw_re, gw_im, 1.f, 0, stream);
}

// Backward of  y = irfftn(contract(rfftn(x)))  WITHOUT an autograd graph around the FFTs: `ghat_ft` = rfftn(dL/dy)
// (same norm as the forward transforms).  The adjoint of the real-to-complex / complex-to-real pair folds into
//   dL/dx = irfftn(gx_ft),  gx_ft = ghat . conj(w)^T                       (the Hermitian weights c cancel)
//   dL/dw[i,o,m] = wscale * c(m) * sum_b conj(x_ft[b,i,m]) ghat[b,o,m]     (c = 1 on the DC / Nyquist column, else 2;
//                  wscale = inverse-transform scale / forward-transform scale: H*W for norm "forward", 1/(H*W) for
//                  "backward", 1 for "ortho")
extern "C" int ppsci_spectral_conv2d_bwd_real(const ppsci_spectral_desc* d, const float* x_ft, const float* w_re,
                                              const float* w_im, const float* ghat_ft, float* gx_ft, float* gw_re,
                                              float* gw_im, float wscale, int w_full, void* stream) {
  if (w_full < 1) {
    ppsci_set_error("spectral_conv2d_bwd_real: w_full (the real grid width) must be positive");
    return PPSCI_E_INVALID;
  }
  return spectral_bwd(d, x_ft, w_re, w_im, ghat_ft, gx_ft, gw_re, gw_im, wscale, w_full, stream);
}

static int spectral_bwd(const ppsci_spectral_desc* d, const float* x_ft, const float* w_re, const float* w_im,
                        const float* gout_ft, float* gx_ft, float* gw_re, float* gw_im, float wscale, int w_full,
                        void* stream, float xscale, int zero_fill, int compact) {
  if (!x_ft || !w_re || !w_im || !gout_ft) {
    ppsci_set_error("spectral_conv2d_bwd: null pointer");
    return PPSCI_E_INVALID;
  }
  SpecWArgs a;
  memset(&a, 0, sizeof(a));
  const bool want_w = gw_re && gw_im;
  if (want_w) {
    int rc = spec_check(d, &a.c0);
    if (rc != PPSCI_OK) return rc;
    a.d = *d;
    a.x = x_ft;
    a.g = gout_ft;
    a.gwr = gw_re;
    a.gwi = gw_im;
    a.total = (long long)d->c_in * d->c_out * d->modes_x * d->modes_y;
    a.wscale = wscale;
    a.w_full = w_full;
    a.compact = compact;
  }
  if (gx_ft) return launch_contract(d, gout_ft, w_re, w_im, gx_ft, 1, stream, xscale, zero_fill, want_w ? &a : nullptr, compact);
  if (want_w) {
    PPSCI_LAUNCH(spectral_wgrad_kernel, SpecWArgs, (int)((a.total + 255) / 256), 256, 0, stream, a);
    int e = PPSCI_LAST_LAUNCH_ERROR();
    if (e != 0) {
      ppsci_set_error("spectral_conv2d_bwd: launch failed (hip error %d)", e);
      return PPSCI_E_LAUNCH;
    }
  }
  return PPSCI_OK;
}

// ------------------------------------------------------------------------------------------ kept modes only
// The spectral convolution keeps modes_x x modes_y of the H x (W/2+1) coefficients (12 x 7 of 64 x 33 at the BASELINE
// shape) and multiplies everything else by zero.  A library FFT computes -- and the inverse reads -- all of them: 8.6 MB of
// spectrum per transform, plus the clearing of it.  These kernels evaluate the transform pair on the kept modes only, as two
// small dense DFTs per [H, W] plane staged in LDS (row DFT onto the kept columns, then column DFT onto the kept rows; the
// inverse the other way round):  ~160 kflop per plane, the plane is read / written once, the spectrum is 84 numbers.
//   ppsci_dft2_kept_fwd : x [n, H, W] real        -> X [n, modes_x, modes_y] complex = rfftn(x) at the kept modes (unscaled)
//   ppsci_dft2_kept_inv : Z [n, modes_x, modes_y] -> y [n, H, W] real = irfftn of the spectrum that is Z at the kept modes
//                         and zero elsewhere (unscaled; like the library C2R: Re of the DC / Nyquist columns only)
// `rows`: 0 = the rows the reference's slice takes from the shifted INPUT spectrum, 1 = the rows its second fftshift puts
// the products to (spec_row_in / spec_row_out: the same for even H).
#include "dft_kept.h"

__global__ void __launch_bounds__(256) dft2_kept_fwd_kernel(DftArgs a) {
  PPSCI_DYN_SMEM(smem);
  float* tw = smem;
  float* th = tw + 2 * a.W * a.my;
  float* pl = th + 2 * a.H * a.mx;
  const int ldp = a.W + 1;  // (odd row stride: the MFMA's A operand reads 16 rows at one column)
  float* T = pl + a.H * ldp;
  const int tid = threadIdx.x, P = a.H * a.W;
  bool first = true;
  for (int p = blockIdx.x; p < a.n; p += gridDim.x) {
    // the plane's loads (the first 16 per thread: all of a 64 x 64 plane) are requested BEFORE the twiddle table's, so
    // that the two memory round trips of the first plane overlap
    const float* x = a.src + (long long)p * P;
    float xr[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int e = tid + 256 * k;
      xr[k] = e < P ? x[e] : 0.f;
    }
    if (first) dft_twiddles(a, tw);
    first = false;
    __syncthreads();  // (the previous plane's T and partial sums consumed)
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int e = tid + 256 * k;
      if (e < P) pl[e + e / a.W] = xr[k];  // = pl[h * ldp + w]
    }
    for (int e = tid + 4096; e < P; e += 256) pl[e + e / a.W] = x[e];
    __syncthreads();
    dft_fwd_stages(a, tw, th, pl, T, a.dst + (long long)p * a.mx * a.my * 2);
  }
}

__global__ void __launch_bounds__(256) dft2_kept_inv_kernel(DftArgs a) {
  PPSCI_DYN_SMEM(smem);
  float* tw = smem;
  float* th = tw + 2 * a.W * a.my;
  float* Z = th + 2 * a.H * a.mx;
  float* T = Z + 2 * a.mx * a.my;
  float* sred = T + 2 * a.H * a.my;  // 512 floats: the row sums' tree
  const int tid = threadIdx.x, P = a.H * a.W, nm = a.mx * a.my;
  bool first = true;
  for (int p = blockIdx.x; p < a.n; p += gridDim.x) {
    const float* z = a.src + (long long)p * nm * 2;
    float zr[2];  // (requested before the twiddle table, see the forward kernel)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int e = tid + 256 * k;
      zr[k] = e < 2 * nm ? z[e] : 0.f;
    }
    if (first) dft_twiddles(a, tw);
    first = false;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int e = tid + 256 * k;
      if (e < 2 * nm) Z[e] = zr[k];
    }
    for (int e = tid + 512; e < 2 * nm; e += 256) Z[e] = z[e];
    __syncthreads();
    float* y = a.dst + (long long)p * P;
    const float sb = (a.rows_out && a.sbias) ? a.sbias[p % a.C] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    dft_inv_stages(a, tw, th, Z, T, [&](int hh, int w, float val) {
      y[(long long)hh * a.W + w] = val;
      const float u = val + sb;
      s1 += u;
      s2 += u * u;
    });
    if (a.rows_out) {  // the block tail's first pass (gn_rowstats_kernel) for free: the plane is in this workgroup's registers
      __syncthreads();
      sred[tid] = s1;
      sred[256 + tid] = s2;
      __syncthreads();
      for (int wd = 128; wd > 0; wd >>= 1) {
        if (tid < wd) {
          sred[tid] += sred[tid + wd];
          sred[256 + tid] += sred[256 + tid + wd];
        }
        __syncthreads();
      }
      if (tid == 0) {
        a.rows_out[(long long)p * 4 + 0] = sred[0];
        a.rows_out[(long long)p * 4 + 1] = sred[256];
      }
    }
  }
}

// The forward spectral convolution and the inverse transform behind it in ONE launch: a workgroup owns an output plane
// (b, o); it contracts that plane's kept modes itself,
//     Z[m] = scale * sum_i x_k[b, i, m] * w[i, o, m]          (complex; fno_block.py:346-372 `_contract_dense_trick`)
// -- 2 Ci loads of 2 mx my floats, coalesced along the modes, all requested before the first sum: one memory round trip --
// and runs dft2_kept_inv_kernel's stages on it.  The contraction as its own launch (spectral_mode_kernel: 9.2 us at the
// BASELINE shape for 11 MFLOP) was pure launch + gather latency in front of a 10.9 us inverse launch.
#define SPECINV_SPLIT 3  // the channel sum in three interleaved parts per mode (256 threads for 84 modes), added in part order
struct SpecInvArgs {
  DftArgs d;
  const float* x;   // [B, Ci, mx, my, 2]
  const float* wr;  // [Ci, Co, mx, my]
  const float* wi;
  int Ci, Co;
  float scale;
  int conj_t;  // 0: planes (b, o), Z = sum_i x[b, i] w[i, o]   (forward);   1: planes (b, i), Z = sum_o x[b, o] conj(w[i, o])
               // (the data gradient: x = the kept modes of dL/dy) -- either way the weights are read along the modes
  int accumulate;  // y += instead of y = (the data gradient joining the skip branch's share of dL/dx, already in y)
};

__global__ void __launch_bounds__(256) spectral_inv_kernel(SpecInvArgs q) {
  PPSCI_DYN_SMEM(smem);
  const DftArgs& a = q.d;
  float* tw = smem;
  float* th = tw + 2 * a.W * a.my;
  float* Z = th + 2 * a.H * a.mx;
  float* T = Z + 2 * a.mx * a.my;
  float* sred = T + 2 * a.H * a.my;            // 512 floats: the row sums' tree
  float* Zp = sred + 512;                      // [SPECINV_SPLIT][mx my][2]: the parts of the channel sum
  const int tid = threadIdx.x, P = a.H * a.W, nm = a.mx * a.my;
  bool first = true;
  for (int p = blockIdx.x; p < a.n; p += gridDim.x) {
    const int Cp = q.conj_t ? q.Ci : q.Co, Cs = q.conj_t ? q.Co : q.Ci;  // planes per sample / summed channels
    const int b = p / Cp, o = p - b * Cp;
    if (first) dft_twiddles(a, tw);
    first = false;
    __syncthreads();  // (the previous plane's readers of Z / Zp are done)
    for (int idx = tid; idx < SPECINV_SPLIT * nm; idx += 256) {
      const int part = idx / nm, m = idx - part * nm;
      float sr = 0.f, si = 0.f;
#pragma unroll 4
      for (int i = part; i < Cs; i += SPECINV_SPLIT) {
        const float* xp = q.x + ((long long)b * Cs + i) * nm * 2 + 2 * m;
        const long long wi_ = (q.conj_t ? (long long)o * q.Co + i : (long long)i * q.Co + o) * nm + m;
        const float xr = xp[0], xi = xp[1], wr = q.wr[wi_], wim = q.conj_t ? -q.wi[wi_] : q.wi[wi_];
        sr += xr * wr - xi * wim;
        si += xr * wim + xi * wr;
      }
      Zp[(part * nm + m) * 2] = sr;
      Zp[(part * nm + m) * 2 + 1] = si;
    }
    __syncthreads();
    for (int e = tid; e < 2 * nm; e += 256) {
      float z = Zp[e];
#pragma unroll
      for (int part = 1; part < SPECINV_SPLIT; ++part) z += Zp[part * nm * 2 + e];
      Z[e] = q.scale * z;
    }
    __syncthreads();
    float* y = a.dst + (long long)p * P;
    const float sb = (a.rows_out && a.sbias) ? a.sbias[p % a.C] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    dft_inv_stages(a, tw, th, Z, T, [&](int hh, int w, float val) {
      float* yp = &y[(long long)hh * a.W + w];
      if (q.accumulate) val += *yp;
      *yp = val;
      const float u = val + sb;
      s1 += u;
      s2 += u * u;
    });
    if (a.rows_out) {
      __syncthreads();
      sred[tid] = s1;
      sred[256 + tid] = s2;
      __syncthreads();
      for (int wd = 128; wd > 0; wd >>= 1) {
        if (tid < wd) {
          sred[tid] += sred[tid + wd];
          sred[256 + tid] += sred[256 + tid + wd];
        }
        __syncthreads();
      }
      if (tid == 0) {
        a.rows_out[(long long)p * 4 + 0] = sred[0];
        a.rows_out[(long long)p * 4 + 1] = sred[256];
      }
    }
  }
}

static long long dft_lds_bytes(int H, int W, int mx, int my, int inverse) {
  if (!inverse) return 4LL * dft_fwd_lds_floats(H, W, mx, my);
  return 4LL * (2LL * W * my + 2LL * H * mx + 2LL * mx * my + 2LL * H * my + 512);
}

// 1 when the kept-mode transforms take this shape (a plane and its tables fit the LDS of two workgroups per CU)
extern "C" int ppsci_dft2_kept_supported(int H, int W, int modes_x, int modes_y) {
  if (H < 2 || W < 2 || modes_x < 1 || modes_y < 1 || modes_x > H || modes_y > W / 2 + 1) return 0;
  // the largest consumer: the stand-alone forward stages, or the inverse stages next to a plane in the block tail's fused
  // kernels (csrc/fno.hip gn_apply / gn_bwd_apply with the transform inside), each + the 4 KB of static LDS those hold
  const long long fwd = dft_lds_bytes(H, W, modes_x, modes_y, 0), inv = dft_lds_bytes(H, W, modes_x, modes_y, 1) + 4LL * H * (W + 1);
  return (fwd > inv ? fwd : inv) + 4096 <= 64 * 1024 ? 1 : 0;
}

// Twiddle tables per (device, H, W, modes, rows): built on the host in double, uploaded at the first call (an eager one:
// the engine captures HIP graphs from the second step on -- the same life cycle as a library FFT plan), kept for the process.
static std::mutex g_dft_mutex;
static std::map<std::tuple<int, int, int, int, int, int, int, int, int>, float*> g_dft_tabs;
// (Hs, Ws) > 0: the table of a RESOLUTION CHANGE (UNO, ppsci_dft2_kept_*_from) -- mode m stands in the row the reference's second
// fftshift gives it in a spectrum laid out for an Hs x Ws grid; irfftn(s=(H, W)) reads that spectrum's rows [0, H) and columns
// [0, W/2 + 1) as frequencies of the H x W grid and drops the rest (zero twiddles).  ratio: column q also carries
// c_W(q) / c_Ws(q) (the forward direction = the way back through the change, see csrc/uno.hip).
static const float* dft_table(int H, int W, int mx, int my, int rows, int Hs, int Ws, int ratio) {
  int dev = 0;
#ifndef PPSCI_EMU
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
#endif
  std::lock_guard<std::mutex> lock(g_dft_mutex);
  const auto key = std::make_tuple(dev, H, W, mx, my, rows, Hs, Ws, ratio);
  auto it = g_dft_tabs.find(key);
  if (it != g_dft_tabs.end()) return it->second;
  std::vector<float> t(2 * ((size_t)W * my + (size_t)H * mx));
  const double two_pi = 6.283185307179586476925;
  for (int w = 0; w < W; ++w)
    for (int q = 0; q < my; ++q) {
      const double ang = two_pi * (double)((long long)w * q % W) / (double)W;
      double f = 1.0;
      if (Hs > 0) {
        if (q > W / 2) f = 0.0;
        else if (ratio) f = ((q == 0 || 2 * q == W) ? 1.0 : 2.0) / ((q == 0 || 2 * q == Ws) ? 1.0 : 2.0);
      }
      t[2 * ((size_t)w * my + q)] = (float)(f * cos(ang));
      t[2 * ((size_t)w * my + q) + 1] = (float)(f * sin(ang));
    }
  float* th = t.data() + 2 * (size_t)W * my;
  const int Hl = Hs > 0 ? Hs : H;  // the grid whose layout places the modes
  const int c0 = (Hl - mx) / 2, sh = rows ? Hl / 2 : Hl - Hl / 2;  // spec_row_out / spec_row_in
  for (int h = 0; h < H; ++h)
    for (int m = 0; m < mx; ++m) {
      const int k = (c0 + m + sh) % Hl;
      const bool kept = k < H;
      const double ang = two_pi * (double)((long long)h * k % H) / (double)H;
      th[2 * ((size_t)h * mx + m)] = kept ? (float)cos(ang) : 0.f;
      th[2 * ((size_t)h * mx + m) + 1] = kept ? (float)sin(ang) : 0.f;
    }
  float* devp = nullptr;
#ifdef PPSCI_EMU
  devp = (float*)malloc(t.size() * sizeof(float));
  if (devp) memcpy(devp, t.data(), t.size() * sizeof(float));
#else
  if (hipMalloc((void**)&devp, t.size() * sizeof(float)) != hipSuccess) return nullptr;
  if (hipMemcpy(devp, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
#endif
  g_dft_tabs[key] = devp;
  return devp;
}
const float* ppsci_dft_table(int H, int W, int mx, int my, int rows) { return dft_table(H, W, mx, my, rows, 0, 0, 0); }

// the transforms between two grids: the mode counts belong to the OTHER grid and may exceed this one's spectrum
extern "C" int ppsci_dft2_kept_from_supported(int H, int W, int modes_x, int modes_y) {
  if (H < 2 || W < 2 || modes_x < 1 || modes_y < 1) return 0;
  const long long fwd = dft_lds_bytes(H, W, modes_x, modes_y, 0), inv = dft_lds_bytes(H, W, modes_x, modes_y, 1);
  return (fwd > inv ? fwd : inv) + 4096 <= 64 * 1024 ? 1 : 0;
}

static int dft_run(int n, int H, int W, int mx, int my, int rows, const float* src, float* dst, int inverse, void* stream,
                   const float* sbias = nullptr, int C = 1, float* rows_out = nullptr, int Hs = 0, int Ws = 0) {
  if (n < 1 || !src || !dst || (rows != 0 && rows != 1) || (Hs > 0) != (Ws > 0) ||
      !(Hs > 0 ? (mx <= Hs && my <= Ws / 2 + 1 && ppsci_dft2_kept_from_supported(H, W, mx, my)) : ppsci_dft2_kept_supported(H, W, mx, my))) {
    ppsci_set_error("dft2_kept: invalid argument or unsupported shape (%d planes of %d x %d, modes %d x %d)", n, H, W, mx, my);
    return PPSCI_E_INVALID;
  }
  const float* tab = dft_table(H, W, mx, my, rows, Hs, Ws, Hs > 0 && !inverse ? 1 : 0);
  if (!tab) {
    ppsci_set_error("dft2_kept: cannot build the twiddle table");
    return PPSCI_E_LAUNCH;
  }
  DftArgs a{src, dst, tab, n, H, W, mx, my, (H - mx) / 2, rows, sbias, rows_out, C > 0 ? C : 1};
  const long long lds = dft_lds_bytes(H, W, mx, my, inverse);
  int grid = n < 8 * PPSCI_NUM_CU ? n : 8 * PPSCI_NUM_CU;
  int se;
  if (inverse) {
    se = PPSCI_SET_MAX_LDS(dft2_kept_inv_kernel, (int)lds);
    if (se == 0) PPSCI_LAUNCH(dft2_kept_inv_kernel, DftArgs, grid, 256, (int)lds, stream, a);
  } else {
    se = PPSCI_SET_MAX_LDS(dft2_kept_fwd_kernel, (int)lds);
    if (se == 0) PPSCI_LAUNCH(dft2_kept_fwd_kernel, DftArgs, grid, 256, (int)lds, stream, a);
  }
  if (se != 0 || PPSCI_LAST_LAUNCH_ERROR() != 0) {
    ppsci_set_error("dft2_kept: launch failed");
    return PPSCI_E_LAUNCH;
  }
  return PPSCI_OK;
}

extern "C" int ppsci_dft2_kept_fwd(int n, int H, int W, int modes_x, int modes_y, int rows, const float* x, float* X,
                                   void* stream) {
  return dft_run(n, H, W, modes_x, modes_y, rows, x, X, 0, stream);
}
extern "C" int ppsci_dft2_kept_inv(int n, int H, int W, int modes_x, int modes_y, int rows, const float* Z, float* y,
                                   void* stream) {
  return dft_run(n, H, W, modes_x, modes_y, rows, Z, y, 1, stream);
}

extern "C" int ppsci_dft2_kept_fwd_from(int n, int H, int W, int modes_x, int modes_y, int Hs, int Ws, const float* x, float* X,
                                        void* stream) {
  if (Hs < 2 || Ws < 2) {
    ppsci_set_error("dft2_kept_fwd_from: the source grid must be given");
    return PPSCI_E_INVALID;
  }
  return dft_run(n, H, W, modes_x, modes_y, 1, x, X, 0, stream, nullptr, 1, nullptr, Hs, Ws);
}
extern "C" int ppsci_dft2_kept_inv_from(int n, int H, int W, int modes_x, int modes_y, int Hs, int Ws, const float* Z, float* y,
                                        void* stream) {
  if (Hs < 2 || Ws < 2) {
    ppsci_set_error("dft2_kept_inv_from: the source grid must be given");
    return PPSCI_E_INVALID;
  }
  return dft_run(n, H, W, modes_x, modes_y, 1, Z, y, 1, stream, nullptr, 1, nullptr, Hs, Ws);
}

// ppsci_dft2_kept_inv + the first pass of the block tail that consumes y: rows_out[plane][4] gets sum(y + sbias[plane % C])
// and the sum of its square (what gn_rowstats_kernel computes from a second read of y); ppsci_fno_tail_fwd_ex(have_rows = 1)
extern "C" int ppsci_dft2_kept_inv_stats(int n, int H, int W, int modes_x, int modes_y, int rows, const float* Z, float* y,
                                         const float* sbias, int C, float* rows_out, void* stream) {
  if (!rows_out || C < 1) {
    ppsci_set_error("dft2_kept_inv_stats: invalid argument");
    return PPSCI_E_INVALID;
  }
  return dft_run(n, H, W, modes_x, modes_y, rows, Z, y, 1, stream, sbias, C, rows_out);
}

// The contraction and its adjoints on kept-mode spectra [B, C, modes_x, modes_y, 2] (no rows to map, nothing to clear)
extern "C" int ppsci_spectral_conv2d_fwd_kept(const ppsci_spectral_desc* d, const float* x_k, const float* w_re,
                                              const float* w_im, float* out_k, float scale, void* stream) {
  if (!d || !x_k || !w_re || !w_im || !out_k) {
    ppsci_set_error("spectral_conv2d_fwd_kept: null pointer");
    return PPSCI_E_INVALID;
  }
  return launch_contract(d, x_k, w_re, w_im, out_k, 0, stream, scale, 0, nullptr, 1);
}
extern "C" int ppsci_spectral_conv2d_bwd_kept(const ppsci_spectral_desc* d, const float* x_k, const float* w_re,
                                              const float* w_im, const float* ghat_k, float* gx_k, float* gw_re,
                                              float* gw_im, float wscale, int w_full, float xscale, void* stream) {
  if (!d || w_full < 1) {
    ppsci_set_error("spectral_conv2d_bwd_kept: invalid argument");
    return PPSCI_E_INVALID;
  }
  return spectral_bwd(d, x_k, w_re, w_im, ghat_k, gx_k, gw_re, gw_im, wscale, w_full, stream, xscale, 0, 1);
}


// ppsci_spectral_conv2d_fwd_kept + ppsci_dft2_kept_inv[_stats] in one launch (spectral_inv_kernel): y [B * c_out planes of H x W].
// _ex: (Hs, Ws) > 0 -- the inverse between two grids (ppsci_dft2_kept_inv_from); conj_t -- the data gradient: x_k = the kept modes of
// dL/dy [B, c_out], y [B * c_in planes] = inverse transform of x_k . conj(w)^T (ppsci_spectral_conv2d_bwd_kept's gx_k + ppsci_dft2_kept_inv).
extern "C" int ppsci_spectral_conv2d_inv_kept_ex(const ppsci_spectral_desc* d, int H, int W, int Hs, int Ws, int conj_t, int rows,
                                                 const float* x_k, const float* w_re, const float* w_im, float scale, float* y,
                                                 const float* sbias, float* rows_out, void* stream) {
  if (!d || !x_k || !w_re || !w_im || !y || (rows != 0 && rows != 1) || d->batch < 1 || d->c_in < 1 || d->c_out < 1 ||
      (Hs > 0) != (Ws > 0) || Hs < 0 ||
      !(Hs > 0 ? (rows == 1 && d->modes_x <= Hs && d->modes_y <= Ws / 2 + 1 && ppsci_dft2_kept_from_supported(H, W, d->modes_x, d->modes_y))
               : ppsci_dft2_kept_supported(H, W, d->modes_x, d->modes_y))) {
    ppsci_set_error("spectral_conv2d_inv_kept: invalid argument or unsupported shape");
    return PPSCI_E_INVALID;
  }
  const int mx = d->modes_x, my = d->modes_y, n = d->batch * ((conj_t & 1) ? d->c_in : d->c_out);
  const float* tab = dft_table(H, W, mx, my, rows, Hs, Ws, 0);
  if (!tab) {
    ppsci_set_error("spectral_conv2d_inv_kept: cannot build the twiddle table");
    return PPSCI_E_LAUNCH;
  }
  SpecInvArgs q;
  q.d = DftArgs{nullptr, y, tab, n, H, W, mx, my, (H - mx) / 2, rows, sbias, rows_out, (conj_t & 1) ? d->c_in : d->c_out};
  q.x = x_k; q.wr = w_re; q.wi = w_im; q.Ci = d->c_in; q.Co = d->c_out; q.scale = scale; q.conj_t = (conj_t & 1) ? 1 : 0; q.accumulate = (conj_t & 2) ? 1 : 0;
  const long long lds = dft_lds_bytes(H, W, mx, my, 1) + 4LL * SPECINV_SPLIT * 2 * mx * my;
  if (lds + 4096 > 64 * 1024) {
    ppsci_set_error("spectral_conv2d_inv_kept: %lld B of LDS", lds);
    return PPSCI_E_UNSUPPORTED;
  }
  const int grid = n < 8 * PPSCI_NUM_CU ? n : 8 * PPSCI_NUM_CU;
  if (PPSCI_SET_MAX_LDS(spectral_inv_kernel, (int)lds) != 0) {
    ppsci_set_error("spectral_conv2d_inv_kept: cannot raise dynamic LDS to %lld B", lds);
    return PPSCI_E_LAUNCH;
  }
  PPSCI_LAUNCH(spectral_inv_kernel, SpecInvArgs, grid, 256, (int)lds, stream, q);
  if (PPSCI_LAST_LAUNCH_ERROR() != 0) {
    ppsci_set_error("spectral_conv2d_inv_kept: launch failed");
    return PPSCI_E_LAUNCH;
  }
  return PPSCI_OK;
}
extern "C" int ppsci_spectral_conv2d_inv_kept(const ppsci_spectral_desc* d, int H, int W, int rows, const float* x_k,
                                              const float* w_re, const float* w_im, float scale, float* y, const float* sbias,
                                              float* rows_out, void* stream) {
  return ppsci_spectral_conv2d_inv_kept_ex(d, H, W, 0, 0, 0, rows, x_k, w_re, w_im, scale, y, sbias, rows_out, stream);
}
