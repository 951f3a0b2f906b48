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
#include <string.h>

extern "C" void ppsci_set_error(const char* fmt, ...);

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

__global__ void __launch_bounds__(64) spectral_contract_kernel(SpecArgs a) {
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  int id = blockIdx.x;
  const int nb = id % a.nblk_n;
  id /= a.nblk_n;
  const int tb = id % a.ntile_b;
  const int mode = id / a.ntile_b;
  const int mx = mode / a.d.modes_y, my = mode - mx * a.d.modes_y;
  // forward: reads the input rows, writes the output rows; data gradient (conj_t): the other way round
  const int r_src = a.conj_t ? spec_row_out(a.d, a.c0, mx) : spec_row_in(a.d, a.c0, mx);
  const int r_dst = a.conj_t ? spec_row_in(a.d, a.c0, mx) : spec_row_out(a.d, a.c0, mx);
  const long long plane = (long long)a.d.h * a.d.wf * 2;
  const long long pix = ((long long)r_src * a.d.wf + my) * 2, pix_dst = ((long long)r_dst * a.d.wf + my) * 2;
  const int K = 2 * a.cin;
  const int brow = tb * 16 + c;       // A operand row (batch index) for this lane
  const bool bok = brow < a.d.batch;
  const int jj = nb * 16 + c;         // B operand column for this lane
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += 4) {
    const int kk = k0 + g;  // this lane's k index for A[i=c][k=g] and B[k=g][j=c]
    float av = 0.f, bv = 0.f;
    if (kk < K) {
      const int ch = kk < a.cin ? kk : kk - a.cin, part = kk < a.cin ? 0 : 1;
      if (bok) av = a.x[((long long)brow * a.cin + ch) * plane + pix + part];
      if (jj < 2 * a.cout) bv = spec_w(a, kk, jj, mode);
    }
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
  }
  // D[row = 4g + rr][col = c]: batch row, expanded output column
  if (jj < 2 * a.cout) {
    const int ch = jj < a.cout ? jj : jj - a.cout, part = jj < a.cout ? 0 : 1;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int b = tb * 16 + 4 * g + rr;
      if (b < a.d.batch) a.out[((long long)b * a.cout + ch) * plane + pix_dst + part] = acc[rr] * a.scale;
    }
  }
}

// gw[i,o,mode] = sum_b conj(x[b,i]) * g[b,o]:  gwr = sum xr*gr + xi*gi,  gwi = sum xr*gi - xi*gr
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
};

__global__ void __launch_bounds__(256) spectral_wgrad_kernel(SpecWArgs a) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= a.total) return;
  const int ms = a.d.modes_x * a.d.modes_y;
  const int mode = (int)(t % ms);
  const long long io = t / ms;
  const int o = (int)(io % a.d.c_out), i = (int)(io / a.d.c_out);
  const int mx = mode / a.d.modes_y, my = mode - mx * a.d.modes_y;
  const long long plane = (long long)a.d.h * a.d.wf * 2;
  const long long pix_x = ((long long)spec_row_in(a.d, a.c0, mx) * a.d.wf + my) * 2;
  const long long pix_g = ((long long)spec_row_out(a.d, a.c0, mx) * a.d.wf + my) * 2;
  float sr = 0.f, si = 0.f;
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
                           int conj_t, void* stream, float scale = 1.f) {
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
  const int grid = d->modes_x * d->modes_y * a.ntile_b * a.nblk_n;
  PPSCI_LAUNCH(spectral_contract_kernel, SpecArgs, grid, 64, 0, stream, a);
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
                        void* stream, float xscale = 1.f);

// Clears n floats (n % 2 == 0: complex spectra) with a plain kernel.  Not hipMemsetAsync: as a node of a captured HIP graph
// the runtime's fill path made every replay of the TFNO step host-bound at 3.5 ms (device time 0.98 ms) once the process
// had returned memory to the driver (measured on MI355X / ROCm 7.0.2: bench.py's cfg 4 entry after the 1 M-point run).
struct ZeroArgs {
  float* p;
  long long n;
};
__global__ void __launch_bounds__(256) spectral_zero_kernel(ZeroArgs a) {
  for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 2; i < a.n; i += (long long)gridDim.x * 512) {
    a.p[i] = 0.f;
    a.p[i + 1] = 0.f;
  }
}
static int spectral_zero(float* p, long long n, void* stream) {
  ZeroArgs a{p, n};
  long long grid = (n / 2 + 255) / 256;
  if (grid > 2048) grid = 2048;
  if (grid < 1) grid = 1;
  PPSCI_LAUNCH(spectral_zero_kernel, ZeroArgs, (int)grid, 256, 0, stream, a);
  return PPSCI_LAST_LAUNCH_ERROR() != 0 ? PPSCI_E_LAUNCH : PPSCI_OK;
}

// out_ft = scale * (x_ft . w) on the kept modes, after clearing the WHOLE output spectrum (`zero_fill` != 0): for callers
// whose inverse transform destroys its input (hipFFT C2R, ppsci_fft2d_c2r) or that hand over uninitialised memory.
extern "C" int ppsci_spectral_conv2d_fwd_scaled(const ppsci_spectral_desc* d, const float* x_ft, const float* w_re,
                                                const float* w_im, float* out_ft, float scale, int zero_fill,
                                                void* stream) {
  if (!d || !x_ft || !w_re || !w_im || !out_ft) {
    ppsci_set_error("spectral_conv2d_fwd_scaled: null pointer");
    return PPSCI_E_INVALID;
  }
  if (zero_fill && spectral_zero(out_ft, (long long)d->batch * d->c_out * d->h * d->wf * 2, stream) != PPSCI_OK) {
    ppsci_set_error("spectral_conv2d_fwd_scaled: clearing the spectrum failed");
    return PPSCI_E_LAUNCH;
  }
  return launch_contract(d, x_ft, w_re, w_im, out_ft, 0, stream, scale);
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
  if (zero_fill && spectral_zero(gx_ft, (long long)d->batch * d->c_in * d->h * d->wf * 2, stream) != PPSCI_OK) {
    ppsci_set_error("spectral_conv2d_bwd_real_scaled: clearing the spectrum failed");
    return PPSCI_E_LAUNCH;
  }
  return spectral_bwd(d, x_ft, w_re, w_im, ghat_ft, gx_ft, gw_re, gw_im, wscale, w_full, stream, xscale);
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
                        void* stream, float xscale) {
  if (!x_ft || !w_re || !w_im || !gout_ft) {
    ppsci_set_error("spectral_conv2d_bwd: null pointer");
    return PPSCI_E_INVALID;
  }
  if (gx_ft) {
    int rc = launch_contract(d, gout_ft, w_re, w_im, gx_ft, 1, stream, xscale);
    if (rc != PPSCI_OK) return rc;
  }
  if (gw_re && gw_im) {
    SpecWArgs a;
    memset(&a, 0, sizeof(a));
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
    PPSCI_LAUNCH(spectral_wgrad_kernel, SpecWArgs, (int)((a.total + 255) / 256), 256, 0, stream, a);
    int e = PPSCI_LAST_LAUNCH_ERROR();
    if (e != 0) {
      ppsci_set_error("spectral_conv2d_bwd: launch failed (hip error %d)", e);
      return PPSCI_E_LAUNCH;
    }
  }
  return PPSCI_OK;
}
