// pirate.hip -- the pointwise stages of ppsci.arch.PirateNet (/root/reference/ppsci/arch/mlp.py:530-820) on Taylor
// streams, forward AND hand-written reverse.  PirateNet is run LAYER BY LAYER (its gates multiply streams of three
// different tensors: it does not fit the register-resident single-kernel MLP sweep of taylor_fwd / taylor_bwd):
//
//   x0      = [cos(B e) ; sin(B e)],  e = the (period-embedded) inputs            FourierEmbedding  mlp.py:117-136
//   U, V    = act(W_u x0 + b_u), act(W_v x0 + b_v)                                  embed_u / embed_v mlp.py:706-745
//   block:  f = act(W1 x + b1);  z1 = f*U + (1-f)*V                                 PirateNetBlock    mlp.py:614-621
//           g = act(W2 z1 + b2); z2 = g*U + (1-g)*V
//           h = act(W3 z2 + b3); x' = alpha*h + (1-alpha)*x
//   y       = W_L x + b_L
//
// Every tensor is a stream block [S][C][NP]: S = 1 + n1 + n2 Taylor streams (value, first derivatives along n1
// directions, second derivatives along the first n2 of them), C features, NP = N rounded up to 16 points (zero
// padding) -- which is the [B, C, P] layout of the 1x1-convolution MFMA GEMMs of fno.hip: a dense layer on all
// streams is ONE ppsci_pw_conv call with B = S (the bias belongs to the value stream only and is added here, in the
// activation stage).  This file holds what sits between the GEMMs:
//
//   ppsci_pirate_embed_fwd / _bwd   inputs -> x0 streams (period + Fourier embedding); gradient of B
//   ppsci_pirate_act_fwd / _bwd     bias + activation (Faa di Bruno to order 2), fused with the gate (Leibniz) or with
//                                   the residual connection; reverse: zbar, Ubar / Vbar accumulation, the skip branch,
//                                   per-chunk partial sums of the bias and alpha gradients
//   ppsci_pirate_out_fwd / _bwd     [S][m][NP] <-> the U / Ubar row blocks [m*S][N] the residual epilogue works on
//
// One thread = one (feature, point): consecutive lanes = consecutive points (coalesced); all sums are block trees
// in LDS with a fixed shape (bit-reproducible, no atomics).
#include "ppsci_common.h"
#include "ppsci_hip.h"

#ifndef PPSCI_EMU
#include <hip/hip_runtime.h>
#endif
#include <math.h>
#include <string.h>

#include "taylor_tile.h"

extern "C" void ppsci_set_error(const char* fmt, ...);

#define PR_BLOCK 256
#define PR_EMB_PTS 1024  // points per workgroup of the embedding reverse kernel

__device__ __forceinline__ void pr_act(int act, float z, float& s, float& d1, float& d2, float& d3) {
  switch (act) {
    case PPSCI_ACT_TANH: ppsci_act_eval<PPSCI_ACT_TANH>(z, s, d1, d2, d3); break;
    case PPSCI_ACT_SILU: ppsci_act_eval<PPSCI_ACT_SILU>(z, s, d1, d2, d3); break;
    case PPSCI_ACT_SIGMOID: ppsci_act_eval<PPSCI_ACT_SIGMOID>(z, s, d1, d2, d3); break;
    case PPSCI_ACT_COS: ppsci_act_eval<PPSCI_ACT_COS>(z, s, d1, d2, d3); break;
    case PPSCI_ACT_GELU: ppsci_act_eval<PPSCI_ACT_GELU>(z, s, d1, d2, d3); break;
    default: ppsci_act_eval<PPSCI_ACT_SIN>(z, s, d1, d2, d3); break;
  }
}

// fixed-shape tree sum over the workgroup; result valid in thread 0
__device__ __forceinline__ float pr_block_sum(float v, float* red) {
  __syncthreads();
  red[threadIdx.x] = v;
  __syncthreads();
  for (int w = PR_BLOCK / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  return red[0];
}

// ------------------------------------------------------------------------------------------ embedding
struct PrEmbArgs {
  ppsci_pirate_embed_desc d;
  const float* x[PPSCI_MAX_IN];
  const float* B;     // [d0][half]
  float* X;           // fwd: [S][2*half][NP]
  const float* Xbar;  // bwd
  float* pB;          // bwd: [chunks][d0*half]
};

// streams (value, first[a], second[a]) of the d0 embedded features of point p
__device__ __forceinline__ void pr_features(const PrEmbArgs& a, long long p, float (&e0)[2 * PPSCI_MAX_IN],
                                            float (&e1)[2 * PPSCI_MAX_IN][PPSCI_MAX_DIRS],
                                            float (&e2)[2 * PPSCI_MAX_IN][PPSCI_MAX_DIRS]) {
  int k = 0;
  for (int j = 0; j < a.d.d_raw; ++j) {
    const float xv = a.x[j][p];
    if (a.d.embed[j] == PPSCI_EMBED_PERIOD) {  // PeriodEmbedding: [cos(w x), sin(w x)] (mlp.py:108-114)
      const float w = a.d.omega[j], cs = cosf(w * xv), sn = sinf(w * xv);
      e0[k] = cs; e0[k + 1] = sn;
      for (int q = 0; q < a.d.n1; ++q) {
        const float dj = a.d.dirs[q][j] * w;
        e1[k][q] = -sn * dj; e1[k + 1][q] = cs * dj;
        e2[k][q] = -cs * dj * dj; e2[k + 1][q] = -sn * dj * dj;
      }
      k += 2;
    } else {
      e0[k] = xv;
      for (int q = 0; q < a.d.n1; ++q) { e1[k][q] = a.d.dirs[q][j]; e2[k][q] = 0.f; }
      k += 1;
    }
  }
}

__global__ void __launch_bounds__(PR_BLOCK) pirate_embed_fwd_kernel(PrEmbArgs a) {
  const int nch = (int)((a.d.NP + PR_BLOCK - 1) / PR_BLOCK);
  const int m = blockIdx.x / nch;
  const long long p = (long long)(blockIdx.x % nch) * PR_BLOCK + threadIdx.x;
  if (p >= a.d.NP) return;
  const int half = a.d.half, n1 = a.d.n1, n2 = a.d.n2, C = 2 * half;
  if (half == 0) {  // no Fourier embedding (ModifiedMLP without `fourier`): x0 = the d0 embedded features themselves
    const long long NP0 = a.d.NP, plane0 = (long long)a.d.d0 * NP0;
    float* o = a.X + (long long)m * NP0 + p;
    if (p >= a.d.N) {
      for (int s = 0; s < 1 + n1 + n2; ++s) o[s * plane0] = 0.f;
      return;
    }
    float e0[2 * PPSCI_MAX_IN], e1[2 * PPSCI_MAX_IN][PPSCI_MAX_DIRS], e2[2 * PPSCI_MAX_IN][PPSCI_MAX_DIRS];
    pr_features(a, p, e0, e1, e2);
    float v0 = 0.f, v1[PPSCI_MAX_DIRS] = {0, 0, 0, 0}, v2[PPSCI_MAX_DIRS] = {0, 0, 0, 0};
    for (int k = 0; k < a.d.d0; ++k)  // select feature m without dynamic register indexing
      if (k == m) {
        v0 = e0[k];
        for (int q = 0; q < n1; ++q) { v1[q] = e1[k][q]; v2[q] = e2[k][q]; }
      }
    o[0] = v0;
    for (int q = 0; q < n1; ++q) o[(1 + q) * plane0] = v1[q];
    for (int q = 0; q < n2; ++q) o[(1 + n1 + q) * plane0] = v2[q];
    return;
  }
  const long long NP = a.d.NP, plane = (long long)C * NP;
  float* oc = a.X + (long long)m * NP + p;            // cos feature m
  float* os = a.X + (long long)(half + m) * NP + p;   // sin feature half + m
  if (p >= a.d.N) {
    for (int s = 0; s < 1 + n1 + n2; ++s) { oc[s * plane] = 0.f; os[s * plane] = 0.f; }
    return;
  }
  float e0[2 * PPSCI_MAX_IN], e1[2 * PPSCI_MAX_IN][PPSCI_MAX_DIRS], e2[2 * PPSCI_MAX_IN][PPSCI_MAX_DIRS];
  pr_features(a, p, e0, e1, e2);
  float t0 = 0.f, t1[PPSCI_MAX_DIRS] = {0, 0, 0, 0}, t2[PPSCI_MAX_DIRS] = {0, 0, 0, 0};
  for (int k = 0; k < a.d.d0; ++k) {
    const float b = a.B[k * half + m];
    t0 += b * e0[k];
    for (int q = 0; q < n1; ++q) { t1[q] += b * e1[k][q]; t2[q] += b * e2[k][q]; }
  }
  const float cs = cosf(t0), sn = sinf(t0);
  oc[0] = cs; os[0] = sn;
  for (int q = 0; q < n1; ++q) {
    oc[(1 + q) * plane] = -sn * t1[q];
    os[(1 + q) * plane] = cs * t1[q];
  }
  for (int q = 0; q < n2; ++q) {
    oc[(1 + n1 + q) * plane] = -cs * t1[q] * t1[q] - sn * t2[q];
    os[(1 + n1 + q) * plane] = -sn * t1[q] * t1[q] + cs * t2[q];
  }
}

// Bbar[k][m] = sum_p sum_s thetabar_s(m, p) e_k,s(p): one workgroup = (feature m, PR_EMB_PTS points)
__global__ void __launch_bounds__(PR_BLOCK) pirate_embed_bwd_kernel(PrEmbArgs a) {
  __shared__ float red[PR_BLOCK];
  const int nch = (int)((a.d.N + PR_EMB_PTS - 1) / PR_EMB_PTS);
  const int m = blockIdx.x / nch, ch = blockIdx.x % nch;
  const int half = a.d.half, n1 = a.d.n1, n2 = a.d.n2, C = 2 * half, d0 = a.d.d0;
  const long long NP = a.d.NP, plane = (long long)C * NP;
  float acc[2 * PPSCI_MAX_IN];
  for (int k = 0; k < 2 * PPSCI_MAX_IN; ++k) acc[k] = 0.f;
  for (int it = 0; it < PR_EMB_PTS / PR_BLOCK; ++it) {
    const long long p = (long long)ch * PR_EMB_PTS + it * PR_BLOCK + threadIdx.x;
    if (p >= a.d.N) continue;
    float e0[2 * PPSCI_MAX_IN], e1[2 * PPSCI_MAX_IN][PPSCI_MAX_DIRS], e2[2 * PPSCI_MAX_IN][PPSCI_MAX_DIRS];
    pr_features(a, p, e0, e1, e2);
    float t0 = 0.f, t1[PPSCI_MAX_DIRS] = {0, 0, 0, 0};
    for (int k = 0; k < d0; ++k) {
      const float b = a.B[k * half + m];
      t0 += b * e0[k];
      for (int q = 0; q < n1; ++q) t1[q] += b * e1[k][q];
    }
    float t2[PPSCI_MAX_DIRS] = {0, 0, 0, 0};
    for (int k = 0; k < d0; ++k) {
      const float b = a.B[k * half + m];
      for (int q = 0; q < n2; ++q) t2[q] += b * e2[k][q];
    }
    const float cs = cosf(t0), sn = sinf(t0);
    const float* gc = a.Xbar + (long long)m * NP + p;
    const float* gs = a.Xbar + (long long)(half + m) * NP + p;
    // adjoints of theta's streams
    float tb0 = -sn * gc[0] + cs * gs[0], tb1[PPSCI_MAX_DIRS], tb2[PPSCI_MAX_DIRS];
    for (int q = 0; q < n1; ++q) {
      const float c1 = gc[(1 + q) * plane], s1 = gs[(1 + q) * plane];
      tb1[q] = -sn * c1 + cs * s1;
      tb0 += -cs * t1[q] * c1 - sn * t1[q] * s1;
      tb2[q] = 0.f;
    }
    for (int q = 0; q < n2; ++q) {
      const float c2 = gc[(1 + n1 + q) * plane], s2 = gs[(1 + n1 + q) * plane];
      tb2[q] = -sn * c2 + cs * s2;
      tb1[q] += 2.f * t1[q] * (-cs * c2 - sn * s2);
      tb0 += (sn * t1[q] * t1[q] - cs * t2[q]) * c2 + (-cs * t1[q] * t1[q] - sn * t2[q]) * s2;
    }
    for (int k = 0; k < d0; ++k) {
      float v = tb0 * e0[k];
      for (int q = 0; q < n1; ++q) v += tb1[q] * e1[k][q];
      for (int q = 0; q < n2; ++q) v += tb2[q] * e2[k][q];
      acc[k] += v;
    }
  }
  for (int k = 0; k < d0; ++k) {
    const float s = pr_block_sum(acc[k], red);
    if (threadIdx.x == 0) a.pB[(long long)ch * d0 * half + k * half + m] = s;
  }
}

// ------------------------------------------------------------------------------------------ bias + activation (+ gate / residual)
struct PrActArgs {
  const float* z;      // [S][H][NP] pre-activation WITHOUT the bias
  const float* bias;   // [H]
  const float* U;      // GATE
  const float* V;
  const float* x;      // RES: the block input
  const float* alpha;  // RES: device scalar
  float* out;          // fwd
  const float* obar;   // bwd: adjoint of `out`
  float* zbar;         // bwd out
  float* Ubar;         // bwd accumulate
  float* Vbar;
  float* xbar;         // bwd out (RES): (1 - alpha) * obar
  float* pb;           // bwd: [chunks][H] partial bias gradient
  float* palpha;       // bwd (RES): [H*chunks] partial alpha gradient
  int H, n1, n2, act, mode;
  long long N, NP;
};

__global__ void __launch_bounds__(PR_BLOCK) pirate_act_fwd_kernel(PrActArgs a) {
  const int nch = (int)((a.NP + PR_BLOCK - 1) / PR_BLOCK);
  const int h = blockIdx.x / nch;
  const long long p = (long long)(blockIdx.x % nch) * PR_BLOCK + threadIdx.x;
  if (p >= a.NP) return;
  const int n1 = a.n1, n2 = a.n2, S = 1 + n1 + n2;
  const long long plane = (long long)a.H * a.NP, e = (long long)h * a.NP + p;
  if (p >= a.N) {
    for (int s = 0; s < S; ++s) a.out[s * plane + e] = 0.f;
    return;
  }
  float s0, d1, d2, d3;
  pr_act(a.act, a.z[e] + a.bias[h], s0, d1, d2, d3);
  float ap[PPSCI_MAX_DIRS], app[PPSCI_MAX_DIRS];
  for (int q = 0; q < n1; ++q) {
    const float zp = a.z[(1 + q) * plane + e];
    ap[q] = d1 * zp;
    app[q] = q < n2 ? d2 * zp * zp + d1 * a.z[(1 + n1 + q) * plane + e] : 0.f;
  }
  if (a.mode == PPSCI_PIRATE_ACT) {
    a.out[e] = s0;
    for (int q = 0; q < n1; ++q) a.out[(1 + q) * plane + e] = ap[q];
    for (int q = 0; q < n2; ++q) a.out[(1 + n1 + q) * plane + e] = app[q];
  } else if (a.mode == PPSCI_PIRATE_GATE) {  // o = V + a (U - V), Leibniz
    const float V0 = a.V[e], D0 = a.U[e] - V0;
    a.out[e] = V0 + s0 * D0;
    for (int q = 0; q < n1; ++q) {
      const float V1 = a.V[(1 + q) * plane + e], D1 = a.U[(1 + q) * plane + e] - V1;
      a.out[(1 + q) * plane + e] = V1 + ap[q] * D0 + s0 * D1;
      if (q < n2) {
        const float V2 = a.V[(1 + n1 + q) * plane + e], D2 = a.U[(1 + n1 + q) * plane + e] - V2;
        a.out[(1 + n1 + q) * plane + e] = V2 + app[q] * D0 + 2.f * ap[q] * D1 + s0 * D2;
      }
    }
  } else {  // x' = alpha h + (1 - alpha) x
    const float al = a.alpha[0], be = 1.f - al;
    a.out[e] = al * s0 + be * a.x[e];
    for (int q = 0; q < n1; ++q) a.out[(1 + q) * plane + e] = al * ap[q] + be * a.x[(1 + q) * plane + e];
    for (int q = 0; q < n2; ++q) a.out[(1 + n1 + q) * plane + e] = al * app[q] + be * a.x[(1 + n1 + q) * plane + e];
  }
}

__global__ void __launch_bounds__(PR_BLOCK) pirate_act_bwd_kernel(PrActArgs a) {
  __shared__ float red[PR_BLOCK];
  const int nch = (int)((a.NP + PR_BLOCK - 1) / PR_BLOCK);
  const int h = blockIdx.x / nch, ch = blockIdx.x % nch;
  const long long p = (long long)ch * PR_BLOCK + threadIdx.x;
  const int n1 = a.n1, n2 = a.n2, S = 1 + n1 + n2;
  const long long plane = (long long)a.H * a.NP, e = (long long)h * a.NP + p;
  float zb0 = 0.f, galpha = 0.f;
  if (p < a.NP && p >= a.N) {
    for (int s = 0; s < S; ++s) {
      a.zbar[s * plane + e] = 0.f;
      if (a.mode == PPSCI_PIRATE_RES) a.xbar[s * plane + e] = 0.f;
    }
  } else if (p < a.N) {
    float s0, d1, d2, d3;
    pr_act(a.act, a.z[e] + a.bias[h], s0, d1, d2, d3);
    float zp[PPSCI_MAX_DIRS], zpp[PPSCI_MAX_DIRS], ap[PPSCI_MAX_DIRS], app[PPSCI_MAX_DIRS];
    for (int q = 0; q < n1; ++q) {
      zp[q] = a.z[(1 + q) * plane + e];
      zpp[q] = q < n2 ? a.z[(1 + n1 + q) * plane + e] : 0.f;
      ap[q] = d1 * zp[q];
      app[q] = q < n2 ? d2 * zp[q] * zp[q] + d1 * zpp[q] : 0.f;
    }
    float ob0 = a.obar[e], ob1[PPSCI_MAX_DIRS], ob2[PPSCI_MAX_DIRS];
    for (int q = 0; q < n1; ++q) {
      ob1[q] = a.obar[(1 + q) * plane + e];
      ob2[q] = q < n2 ? a.obar[(1 + n1 + q) * plane + e] : 0.f;
    }
    // adjoint of the activation streams a = (s0, ap, app)
    float ab0, ab1[PPSCI_MAX_DIRS], ab2[PPSCI_MAX_DIRS];
    if (a.mode == PPSCI_PIRATE_ACT) {
      ab0 = ob0;
      for (int q = 0; q < n1; ++q) { ab1[q] = ob1[q]; ab2[q] = ob2[q]; }
    } else if (a.mode == PPSCI_PIRATE_GATE) {
      const float V0 = a.V[e], D0 = a.U[e] - V0;
      ab0 = ob0 * D0;
      float Db0 = ob0 * s0;
      for (int q = 0; q < n1; ++q) {
        const float D1 = a.U[(1 + q) * plane + e] - a.V[(1 + q) * plane + e];
        float D2 = 0.f;
        if (q < n2) D2 = a.U[(1 + n1 + q) * plane + e] - a.V[(1 + n1 + q) * plane + e];
        ab0 += ob1[q] * D1 + ob2[q] * D2;
        ab1[q] = ob1[q] * D0 + 2.f * ob2[q] * D1;
        ab2[q] = ob2[q] * D0;
        Db0 += ob1[q] * ap[q] + ob2[q] * app[q];
        const float Db1 = ob1[q] * s0 + 2.f * ob2[q] * ap[q];
        a.Ubar[(1 + q) * plane + e] += Db1;
        a.Vbar[(1 + q) * plane + e] += ob1[q] - Db1;
        if (q < n2) {
          const float Db2 = ob2[q] * s0;
          a.Ubar[(1 + n1 + q) * plane + e] += Db2;
          a.Vbar[(1 + n1 + q) * plane + e] += ob2[q] - Db2;
        }
      }
      a.Ubar[e] += Db0;
      a.Vbar[e] += ob0 - Db0;
    } else {
      const float al = a.alpha[0], be = 1.f - al;
      ab0 = al * ob0;
      a.xbar[e] = be * ob0;
      galpha = (s0 - a.x[e]) * ob0;
      for (int q = 0; q < n1; ++q) {
        ab1[q] = al * ob1[q];
        a.xbar[(1 + q) * plane + e] = be * ob1[q];
        galpha += (ap[q] - a.x[(1 + q) * plane + e]) * ob1[q];
        ab2[q] = al * ob2[q];
        if (q < n2) {
          a.xbar[(1 + n1 + q) * plane + e] = be * ob2[q];
          galpha += (app[q] - a.x[(1 + n1 + q) * plane + e]) * ob2[q];
        }
      }
    }
    // adjoint of z's streams (Faa di Bruno, order 2)
    zb0 = d1 * ab0;
    for (int q = 0; q < n1; ++q) {
      float z1b = d1 * ab1[q];
      zb0 += d2 * zp[q] * ab1[q];
      if (q < n2) {
        z1b += 2.f * d2 * zp[q] * ab2[q];
        zb0 += (d3 * zp[q] * zp[q] + d2 * zpp[q]) * ab2[q];
        a.zbar[(1 + n1 + q) * plane + e] = d1 * ab2[q];
      }
      a.zbar[(1 + q) * plane + e] = z1b;
    }
    a.zbar[e] = zb0;
  }
  const float sb = pr_block_sum(zb0, red);
  if (threadIdx.x == 0) a.pb[(long long)ch * a.H + h] = sb;
  if (a.mode == PPSCI_PIRATE_RES) {
    const float sa = pr_block_sum(galpha, red);
    if (threadIdx.x == 0) a.palpha[blockIdx.x] = sa;
  }
}

// ------------------------------------------------------------------------------------------ output rows
struct PrOutArgs {
  const float* src;
  const float* bias;
  float* dst;
  int S, m;
  long long N, NP;
};

__global__ void __launch_bounds__(PR_BLOCK) pirate_out_fwd_kernel(PrOutArgs a) {  // Y [S][m][NP] (+ bias) -> U [m*S][N]
  const long long t = (long long)blockIdx.x * PR_BLOCK + threadIdx.x;
  if (t >= (long long)a.S * a.m * a.N) return;
  const long long p = t % a.N;
  const int row = (int)(t / a.N), o = row / a.S, s = row % a.S;
  a.dst[t] = a.src[((long long)s * a.m + o) * a.NP + p] + (s == 0 ? a.bias[o] : 0.f);
}

__global__ void __launch_bounds__(PR_BLOCK) pirate_out_bwd_kernel(PrOutArgs a) {  // Ubar [m*S][N] -> Ybar [S][m][NP]
  const long long t = (long long)blockIdx.x * PR_BLOCK + threadIdx.x;
  if (t >= (long long)a.S * a.m * a.NP) return;
  const long long p = t % a.NP;
  const int row = (int)(t / a.NP), s = row / a.m, o = row % a.m;
  a.dst[t] = p < a.N ? a.src[((long long)o * a.S + s) * a.N + p] : 0.f;
}

// ------------------------------------------------------------------------------------------ C ABI
static int emb_check(const ppsci_pirate_embed_desc* d) {
  if (!d || d->d_raw < 1 || d->d_raw > PPSCI_MAX_IN || d->d0 < d->d_raw || d->d0 > 2 * PPSCI_MAX_IN || d->half < 0 ||
      d->n1 < 0 || d->n1 > PPSCI_MAX_DIRS || d->n2 < 0 || d->n2 > d->n1 || d->N < 1 || d->NP < d->N || (d->NP & 15)) {
    ppsci_set_error("pirate_embed: invalid descriptor");
    return PPSCI_E_INVALID;
  }
  return PPSCI_OK;
}

extern "C" int64_t ppsci_pirate_embed_chunks(int64_t N) { return (N + PR_EMB_PTS - 1) / PR_EMB_PTS; }
extern "C" int64_t ppsci_pirate_act_chunks(int64_t NP) { return (NP + PR_BLOCK - 1) / PR_BLOCK; }

extern "C" int ppsci_pirate_embed_fwd(const ppsci_pirate_embed_desc* d, const float* const* inputs_host, const float* B,
                                      float* X, void* stream) {
  if (emb_check(d) != PPSCI_OK) return PPSCI_E_INVALID;
  if (!inputs_host || (!B && d->half > 0) || !X) { ppsci_set_error("pirate_embed_fwd: null argument"); return PPSCI_E_INVALID; }
  PrEmbArgs a;
  memset(&a, 0, sizeof(a));
  a.d = *d;
  for (int j = 0; j < d->d_raw; ++j) a.x[j] = inputs_host[j];
  a.B = B; a.X = X;
  const long long grid = (long long)(d->half > 0 ? d->half : d->d0) * ((d->NP + PR_BLOCK - 1) / PR_BLOCK);
  PPSCI_LAUNCH(pirate_embed_fwd_kernel, PrEmbArgs, (int)grid, PR_BLOCK, 0, stream, a);
  int e = PPSCI_LAST_LAUNCH_ERROR();
  if (e != 0) { ppsci_set_error("pirate_embed_fwd: launch failed (%d)", e); return PPSCI_E_LAUNCH; }
  return PPSCI_OK;
}

extern "C" int ppsci_pirate_embed_bwd(const ppsci_pirate_embed_desc* d, const float* const* inputs_host, const float* B,
                                      const float* Xbar, float* partials, void* stream) {
  if (emb_check(d) != PPSCI_OK) return PPSCI_E_INVALID;
  if (!inputs_host || !B || !Xbar || !partials || d->half < 1) { ppsci_set_error("pirate_embed_bwd: null argument"); return PPSCI_E_INVALID; }
  PrEmbArgs a;
  memset(&a, 0, sizeof(a));
  a.d = *d;
  for (int j = 0; j < d->d_raw; ++j) a.x[j] = inputs_host[j];
  a.B = B; a.Xbar = Xbar; a.pB = partials;
  const long long grid = (long long)d->half * ppsci_pirate_embed_chunks(d->N);
  PPSCI_LAUNCH(pirate_embed_bwd_kernel, PrEmbArgs, (int)grid, PR_BLOCK, 0, stream, a);
  int e = PPSCI_LAST_LAUNCH_ERROR();
  if (e != 0) { ppsci_set_error("pirate_embed_bwd: launch failed (%d)", e); return PPSCI_E_LAUNCH; }
  return PPSCI_OK;
}

static int act_check(int mode, int act, int H, int64_t N, int64_t NP, int n1, int n2) {
  if (mode < PPSCI_PIRATE_ACT || mode > PPSCI_PIRATE_RES || H < 1 || N < 1 || NP < N || (NP & 15) || n1 < 0 ||
      n1 > PPSCI_MAX_DIRS || n2 < 0 || n2 > n1) {
    ppsci_set_error("pirate_act: invalid argument");
    return PPSCI_E_INVALID;
  }
  if (act != PPSCI_ACT_TANH && act != PPSCI_ACT_SILU && act != PPSCI_ACT_SIGMOID && act != PPSCI_ACT_COS &&
      act != PPSCI_ACT_GELU && act != PPSCI_ACT_SIN) {
    ppsci_set_error("pirate_act: activation %d has no PirateNet kernel (tanh, silu, sigmoid, sin, cos, gelu)", act);
    return PPSCI_E_UNSUPPORTED;
  }
  return PPSCI_OK;
}

extern "C" int ppsci_pirate_act_fwd(int mode, int act, int H, int64_t N, int64_t NP, int n1, int n2, const float* z,
                                    const float* bias, const float* U, const float* V, const float* x, const float* alpha,
                                    float* out, void* stream) {
  int rc = act_check(mode, act, H, N, NP, n1, n2);
  if (rc != PPSCI_OK) return rc;
  if (!z || !bias || !out || (mode == PPSCI_PIRATE_GATE && (!U || !V)) || (mode == PPSCI_PIRATE_RES && (!x || !alpha))) {
    ppsci_set_error("pirate_act_fwd: null argument");
    return PPSCI_E_INVALID;
  }
  PrActArgs a;
  memset(&a, 0, sizeof(a));
  a.z = z; a.bias = bias; a.U = U; a.V = V; a.x = x; a.alpha = alpha; a.out = out;
  a.H = H; a.n1 = n1; a.n2 = n2; a.act = act; a.mode = mode; a.N = N; a.NP = NP;
  const long long grid = (long long)H * ppsci_pirate_act_chunks(NP);
  PPSCI_LAUNCH(pirate_act_fwd_kernel, PrActArgs, (int)grid, PR_BLOCK, 0, stream, a);
  int e = PPSCI_LAST_LAUNCH_ERROR();
  if (e != 0) { ppsci_set_error("pirate_act_fwd: launch failed (%d)", e); return PPSCI_E_LAUNCH; }
  return PPSCI_OK;
}

extern "C" int ppsci_pirate_act_bwd(int mode, int act, int H, int64_t N, int64_t NP, int n1, int n2, const float* z,
                                    const float* bias, const float* U, const float* V, const float* x, const float* alpha,
                                    const float* obar, float* zbar, float* Ubar, float* Vbar, float* xbar,
                                    float* partials_b, float* partials_alpha, void* stream) {
  int rc = act_check(mode, act, H, N, NP, n1, n2);
  if (rc != PPSCI_OK) return rc;
  if (!z || !bias || !obar || !zbar || !partials_b || (mode == PPSCI_PIRATE_GATE && (!U || !V || !Ubar || !Vbar)) ||
      (mode == PPSCI_PIRATE_RES && (!x || !alpha || !xbar || !partials_alpha))) {
    ppsci_set_error("pirate_act_bwd: null argument");
    return PPSCI_E_INVALID;
  }
  PrActArgs a;
  memset(&a, 0, sizeof(a));
  a.z = z; a.bias = bias; a.U = U; a.V = V; a.x = x; a.alpha = alpha; a.obar = obar; a.zbar = zbar;
  a.Ubar = Ubar; a.Vbar = Vbar; a.xbar = xbar; a.pb = partials_b; a.palpha = partials_alpha;
  a.H = H; a.n1 = n1; a.n2 = n2; a.act = act; a.mode = mode; a.N = N; a.NP = NP;
  const long long grid = (long long)H * ppsci_pirate_act_chunks(NP);
  PPSCI_LAUNCH(pirate_act_bwd_kernel, PrActArgs, (int)grid, PR_BLOCK, 0, stream, a);
  int e = PPSCI_LAST_LAUNCH_ERROR();
  if (e != 0) { ppsci_set_error("pirate_act_bwd: launch failed (%d)", e); return PPSCI_E_LAUNCH; }
  return PPSCI_OK;
}

extern "C" int ppsci_pirate_out_fwd(int S, int m, int64_t N, int64_t NP, const float* Y, const float* bias, float* U,
                                    void* stream) {
  if (S < 1 || m < 1 || N < 1 || NP < N || !Y || !bias || !U) {
    ppsci_set_error("pirate_out_fwd: invalid argument");
    return PPSCI_E_INVALID;
  }
  PrOutArgs a{Y, bias, U, S, m, N, NP};
  const long long total = (long long)S * m * N;
  PPSCI_LAUNCH(pirate_out_fwd_kernel, PrOutArgs, (int)((total + PR_BLOCK - 1) / PR_BLOCK), PR_BLOCK, 0, stream, a);
  int e = PPSCI_LAST_LAUNCH_ERROR();
  if (e != 0) { ppsci_set_error("pirate_out_fwd: launch failed (%d)", e); return PPSCI_E_LAUNCH; }
  return PPSCI_OK;
}

extern "C" int ppsci_pirate_out_bwd(int S, int m, int64_t N, int64_t NP, const float* Ubar, float* Ybar, void* stream) {
  if (S < 1 || m < 1 || N < 1 || NP < N || !Ubar || !Ybar) {
    ppsci_set_error("pirate_out_bwd: invalid argument");
    return PPSCI_E_INVALID;
  }
  PrOutArgs a{Ubar, nullptr, Ybar, S, m, N, NP};
  const long long total = (long long)S * m * NP;
  PPSCI_LAUNCH(pirate_out_bwd_kernel, PrOutArgs, (int)((total + PR_BLOCK - 1) / PR_BLOCK), PR_BLOCK, 0, stream, a);
  int e = PPSCI_LAST_LAUNCH_ERROR();
  if (e != 0) { ppsci_set_error("pirate_out_bwd: launch failed (%d)", e); return PPSCI_E_LAUNCH; }
  return PPSCI_OK;
}
