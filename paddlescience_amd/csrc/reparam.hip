// reparam.hip -- factored / tied layer parametrisations of ppsci.arch.MLP, applied to the parameter buffers
// (never to activations): the Taylor kernels always see one plain [in, out] matrix + bias per layer.
//
//   WeightNormLinear            /root/reference/ppsci/arch/mlp.py:31-54    W = g * v / ||v||_col
//   RandomWeightFactorization   /root/reference/ppsci/arch/mlp.py:57-92    W = g * v
//   FourierEmbedding            /root/reference/ppsci/arch/mlp.py:117-136  [cos(x B), sin(x B)]: the kernels run it as
//                               a first layer with the matrix [B, B] and zero bias (cos on the first half of its
//                               features, sin on the second, taylor_tile.h), so B is duplicated here
//
// materialize: trainable tensors -> the layer's slice of the kernel parameter buffer (before the forward sweep);
// pullback   : gradient of that slice -> gradients of the trainable tensors (after the reverse sweep).
#include "ppsci_common.h"
#include <string.h>

#ifndef PPSCI_EMU
#include <hip/hip_runtime.h>
#endif

extern "C" void ppsci_set_error(const char* fmt, ...);

struct ReparamArgs {
  int kind, fin, fout;
  const float* v;   // [fin, fout]  (FOURIER: [fin, fout / 2])
  const float* g;   // [fout]       (WEIGHT_NORM / RWF)
  const float* b;   // [fout] or null
  float* W;         // [fin, fout]
  float* b_out;     // [fout] or null
  // pullback
  const float* gW;  // [fin, fout]
  const float* gb;  // [fout] or null
  float* gv;
  float* gg;
  float* gb_out;
};

// One workgroup = RP_COLS output columns x RP_RG row groups (256 threads): thread (tc, rg) owns rows rg, rg + RP_RG, ...
// of column j (neighbouring threads = neighbouring columns: 64 B runs); column sums (weight-norm's ||v||, the g
// gradients) are the row groups' partial sums added through LDS in group order (fixed order, no atomics).
#define RP_COLS 16
#define RP_RG 16

__device__ __forceinline__ float rp_colsum(float v, float* red, int tc, int rg) {
  __syncthreads();
  red[rg * RP_COLS + tc] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < RP_RG; ++k) s += red[k * RP_COLS + tc];
  return s;
}

__device__ __forceinline__ void linear_materialize_body(const ReparamArgs& a, int wg, float* red) {
  const int tc = threadIdx.x % RP_COLS, rg = threadIdx.x / RP_COLS;
  const int j = wg * RP_COLS + tc;
  const bool live = j < a.fout;
  if (a.kind == PPSCI_LINEAR_BROADCAST) {
    if (live && rg == 0) a.W[j] = a.v[0];
    return;
  }
  if (a.kind == PPSCI_LINEAR_FOURIER) {
    const int half = a.fout / 2, jj = j < half ? j : j - half;
    if (live) {
      for (int i = rg; i < a.fin; i += RP_RG) a.W[i * a.fout + j] = a.v[i * half + jj];
      if (a.b_out && rg == 0) a.b_out[j] = 0.f;
    }
    return;
  }
  if (a.kind == PPSCI_LINEAR_PLAIN) {
    if (live)
      for (int i = rg; i < a.fin; i += RP_RG) a.W[i * a.fout + j] = a.v[i * a.fout + j];
  } else if (a.kind == PPSCI_LINEAR_RWF) {
    if (live) {
      const float gj = a.g[j];
      for (int i = rg; i < a.fin; i += RP_RG) a.W[i * a.fout + j] = gj * a.v[i * a.fout + j];
    }
  } else {  // weight norm: weight_g * weight_v / norm, in that order (mlp.py:53)
    float ss = 0.f;
    if (live)
      for (int i = rg; i < a.fin; i += RP_RG) {
        const float x = a.v[i * a.fout + j];
        ss += x * x;
      }
    ss = rp_colsum(ss, red, tc, rg);
    if (live) {
      const float nrm = sqrtf(ss), gj = a.g[j];
      for (int i = rg; i < a.fin; i += RP_RG) a.W[i * a.fout + j] = gj * a.v[i * a.fout + j] / nrm;
    }
  }
  if (live && rg == 0 && a.b_out && a.b) a.b_out[j] = a.b[j];
}

__global__ void __launch_bounds__(256) linear_materialize_kernel(ReparamArgs a) {
  __shared__ float red[RP_COLS * RP_RG];
  linear_materialize_body(a, (int)blockIdx.x, red);
}

__device__ __forceinline__ void linear_pullback_body(const ReparamArgs& a, int wg, float* red) {
  const int tc = threadIdx.x % RP_COLS, rg = threadIdx.x / RP_COLS;
  const int j = wg * RP_COLS + tc;
  if (a.kind == PPSCI_LINEAR_FOURIER) {
    const int half = a.fout / 2;
    if (j >= half) return;
    for (int i = rg; i < a.fin; i += RP_RG) a.gv[i * half + j] = a.gW[i * a.fout + j] + a.gW[i * a.fout + j + half];
    return;
  }
  if (a.kind == PPSCI_LINEAR_BROADCAST) {  // one thread: a fixed-order sum of at most 256 values
    if (wg != 0 || threadIdx.x != 0) return;
    float sum = 0.f;
    for (int k = 0; k < a.fout; ++k) sum += a.gW[k];
    a.gv[0] = sum;
    return;
  }
  const bool live = j < a.fout;
  if (a.kind == PPSCI_LINEAR_PLAIN) {
    if (live)
      for (int i = rg; i < a.fin; i += RP_RG) a.gv[i * a.fout + j] = a.gW[i * a.fout + j];
  } else if (a.kind == PPSCI_LINEAR_RWF) {
    float dot = 0.f;
    if (live) {
      const float gj = a.g[j];
      for (int i = rg; i < a.fin; i += RP_RG) {
        const float gw = a.gW[i * a.fout + j];
        dot += gw * a.v[i * a.fout + j];
        a.gv[i * a.fout + j] = gw * gj;
      }
    }
    dot = rp_colsum(dot, red, tc, rg);
    if (live && rg == 0) a.gg[j] = dot;
  } else {
    float ss = 0.f, dot = 0.f;
    if (live)
      for (int i = rg; i < a.fin; i += RP_RG) {
        const float x = a.v[i * a.fout + j];
        ss += x * x;
        dot += a.gW[i * a.fout + j] * x;
      }
    ss = rp_colsum(ss, red, tc, rg);
    dot = rp_colsum(dot, red, tc, rg);
    if (live) {
      const float nrm = sqrtf(ss), gj = a.g[j];
      const float sc = gj / nrm, c = dot / ss;
      for (int i = rg; i < a.fin; i += RP_RG) a.gv[i * a.fout + j] = sc * (a.gW[i * a.fout + j] - c * a.v[i * a.fout + j]);
      if (rg == 0) a.gg[j] = dot / nrm;
    }
  }
  if (live && rg == 0 && a.gb_out && a.gb) a.gb_out[j] = a.gb[j];
}

__global__ void __launch_bounds__(256) linear_pullback_kernel(ReparamArgs a) {
  __shared__ float red[RP_COLS * RP_RG];
  linear_pullback_body(a, (int)blockIdx.x, red);
}

// Up to RP_MAX_JOBS layers in ONE launch (a PirateNet has 11 re-parametrised layers: 12 launches of 6 - 9 us each per
// direction were launch latency): workgroup b works on the job whose [first, first + count) range holds it.
#define RP_MAX_JOBS 16
struct ReparamMulti {
  ReparamArgs job[RP_MAX_JOBS];
  int first[RP_MAX_JOBS + 1];
  int n, back;
};
__global__ void __launch_bounds__(256) linear_multi_kernel(ReparamMulti m) {
  __shared__ float red[RP_COLS * RP_RG];
  int s = 0;
  while (s + 1 < m.n && (int)blockIdx.x >= m.first[s + 1]) ++s;
  const int wg = (int)blockIdx.x - m.first[s];
  if (m.back) linear_pullback_body(m.job[s], wg, red);
  else linear_materialize_body(m.job[s], wg, red);
}

static bool reparam_kind_ok(int kind) { return kind >= PPSCI_LINEAR_PLAIN && kind <= PPSCI_LINEAR_BROADCAST; }

extern "C" int ppsci_linear_materialize(int kind, int fin, int fout, const float* v, const float* g, const float* b,
                                        float* W, float* b_out, void* stream) {
  const bool needs_g = kind == PPSCI_LINEAR_WEIGHT_NORM || kind == PPSCI_LINEAR_RWF;
  if (!reparam_kind_ok(kind) || fin < 1 || fout < 1 || !v || !W || (needs_g && !g) ||
      (kind == PPSCI_LINEAR_FOURIER && (fout & 1))) {
    ppsci_set_error("linear_materialize: invalid argument");
    return PPSCI_E_INVALID;
  }
  ReparamArgs a{};
  a.kind = kind, a.fin = fin, a.fout = fout, a.v = v, a.g = g, a.b = b, a.W = W, a.b_out = b_out;
  PPSCI_LAUNCH(linear_materialize_kernel, ReparamArgs, (fout + RP_COLS - 1) / RP_COLS, 256, 0, stream, a);
  int err = PPSCI_LAST_LAUNCH_ERROR();
  if (err != 0) {
    ppsci_set_error("linear_materialize: launch failed (hip error %d)", err);
    return PPSCI_E_LAUNCH;
  }
  return PPSCI_OK;
}

extern "C" int ppsci_linear_pullback(int kind, int fin, int fout, const float* v, const float* g, const float* gW,
                                     const float* gb, float* gv, float* gg, float* gb_out, void* stream) {
  const bool needs_g = kind == PPSCI_LINEAR_WEIGHT_NORM || kind == PPSCI_LINEAR_RWF;
  if (!reparam_kind_ok(kind) || fin < 1 || fout < 1 || !gW || !gv || (needs_g && (!v || !g || !gg)) ||
      (kind == PPSCI_LINEAR_FOURIER && (fout & 1))) {
    ppsci_set_error("linear_pullback: invalid argument");
    return PPSCI_E_INVALID;
  }
  ReparamArgs a{};
  a.kind = kind, a.fin = fin, a.fout = fout, a.v = v, a.g = g, a.gW = gW, a.gb = gb, a.gv = gv, a.gg = gg,
  a.gb_out = gb_out;
  PPSCI_LAUNCH(linear_pullback_kernel, ReparamArgs, (fout + RP_COLS - 1) / RP_COLS, 256, 0, stream, a);
  int err = PPSCI_LAST_LAUNCH_ERROR();
  if (err != 0) {
    ppsci_set_error("linear_pullback: launch failed (hip error %d)", err);
    return PPSCI_E_LAUNCH;
  }
  return PPSCI_OK;
}

// n <= 16 layers per launch; `back` == 0: materialize (v, g, b -> W, b_out), 1: pullback (v, g, gW, gb -> gv, gg, gb_out)
extern "C" int ppsci_linear_multi(int n, const ppsci_linear_job* jobs, int back, void* stream) {
  if (n < 1 || n > RP_MAX_JOBS || !jobs) {
    ppsci_set_error("linear_multi: 1 .. %d jobs", RP_MAX_JOBS);
    return PPSCI_E_INVALID;
  }
  ReparamMulti m;
  memset(&m, 0, sizeof(m));
  m.n = n, m.back = back ? 1 : 0;
  int total = 0;
  for (int s = 0; s < n; ++s) {
    const ppsci_linear_job& j = jobs[s];
    const bool needs_g = j.kind == PPSCI_LINEAR_WEIGHT_NORM || j.kind == PPSCI_LINEAR_RWF;
    const bool ok = back ? (j.gW && j.gv && (!needs_g || (j.v && j.g && j.gg))) : (j.v && j.W && (!needs_g || j.g));
    if (!reparam_kind_ok(j.kind) || j.fin < 1 || j.fout < 1 || !ok || (j.kind == PPSCI_LINEAR_FOURIER && (j.fout & 1))) {
      ppsci_set_error("linear_multi: invalid job %d", s);
      return PPSCI_E_INVALID;
    }
    ReparamArgs a{};
    a.kind = j.kind, a.fin = j.fin, a.fout = j.fout, a.v = j.v, a.g = j.g, a.b = j.b, a.W = j.W, a.b_out = j.b_out;
    a.gW = j.gW, a.gb = j.gb, a.gv = j.gv, a.gg = j.gg, a.gb_out = j.gb_out;
    m.job[s] = a;
    m.first[s] = total;
    total += (j.fout + RP_COLS - 1) / RP_COLS;
  }
  m.first[n] = total;
  PPSCI_LAUNCH(linear_multi_kernel, ReparamMulti, total, 256, 0, stream, m);
  if (PPSCI_LAST_LAUNCH_ERROR() != 0) {
    ppsci_set_error("linear_multi: launch failed");
    return PPSCI_E_LAUNCH;
  }
  return PPSCI_OK;
}

// ---- per-layer widths (mlp.py:199-201: hidden_size as a tuple): the Taylor kernels run ONE padded width H = max(widths);
// a layer's [fin_s, fout_s] matrix is the top-left block of its [fin_d, fout_d] kernel-layout matrix, everything else is zero.
// Padded features evaluate act(0) but feed only zero weights; their gradient entries are never pulled back, so they stay zero.
struct PadArgs {
  int fin_s, fout_s, fin_d, fout_d, back;
  const float* src;   // forward: trainable W [fin_s, fout_s];  back: kernel-layout gradient [fin_d, fout_d]
  const float* srcb;  // bias [fout_s] / [fout_d] or null
  float* dst;
  float* dstb;
};

__global__ void __launch_bounds__(256) linear_pad_kernel(PadArgs a) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (!a.back) {
    if (j >= a.fout_d) return;
    for (int i = 0; i < a.fin_d; ++i) a.dst[i * a.fout_d + j] = (i < a.fin_s && j < a.fout_s) ? a.src[i * a.fout_s + j] : 0.f;
    if (a.dstb) a.dstb[j] = (a.srcb && j < a.fout_s) ? a.srcb[j] : 0.f;
  } else {
    if (j >= a.fout_s) return;
    for (int i = 0; i < a.fin_s; ++i) a.dst[i * a.fout_s + j] = a.src[i * a.fout_d + j];
    if (a.dstb && a.srcb) a.dstb[j] = a.srcb[j];
  }
}

static int launch_pad(PadArgs& a, void* stream) {
  if (a.fin_s < 1 || a.fout_s < 1 || a.fin_d < a.fin_s || a.fout_d < a.fout_s || !a.src || !a.dst) {
    ppsci_set_error("linear_pad: invalid argument");
    return PPSCI_E_INVALID;
  }
  PPSCI_LAUNCH(linear_pad_kernel, PadArgs, ((a.back ? a.fout_s : a.fout_d) + 255) / 256, 256, 0, stream, a);
  if (PPSCI_LAST_LAUNCH_ERROR() != 0) {
    ppsci_set_error("linear_pad: launch failed");
    return PPSCI_E_LAUNCH;
  }
  return PPSCI_OK;
}

extern "C" int ppsci_linear_pad(int fin_src, int fout_src, int fin_dst, int fout_dst, const float* v, const float* b, float* W,
                                float* b_out, void* stream) {
  PadArgs a{fin_src, fout_src, fin_dst, fout_dst, 0, v, b, W, b_out};
  return launch_pad(a, stream);
}

extern "C" int ppsci_linear_unpad(int fin_src, int fout_src, int fin_dst, int fout_dst, const float* gW, const float* gb,
                                  float* gv, float* gb_out, void* stream) {
  PadArgs a{fin_src, fout_src, fin_dst, fout_dst, 1, gW, gb, gv, gb_out};
  return launch_pad(a, stream);
}
