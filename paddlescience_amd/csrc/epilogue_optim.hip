// epilogue_optim.hip -- pointwise epilogue VM (+ fused MSE and its adjoint), row reduction, Adam.
//
// Epilogue: the VM itself is in epilogue_vm.h (shared with the one-launch step kernel); here its stand-alone launch.
// Adam: paddle.optimizer.Adam as configured by /root/reference/ppsci/optimizer/optimizer.py:225-248.
#include "ppsci_common.h"

#ifndef PPSCI_EMU
#include <hip/hip_runtime.h>
#endif
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "epilogue_vm.h"

template <int MODE>
__global__ void __launch_bounds__(EPI_BLOCK) epilogue_kernel(EpiArgs a) {
  PPSCI_DYN_SMEM(red);  // [EPI_BLOCK] (+ EPI_RF_LDS: [2][n][EPI_BLOCK])
  epilogue_body<MODE>(a, red);
}

// ---------------------------------------------------------------------------------- reduce / Adam
struct ReduceArgs {
  const float* partials;
  float* out;
  long long rows, cols;
  int accumulate;
  int groups;  // row groups per workgroup (8 or 32)
};

// 256 threads = `cols_wg` columns x `groups` row groups (32 x 8, or 8 x 32 for tall narrow inputs: many partial rows of a
// small weight matrix); each thread sums rows r = rg, rg + groups, ... of its column in a fixed order, the group sums are
// combined in a fixed order through LDS (deterministic).
#define RED_COLS 32
#define RED_GROUPS 8
__global__ void __launch_bounds__(256) reduce_rows_kernel(ReduceArgs a) {
  PPSCI_DYN_SMEM(red);  // [groups][cols_wg] = 256 floats
  const int groups = a.groups, cw = 256 / groups;
  const int tc = threadIdx.x % cw, rg = threadIdx.x / cw;
  const long long j = (long long)blockIdx.x * cw + tc;
  float s = 0.f;
  if (j < a.cols) {
#pragma unroll 8
    for (long long r = rg; r < a.rows; r += groups) s += a.partials[r * a.cols + j];
  }
  red[rg * cw + tc] = s;
  __syncthreads();
  if (rg == 0 && j < a.cols) {
    float t = red[tc];
    for (int k = 1; k < groups; ++k) t += red[k * cw + tc];
    a.out[j] = a.accumulate ? a.out[j] + t : t;
  }
}

// Several independent row reductions in ONE launch (the weight-gradient partials of the layers of an FNO backward: eight
// reductions of a few microseconds each, which are launch latency, not work): workgroup b belongs to the segment whose
// [first, first + count) range holds it and does there exactly what reduce_rows_kernel does.
#define RED_MAX_SEG 16
struct ReduceMultiArgs {
  ReduceArgs seg[RED_MAX_SEG];
  int first[RED_MAX_SEG + 1];  // workgroup ranges of the segments
  int nseg;
};
__global__ void __launch_bounds__(256) reduce_rows_multi_kernel(ReduceMultiArgs m) {
  PPSCI_DYN_SMEM(red);
  int s = 0;
  while (s + 1 < m.nseg && (int)blockIdx.x >= m.first[s + 1]) ++s;
  const ReduceArgs& a = m.seg[s];
  const int wg = (int)blockIdx.x - m.first[s];
  const int groups = a.groups, cw = 256 / groups;
  const int tc = threadIdx.x % cw, rg = threadIdx.x / cw;
  const long long j = (long long)wg * cw + tc;
  float v = 0.f;
  if (j < a.cols) {
#pragma unroll 8
    for (long long r = rg; r < a.rows; r += groups) v += a.partials[r * a.cols + j];
  }
  red[rg * cw + tc] = v;
  __syncthreads();
  if (rg == 0 && j < a.cols) {
    float t = red[tc];
    for (int k = 1; k < groups; ++k) t += red[k * cw + tc];
    a.out[j] = a.accumulate ? a.out[j] + t : t;
  }
}

// One row: a plain copy / accumulate of `cols` floats (gradient hand-over between stages), every thread busy, float4
// where the pointers allow it.
__global__ void __launch_bounds__(256) reduce_rows_one_kernel(ReduceArgs a) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (a.groups) {  // float4 path
    const long long n4 = a.cols >> 2;
    if (t < n4) {
      f32x4 v = ((const f32x4*)a.partials)[t];
      if (a.accumulate) v += ((const f32x4*)a.out)[t];
      ((f32x4*)a.out)[t] = v;
    } else if (t == n4) {
      for (long long j = 4 * n4; j < a.cols; ++j) a.out[j] = a.accumulate ? a.out[j] + a.partials[j] : a.partials[j];
    }
  } else if (t < a.cols) {
    a.out[t] = a.accumulate ? a.out[t] + a.partials[t] : a.partials[t];
  }
}

// Few columns, many rows (loss partials: one row per wave): one workgroup per column, thread t sums rows t, t+256, ...
// in order, then a fixed-shape LDS tree (deterministic).
__global__ void __launch_bounds__(256) reduce_rows_narrow_kernel(ReduceArgs a) {
  PPSCI_DYN_SMEM(red);  // [256]
  const long long j = blockIdx.x;
  float s = 0.f;
#pragma unroll 8
  for (long long r = threadIdx.x; r < a.rows; r += 256) s += a.partials[r * a.cols + j];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) a.out[j] = a.accumulate ? a.out[j] + red[0] : red[0];
}

struct AdamArgs {
  float* p;
  const float* g;
  float* m;
  float* v;
  long long n;
  float lr_t, beta1, beta2, eps_t, grad_scale;
};

// The update of ONE parameter from its (raw) gradient.  Every multiply-add is written as an explicit fmaf in a fixed form: the
// compiler is free to contract  beta1 * m + (1 - beta1) * g  either way, and it chose differently in adam_kernel and in
// reduce_rows_multi_adam_kernel (seen on MI355X: the two paths differed by one ulp per step).  One inline function, one
// rounding sequence, wherever this file applies Adam.
__device__ __forceinline__ void ppsci_adam_one(const AdamArgs& a, long long j, float graw) {
  const float g = a.grad_scale * graw;
  const float m = __builtin_fmaf(a.beta1, a.m[j], (1.f - a.beta1) * g);
  const float v = __builtin_fmaf(a.beta2, a.v[j], ((1.f - a.beta2) * g) * g);
  a.m[j] = m;
  a.v[j] = v;
  a.p[j] = __builtin_fmaf(-a.lr_t, m / (sqrtf(v) + a.eps_t), a.p[j]);
}

__global__ void __launch_bounds__(256) adam_kernel(AdamArgs a) {
  const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
  if (j >= a.n) return;
  ppsci_adam_one(a, j, a.g[j]);
}

// SGD / Momentum / RMSProp / AdamW on the flat parameter buffer (paddle.optimizer semantics as wrapped by
// /root/reference/ppsci/optimizer/optimizer.py:39-176, :326-383, :386-495).  hy: lr, grad_scale, l2 (L2Decay
// coefficient added to the gradient), a, b, c, flag:
//   SGD       p -= lr g
//   MOMENTUM  v = a v + g;  p -= lr (flag ? g + a v : v)                          a = momentum, flag = nesterov
//   RMSPROP   r = a r + (1-a) g^2;  [flag: mg = a mg + (1-a) g]  v = c v + lr g / sqrt(r - mg^2 + b);  p -= v
//                                                                    a = rho, b = epsilon, c = momentum, flag = centered
//   ADAMW     p *= (1 - lr_base * coeff) [in hy.c];  then Adam with lr_t (hy.lr) and eps_t (hy.b), a = beta1, flag: beta2 in hy.d
struct OptimArgs {
  float* p;
  const float* g;
  float* s1;
  float* s2;
  float* s3;
  long long n;
  int kind, flag;
  float lr, grad_scale, l2, a, b, c, d;
};

__global__ void __launch_bounds__(256) optim_kernel(OptimArgs o) {
  const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
  if (j >= o.n) return;
  float p = o.p[j];
  float g = o.grad_scale * o.g[j] + o.l2 * p;
  if (o.kind == PPSCI_OPT_SGD) {
    p -= o.lr * g;
  } else if (o.kind == PPSCI_OPT_MOMENTUM) {
    const float v = o.a * o.s1[j] + g;
    o.s1[j] = v;
    p -= o.lr * (o.flag ? g + o.a * v : v);
  } else if (o.kind == PPSCI_OPT_RMSPROP) {
    const float r = o.a * o.s1[j] + (1.f - o.a) * g * g;
    o.s1[j] = r;
    float mg = 0.f;
    if (o.flag) {
      mg = o.a * o.s3[j] + (1.f - o.a) * g;
      o.s3[j] = mg;
    }
    const float v = o.c * o.s2[j] + o.lr * g / sqrtf(r - mg * mg + o.b);
    o.s2[j] = v;
    p -= v;
  } else {  // PPSCI_OPT_ADAMW
    p *= o.c;
    const float m = o.a * o.s1[j] + (1.f - o.a) * g;
    const float v = o.d * o.s2[j] + (1.f - o.d) * g * g;
    o.s1[j] = m;
    o.s2[j] = v;
    p -= o.lr * (m / (sqrtf(v) + o.b));
  }
  o.p[j] = p;
}

// ------------------------------------------------------------------------------------ host side
static thread_local char g_err[512] = "";

extern "C" void ppsci_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* ppsci_last_error(void) { return g_err; }

static int g_max_grid = 0;
extern "C" void ppsci_set_max_grid(int max_blocks) { g_max_grid = max_blocks > 0 ? max_blocks : 0; }
extern "C" int ppsci_get_max_grid(void) { return g_max_grid; }
static int g_wide_min_nb = 8;
extern "C" void ppsci_set_wide_min_nb(int nb) { g_wide_min_nb = nb; }
extern "C" int ppsci_get_wide_min_nb(void) { return g_wide_min_nb; }
static int g_bwd_lw = 1;
extern "C" void ppsci_set_bwd_layerwise(int on) { g_bwd_lw = on ? 1 : 0; }
extern "C" int ppsci_get_bwd_layerwise(void) { return g_bwd_lw; }
static int g_bwd_accum = 1;
extern "C" void ppsci_set_bwd_accum(int on) { g_bwd_accum = on; }
extern "C" int ppsci_get_bwd_accum(void) { return g_bwd_accum; }

extern "C" int ppsci_is_device_build(void) {
#ifdef PPSCI_EMU
  return 0;
#else
  return 1;
#endif
}

// The cross-workgroup "last one out" reductions (taylor_step_tail.h, epilogue_vm.h, taylor_fused.inc) publish their rows
// with agent-scope stores + s_waitcnt vmcnt(0) + a relaxed ticket: that is gfx950 behaviour (write-through stores
// counted by vmcnt, no threadgroup-split mode), not the HIP memory model.  The code object only holds gfx950 ISA, so
// another device could not run it anyway; this check turns that into a readable error before the first launch.
extern "C" int ppsci_check_device(void) {
#ifdef PPSCI_EMU
  return PPSCI_OK;
#else
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    ppsci_set_error("no HIP device");
    return PPSCI_E_LAUNCH;
  }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    ppsci_set_error("device %d is %s: this library is written for gfx950 (MI355X) only", dev, prop.gcnArchName);
    return PPSCI_E_UNSUPPORTED;
  }
  return PPSCI_OK;
#endif
}

extern "C" int64_t ppsci_param_count(const ppsci_mlp_desc* d) {
  ppsci_derived q;
  if (!d || ppsci_derive(d, &q) != PPSCI_OK) return -1;
  return q.P;
}

extern "C" int64_t ppsci_stash_bytes(const ppsci_mlp_desc* d, int64_t n_points) {
  ppsci_derived q;
  if (!d || n_points <= 0 || ppsci_derive(d, &q) != PPSCI_OK) return 0;
  const int64_t ntiles = (n_points + PPSCI_TILE - 1) / PPSCI_TILE;
  const int64_t S = 1 + d->n1 + d->n2 + d->n3 + d->n4;
  return ntiles * d->n_hidden * S * q.NB * 64 * 16;
}

static int epi_grid(int64_t n_points, int* iters) {
  int64_t blocks = (n_points + EPI_BLOCK - 1) / EPI_BLOCK;
  int64_t grid = blocks < 2048 ? blocks : 2048;
  if (grid < 1) grid = 1;
  *iters = (int)((blocks + grid - 1) / grid);
  return (int)grid;
}

extern "C" int64_t ppsci_epilogue_partial_rows(int64_t n_points) {
  int iters;
  return n_points > 0 ? epi_grid(n_points, &iters) : 0;
}

extern "C" int ppsci_epilogue(const ppsci_epilogue_desc* e, int64_t n_points, const float* const* inputs_host,
                              const float* U, const float* const* aux_host, float* residual_out, float* Ubar,
                              float* loss_partials, void* stream) {
  return ppsci_epilogue_params(e, n_points, inputs_host, U, aux_host, residual_out, Ubar, loss_partials, nullptr,
                               nullptr, stream);
}

extern "C" int ppsci_epilogue_params(const ppsci_epilogue_desc* e, int64_t n_points, const float* const* inputs_host,
                                     const float* U, const float* const* aux_host, float* residual_out, float* Ubar,
                                     float* loss_partials, const float* eq_params, float* eq_param_partials,
                                     void* stream) {
  return ppsci_epilogue_losses(e, n_points, inputs_host, U, aux_host, residual_out, Ubar, loss_partials, eq_params,
                               eq_param_partials, nullptr, nullptr, stream);
}

extern "C" int ppsci_epilogue_losses(const ppsci_epilogue_desc* e, int64_t n_points, const float* const* inputs_host,
                                     const float* U, const float* const* aux_host, float* residual_out, float* Ubar,
                                     float* loss_partials, const float* eq_params, float* eq_param_partials,
                                     float* loss_terms, void* counter, void* stream) {
  if ((loss_terms != nullptr) != (counter != nullptr)) {
    ppsci_set_error("epilogue: loss_terms and counter come together");
    return PPSCI_E_INVALID;
  }
  if (!e || n_points <= 0 || !loss_partials || e->n_instr < 1 || e->n_instr > PPSCI_MAX_PROG || e->n_res < 0 ||
      e->n_res > PPSCI_MAX_RES || e->n_in < 0 || e->n_in > PPSCI_MAX_IN || e->n_aux < 0 || e->n_aux > PPSCI_MAX_AUX) {
    ppsci_set_error("epilogue: invalid argument");
    return PPSCI_E_INVALID;
  }
  bool uses_params = false;
  for (int i = 0; i < e->n_instr; ++i) {
    const ppsci_instr& ins = e->prog[i];
    bool ok = ins.op >= 0 && ins.op < PPSCI_OP_COUNT;
    if (ok) {
      if (ins.op == PPSCI_OP_LD_IN) ok = ins.a >= 0 && ins.a < e->n_in && inputs_host;
      else if (ins.op == PPSCI_OP_LD_U) ok = ins.a >= 0 && ins.a < e->n_streams && U;
      else if (ins.op == PPSCI_OP_LD_AUX) ok = ins.a >= 0 && ins.a < e->n_aux && aux_host;
      else if (ins.op == PPSCI_OP_LD_PARAM) {
        ok = ins.a >= 0 && ins.a < PPSCI_MAX_EPARAM && eq_params && (Ubar == nullptr || eq_param_partials);
        uses_params = true;
      }
      else if (ins.op != PPSCI_OP_CONST) {
        ok = ins.a >= 0 && ins.a < i;
        const bool binary = ins.op == PPSCI_OP_ADD || ins.op == PPSCI_OP_SUB || ins.op == PPSCI_OP_MUL ||
                            ins.op == PPSCI_OP_DIV || ins.op == PPSCI_OP_POW || ins.op == PPSCI_OP_MAX ||
                            ins.op == PPSCI_OP_MIN || ins.op == PPSCI_OP_ATAN2;
        if (binary) ok = ok && ins.b >= 0 && ins.b < i;
      }
    }
    if (!ok) {
      ppsci_set_error("epilogue: bad instruction %d (op %d a %d b %d)", i, ins.op, ins.a, ins.b);
      return PPSCI_E_INVALID;
    }
  }
  for (int k = 0; k < e->n_res; ++k) {
    const ppsci_residual& r = e->res[k];
    if (r.value < 0 || r.value >= e->n_instr || r.label >= e->n_aux || r.weight >= e->n_aux || r.area >= e->n_aux ||
        r.kind < PPSCI_LOSS_MSE || r.kind > PPSCI_LOSS_LINEAR || r.scale_param < 0 || r.scale_param > PPSCI_MAX_EPARAM ||
        (r.scale_param > 0 && !eq_params)) {
      ppsci_set_error("epilogue: bad residual %d", k);
      return PPSCI_E_INVALID;
    }
  }
  EpiArgs a;
  memset(&a, 0, sizeof(a));
  a.e = *e;
  for (int j = 0; j < e->n_in; ++j) a.x[j] = inputs_host[j];
  for (int j = 0; j < e->n_aux; ++j) a.aux[j] = aux_host[j];
  a.U = U;
  a.resid = residual_out;
  a.Ubar = Ubar;
  a.partials = loss_partials;
  a.ep = eq_params;
  a.ep_part = (uses_params && Ubar != nullptr) ? eq_param_partials : nullptr;
  a.N = n_points;
  a.loss_out = loss_terms;
  a.counter = (unsigned*)counter;
  epi_fill_loads(a);
  const int grid = epi_grid(n_points, &a.iters);
  if (a.e.n_instr <= EPI_LDS_PROG) {
    const int lds = (EPI_BLOCK + 2 * a.e.n_instr * EPI_BLOCK) * (int)sizeof(float);
    PPSCI_LAUNCH(epilogue_kernel<EPI_RF_LDS>, EpiArgs, grid, EPI_BLOCK, lds, stream, a);
  } else {
    PPSCI_LAUNCH(epilogue_kernel<EPI_RF_SCRATCH>, EpiArgs, grid, EPI_BLOCK, EPI_BLOCK * sizeof(float), stream, a);
  }
  int err = PPSCI_LAST_LAUNCH_ERROR();
  if (err != 0) {
    ppsci_set_error("epilogue: launch failed (hip error %d)", err);
    return PPSCI_E_LAUNCH;
  }
  return PPSCI_OK;
}

// ---- CausalMSELoss (mse.py:158-177): window means, then the exclusive running sum as a per-point factor
struct CausalArgs {
  const float *value, *label, *weight, *area;
  float *chunk, *cw;
  long long N;
  int n_chunks, per;  // per = N / n_chunks
  float tol;
};

__global__ void __launch_bounds__(256) causal_chunk_mean_kernel(CausalArgs a) {
  PPSCI_DYN_SMEM(red);  // [256]
  const int k = blockIdx.x, tid = threadIdx.x;
  float s = 0.f;
  for (int i = tid; i < a.per; i += 256) {
    const long long p = (long long)k * a.per + i;
    const float d = a.value[p] - (a.label ? a.label[p] : 0.f);
    float l = d * d;
    if (a.weight) l *= a.weight[p];
    if (a.area) l *= a.area[p];
    s += l;
  }
  red[tid] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st) red[tid] += red[tid + st];
    __syncthreads();
  }
  if (tid == 0) a.chunk[k] = red[0] / (float)a.per;
}

__global__ void __launch_bounds__(256) causal_weight_kernel(CausalArgs a) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= a.N) return;
  const int k = (int)(p / a.per);
  float acc = 0.f;
  for (int j = 0; j < k; ++j) acc += a.chunk[j];  // acc_mat @ means: strictly lower-triangular ones (mse.py:153-155)
  const float w = expf(-a.tol * acc);
  a.cw[p] = a.area ? w * a.area[p] : w;
}

extern "C" int ppsci_causal_weights(int64_t n_points, int n_chunks, float tol, const float* value,
                                    const float* label, const float* weight, const float* area,
                                    float* chunk_scratch, float* cw, void* stream) {
  if (n_points <= 0 || n_chunks <= 0 || n_points % n_chunks != 0 || !value || !chunk_scratch || !cw) {
    ppsci_set_error("causal_weights: invalid argument (N must be a multiple of n_chunks)");
    return PPSCI_E_INVALID;
  }
  CausalArgs a{value, label, weight, area, chunk_scratch, cw, (long long)n_points, n_chunks,
               (int)(n_points / n_chunks), tol};
  PPSCI_LAUNCH(causal_chunk_mean_kernel, CausalArgs, n_chunks, 256, 256 * sizeof(float), stream, a);
  PPSCI_LAUNCH(causal_weight_kernel, CausalArgs, (int)((n_points + 255) / 256), 256, 0, stream, a);
  int err = PPSCI_LAST_LAUNCH_ERROR();
  if (err != 0) {
    ppsci_set_error("causal_weights: launch failed (hip error %d)", err);
    return PPSCI_E_LAUNCH;
  }
  return PPSCI_OK;
}

extern "C" int ppsci_reduce_rows(const float* partials, int64_t rows, int64_t cols, float* out, int accumulate,
                                 void* stream) {
  if (!partials || !out || rows < 0 || cols <= 0) {
    ppsci_set_error("reduce_rows: invalid argument");
    return PPSCI_E_INVALID;
  }
  ReduceArgs a{partials, out, rows, cols, accumulate, RED_GROUPS};
  if (rows == 1) {
    a.groups = (((uintptr_t)partials | (uintptr_t)out) & 15) == 0 ? 1 : 0;  // here: 1 = float4 path
    const long long nthr = a.groups ? (cols >> 2) + 1 : cols;
    PPSCI_LAUNCH(reduce_rows_one_kernel, ReduceArgs, (int)((nthr + 255) / 256), 256, 0, stream, a);
  } else if (cols <= 8 && rows >= 512) {
    PPSCI_LAUNCH(reduce_rows_narrow_kernel, ReduceArgs, (int)cols, 256, 256 * sizeof(float), stream, a);
  } else {
    // tall inputs: more row parallelism and more workgroups; narrower column runs only while the matrix is small
    if (rows >= 128 && cols <= 2048) a.groups = 32;
    else if (rows >= 128 && cols <= 32768) a.groups = 16;
    const int cw = 256 / a.groups;
    const int grid = (int)((cols + cw - 1) / cw);
    PPSCI_LAUNCH(reduce_rows_kernel, ReduceArgs, grid, 256, 256 * sizeof(float), stream, a);
  }
  int err = PPSCI_LAST_LAUNCH_ERROR();
  if (err != 0) {
    ppsci_set_error("reduce_rows: launch failed (hip error %d)", err);
    return PPSCI_E_LAUNCH;
  }
  return PPSCI_OK;
}

extern "C" int ppsci_reduce_rows_multi(int nseg, const ppsci_reduce_seg* segs, void* stream) {
  if (nseg < 1 || nseg > RED_MAX_SEG || !segs) {
    ppsci_set_error("reduce_rows_multi: 1 .. %d segments", RED_MAX_SEG);
    return PPSCI_E_INVALID;
  }
  ReduceMultiArgs m;
  memset(&m, 0, sizeof(m));
  m.nseg = nseg;
  int total = 0;
  for (int s = 0; s < nseg; ++s) {
    const ppsci_reduce_seg& g = segs[s];
    if (!g.partials || !g.out || g.rows < 1 || g.cols < 1) {
      ppsci_set_error("reduce_rows_multi: invalid segment %d", s);
      return PPSCI_E_INVALID;
    }
    ReduceArgs a{g.partials, g.out, g.rows, g.cols, g.accumulate, RED_GROUPS};
    if (g.cols <= 8 && g.rows >= 512) a.groups = 256;  // few columns, many rows: a whole workgroup per column
    else if (g.rows >= 128 && g.cols <= 2048) a.groups = 32;  // (as ppsci_reduce_rows)
    else if (g.rows >= 128 && g.cols <= 32768) a.groups = 16;
    const int cw = 256 / a.groups;
    m.seg[s] = a;
    m.first[s] = total;
    total += (int)((g.cols + cw - 1) / cw);
  }
  m.first[nseg] = total;
  PPSCI_LAUNCH(reduce_rows_multi_kernel, ReduceMultiArgs, total, 256, 256 * sizeof(float), stream, m);
  if (PPSCI_LAST_LAUNCH_ERROR() != 0) {
    ppsci_set_error("reduce_rows_multi: launch failed");
    return PPSCI_E_LAUNCH;
  }
  return PPSCI_OK;
}

extern "C" int ppsci_adam_step(int64_t n, float* params, const float* grad, float* m, float* v, float lr,
                               float beta1, float beta2, float eps, int64_t step_t, float grad_scale,
                               void* stream) {
  if (!params || !grad || !m || !v || n <= 0 || step_t < 1) {
    ppsci_set_error("adam_step: invalid argument");
    return PPSCI_E_INVALID;
  }
  const double b1t = pow((double)beta1, (double)step_t), b2t = pow((double)beta2, (double)step_t);
  const double c2 = sqrt(1.0 - b2t);
  AdamArgs a{params, grad, m, v, n, (float)(lr * c2 / (1.0 - b1t)), beta1, beta2, (float)(eps * c2), grad_scale};
  const int grid = (int)((n + 255) / 256);
  PPSCI_LAUNCH(adam_kernel, AdamArgs, grid, 256, 0, stream, a);
  int err = PPSCI_LAST_LAUNCH_ERROR();
  if (err != 0) {
    ppsci_set_error("adam_step: launch failed (hip error %d)", err);
    return PPSCI_E_LAUNCH;
  }
  return PPSCI_OK;
}

// ---- the row reductions that END a backward pass and the Adam update behind them, in ONE launch -------------------------
// (SPINN: the per-tile gradient rows of the three branch nets + the loss rows; FNO: the per-chunk partials of every 1x1
// convolution's weight gradient -- each was a ppsci_reduce_rows_multi launch followed by ppsci_adam_step, two launch-floor
// kernels.)  A workgroup of the reduction part does exactly what reduce_rows_multi_kernel does; the thread that finishes a
// column whose destination lies inside the gradient buffer applies the update of that parameter at once.  Parameters whose
// gradient no segment writes (other kernels wrote it: spectral weights, norm parameters) are updated by the plain-Adam
// workgroups behind the reduction part.  Every parameter must get its gradient from at most ONE segment.
#define ADAM_MAX_PLAIN (RED_MAX_SEG + 1)
struct ReduceAdamArgs {
  ReduceMultiArgs r;
  AdamArgs a;
  long long goff[RED_MAX_SEG];               // segment s writes grad[goff[s] .. goff[s] + cols); -1: not a gradient (loss rows)
  long long plain0[ADAM_MAX_PLAIN], plain1[ADAM_MAX_PLAIN];  // parameter ranges no segment covers
  int pfirst[ADAM_MAX_PLAIN + 1];             // their workgroup ranges (relative to the first plain workgroup)
  int nplain, wg_plain;                       // wg_plain: index of the first plain workgroup
};

__global__ void __launch_bounds__(256) reduce_rows_multi_adam_kernel(ReduceAdamArgs q) {
  PPSCI_DYN_SMEM(red);
  if ((int)blockIdx.x >= q.wg_plain) {  // plain Adam on an uncovered parameter range
    const int wgp = (int)blockIdx.x - q.wg_plain;
    int k = 0;
    while (k + 1 < q.nplain && wgp >= q.pfirst[k + 1]) ++k;
    const long long j = q.plain0[k] + (long long)(wgp - q.pfirst[k]) * 256 + threadIdx.x;
    if (j < q.plain1[k]) ppsci_adam_one(q.a, j, q.a.g[j]);
    return;
  }
  const ReduceMultiArgs& m = q.r;
  int s = 0;
  while (s + 1 < m.nseg && (int)blockIdx.x >= m.first[s + 1]) ++s;
  const ReduceArgs& a = m.seg[s];
  const int wg = (int)blockIdx.x - m.first[s];
  const int groups = a.groups, cw = 256 / groups;
  const int tc = threadIdx.x % cw, rg = threadIdx.x / cw;
  const long long j = (long long)wg * cw + tc;
  float v = 0.f;
  if (j < a.cols) {
#pragma unroll 8
    for (long long r = rg; r < a.rows; r += groups) v += a.partials[r * a.cols + j];
  }
  red[rg * cw + tc] = v;
  __syncthreads();
  if (rg == 0 && j < a.cols) {
    float t = red[tc];
    for (int k = 1; k < groups; ++k) t += red[k * cw + tc];
    if (a.accumulate) t += a.out[j];
    a.out[j] = t;
    if (q.goff[s] >= 0) ppsci_adam_one(q.a, q.goff[s] + j, t);
  }
}

extern "C" int ppsci_reduce_rows_multi_adam(int nseg, const ppsci_reduce_seg* segs, int64_t n, float* params, float* grad,
                                            float* mom, float* var, float lr, float beta1, float beta2, float eps,
                                            int64_t step_t, float grad_scale, void* stream) {
  if (nseg < 1 || nseg > RED_MAX_SEG || !segs || !params || !grad || !mom || !var || n <= 0 || step_t < 1) {
    ppsci_set_error("reduce_rows_multi_adam: invalid argument");
    return PPSCI_E_INVALID;
  }
  ReduceAdamArgs q;
  memset(&q, 0, sizeof(q));
  q.r.nseg = nseg;
  int total = 0;
  // gradient ranges of the segments (sorted below to find what they leave uncovered)
  long long lo[RED_MAX_SEG], hi[RED_MAX_SEG];
  int ncov = 0;
  for (int s = 0; s < nseg; ++s) {
    const ppsci_reduce_seg& g = segs[s];
    if (!g.partials || !g.out || g.rows < 1 || g.cols < 1) {
      ppsci_set_error("reduce_rows_multi_adam: invalid segment %d", s);
      return PPSCI_E_INVALID;
    }
    ReduceArgs a{(const float*)g.partials, (float*)g.out, g.rows, g.cols, g.accumulate, RED_GROUPS};
    if (g.cols <= 8 && g.rows >= 512) a.groups = 256;
    else if (g.rows >= 128 && g.cols <= 2048) a.groups = 32;
    else if (g.rows >= 128 && g.cols <= 32768) a.groups = 16;
    const int cw = 256 / a.groups;
    q.r.seg[s] = a;
    q.r.first[s] = total;
    total += (int)((g.cols + cw - 1) / cw);
    const float* o = (const float*)g.out;
    if (o >= grad && o < grad + n) {
      const long long off = o - grad;
      if (off + g.cols > n || g.accumulate) {
        ppsci_set_error("reduce_rows_multi_adam: gradient segment %d leaves the buffer or accumulates", s);
        return PPSCI_E_INVALID;
      }
      q.goff[s] = off;
      lo[ncov] = off;
      hi[ncov] = off + g.cols;
      ++ncov;
    } else {
      q.goff[s] = -1;
    }
  }
  q.r.first[nseg] = total;
  for (int i = 1; i < ncov; ++i)  // insertion sort by start
    for (int k = i; k > 0 && lo[k] < lo[k - 1]; --k) {
      const long long tl = lo[k], th = hi[k];
      lo[k] = lo[k - 1]; hi[k] = hi[k - 1];
      lo[k - 1] = tl; hi[k - 1] = th;
    }
  long long cur = 0;
  int np = 0, wgp = 0;
  for (int i = 0; i <= ncov; ++i) {
    const long long end = i < ncov ? lo[i] : n;
    if (i < ncov && lo[i] < cur) {
      ppsci_set_error("reduce_rows_multi_adam: two segments write the same parameters");
      return PPSCI_E_INVALID;
    }
    if (end > cur) {
      q.plain0[np] = cur;
      q.plain1[np] = end;
      q.pfirst[np] = wgp;
      wgp += (int)((end - cur + 255) / 256);
      ++np;
    }
    if (i < ncov) cur = hi[i];
  }
  q.pfirst[np] = wgp;
  q.nplain = np;
  q.wg_plain = total;
  const double b1t = pow((double)beta1, (double)step_t), b2t = pow((double)beta2, (double)step_t);
  const double c2 = sqrt(1.0 - b2t);
  q.a = AdamArgs{params, grad, mom, var, n, (float)(lr * c2 / (1.0 - b1t)), beta1, beta2, (float)(eps * c2), grad_scale};
  PPSCI_LAUNCH(reduce_rows_multi_adam_kernel, ReduceAdamArgs, total + wgp, 256, 256 * sizeof(float), stream, q);
  if (PPSCI_LAST_LAUNCH_ERROR() != 0) {
    ppsci_set_error("reduce_rows_multi_adam: launch failed");
    return PPSCI_E_LAUNCH;
  }
  return PPSCI_OK;
}

extern "C" int ppsci_optim_step(int kind, int64_t n, float* params, const float* grad, float* state1, float* state2,
                                float* state3, const float* hyper, int flag, void* stream) {
  if (!params || !grad || !hyper || n <= 0 || kind < PPSCI_OPT_SGD || kind > PPSCI_OPT_ADAMW ||
      (kind != PPSCI_OPT_SGD && !state1) || ((kind == PPSCI_OPT_RMSPROP || kind == PPSCI_OPT_ADAMW) && !state2) ||
      (kind == PPSCI_OPT_RMSPROP && flag && !state3)) {
    ppsci_set_error("optim_step: invalid argument");
    return PPSCI_E_INVALID;
  }
  OptimArgs o{params, grad, state1, state2, state3, n, kind, flag,
              hyper[0], hyper[1], hyper[2], hyper[3], hyper[4], hyper[5], hyper[6]};
  PPSCI_LAUNCH(optim_kernel, OptimArgs, (int)((n + 255) / 256), 256, 0, stream, o);
  int err = PPSCI_LAST_LAUNCH_ERROR();
  if (err != 0) {
    ppsci_set_error("optim_step: launch failed (hip error %d)", err);
    return PPSCI_E_LAUNCH;
  }
  return PPSCI_OK;
}
