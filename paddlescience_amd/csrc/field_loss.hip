// field_loss.hip -- the losses of the operator-learning path on [rows][H][W] fields (rows = batch x channels), value AND
// adjoint, as three small kernels: per-row sums -> per-row terms + total -> adjoint field.
//
// Replaces, for the FNO training / evaluation step,
//   LpLoss.rel / .abs (p = 2)       /root/reference/examples/neuraloperator/metric.py:69-176
//   H1Loss.rel / .abs (d = 2)       metric.py:196-383 with central_diff_2d :36-55 (periodic wrap-around, or one-sided
//                                   differences on the first / last row / column when fix_x_bnd / fix_y_bnd)
//   MSELoss.forward on fields       /root/reference/ppsci/loss/mse.py:82-105 (mode "sq")
// and the reverse pass the reference gets from autograd through them.  With e = x - y and the linear difference
// operators Dx, Dy:   S_diff = |e|^2 (+ |Dx e|^2 + |Dy e|^2),  S_y = |y|^2 (+ |Dx y|^2 + |Dy y|^2) per row, and
//   rel: term = sqrt(S_diff) / sqrt(S_y)      abs: term = sqrt(c S_diff)      sq: term = S_diff
//   (p = 1, order 2: S = sums of magnitudes, rel: S_diff / S_y, abs: c S_diff, adjoint field sign(e))
//   loss = coef * sum_rows term               d loss / d x = a_row (e + Dx^T Dx e + Dy^T Dy e)
// also the elementwise tanh of FNOBlocks' `stabilizer="tanh"` (fno_block.py:1199) and its adjoint.
#include "ppsci_common.h"

#include <math.h>

extern "C" void ppsci_set_error(const char* fmt, ...);

// D[k][i] of the 1-D difference operator on n samples with spacing 1/ih (n >= 3)
__device__ __forceinline__ float fl_dcoef(int k, int i, int n, float ih, int fix) {
  if (fix && k == 0) return i == 1 ? ih : (i == 0 ? -ih : 0.f);
  if (fix && k == n - 1) return i == n - 1 ? ih : (i == n - 2 ? -ih : 0.f);
  const int up = k + 1 == n ? 0 : k + 1, dn = k == 0 ? n - 1 : k - 1;
  return (i == up ? 0.5f * ih : 0.f) - (i == dn ? 0.5f * ih : 0.f);
}
// (D v)[k] for v given as a strided line
__device__ __forceinline__ float fl_dapply(const float* v, long long stride, int k, int n, float ih, int fix) {
  if (fix && k == 0) return (v[stride] - v[0]) * ih;
  if (fix && k == n - 1) return (v[(long long)(n - 1) * stride] - v[(long long)(n - 2) * stride]) * ih;
  const int up = k + 1 == n ? 0 : k + 1, dn = k == 0 ? n - 1 : k - 1;
  return (v[(long long)up * stride] - v[(long long)dn * stride]) * (0.5f * ih);
}

struct FieldArgs {
  const float* x;
  const float* y;
  float* sums;           // [rows][2]: S_diff, S_y
  const float* rowcoef;  // [rows] (adjoint)
  float* gx;             // [rows][H][W] (adjoint)
  int rows, H, W, order, fix_x, fix_y;
  float ihx, ihy;
};

// one workgroup per row; fixed-order block reduction (deterministic)
__global__ void __launch_bounds__(256) field_sums_kernel(FieldArgs a) {
  __shared__ float red[2][4];
  const int r = blockIdx.x;
  const long long P = (long long)a.H * a.W;
  const float* x = a.x + r * P;
  const float* y = a.y + r * P;
  float sd = 0.f, sy = 0.f;
  for (int p = threadIdx.x; p < (int)P; p += 256) {
    const float yv = y[p], e = x[p] - yv;
    if (a.order == 2) {  // p = 1: sums of magnitudes
      sd += fabsf(e);
      sy += fabsf(yv);
      continue;
    }
    sd += e * e;
    sy += yv * yv;
    if (a.order == 1) {
      const int i = p / a.W, j = p - i * a.W;  // (only the H1 terms need the position)
      const float dxx = fl_dapply(x + j, a.W, i, a.H, a.ihx, a.fix_x), dxy = fl_dapply(y + j, a.W, i, a.H, a.ihx, a.fix_x);
      const float dyx = fl_dapply(x + (long long)i * a.W, 1, j, a.W, a.ihy, a.fix_y);
      const float dyy = fl_dapply(y + (long long)i * a.W, 1, j, a.W, a.ihy, a.fix_y);
      sd += (dxx - dxy) * (dxx - dxy) + (dyx - dyy) * (dyx - dyy);
      sy += dxy * dxy + dyy * dyy;
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    sd += __shfl_xor(sd, off, 64);
    sy += __shfl_xor(sy, off, 64);
  }
  if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = sd, red[1][threadIdx.x >> 6] = sy;
  __syncthreads();
  if (threadIdx.x < 2) a.sums[2 * r + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

struct FieldFinishArgs {
  const float* sums;
  float* loss;     // [1]
  float* rowcoef;  // [rows] or null
  int rows, mode;
  float abs_const, coef;
};

// one workgroup: the rows' terms in index order, loss = coef * sum, and the adjoint coefficient of every row
__global__ void __launch_bounds__(64) field_finish_kernel(FieldFinishArgs a) {
  float part = 0.f;
  for (int r = threadIdx.x; r < a.rows; r += 64) {
    const float sd = a.sums[2 * r], sy = a.sums[2 * r + 1];
    float term, ar;
    if (a.mode == 0) {  // rel: sqrt(sd) / sqrt(sy)
      const float nd = sqrtf(sd), ny = sqrtf(sy);
      term = nd / ny;
      ar = 1.f / (nd * ny);
    } else if (a.mode == 1) {  // abs: sqrt(c sd)
      term = sqrtf(a.abs_const * sd);
      ar = a.abs_const / term;
    } else if (a.mode == 2) {  // sq
      term = sd;
      ar = 2.f;
    } else if (a.mode == 3) {  // rel, p = 1: sum|e| / sum|y|   (the adjoint field is sign(e))
      term = sd / sy;
      ar = 1.f / sy;
    } else {  // abs, p = 1: c sum|e|
      term = a.abs_const * sd;
      ar = a.abs_const;
    }
    part += term;
    if (a.rowcoef != nullptr) a.rowcoef[r] = a.coef * ar;
  }
  // lanes hold interleaved rows: sum them in lane order (fixed shape)
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off, 64);
  if (threadIdx.x == 0) a.loss[0] = a.coef * part;
}

// gx = rowcoef (e + Dx^T Dx e + Dy^T Dy e)
__global__ void __launch_bounds__(256) field_adjoint_kernel(FieldArgs a) {
  const long long P = (long long)a.H * a.W, total = (long long)a.rows * P;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int r = (int)(idx / P);
    const long long p = idx - (long long)r * P;
    const int i = (int)(p / a.W), j = (int)(p - (long long)i * a.W);
    const float* x = a.x + r * P;
    const float* y = a.y + r * P;
    float acc = x[p] - y[p];
    if (a.order == 2) acc = acc > 0.f ? 1.f : (acc < 0.f ? -1.f : 0.f);
    if (a.order == 1) {
      // (Dx^T g)[i] = sum_k D[k][i] g[k], g = Dx e down column j: the rows k that touch sample i are i-1, i, i+1 (wrapped)
      int ks[3] = {i == 0 ? a.H - 1 : i - 1, i, i + 1 == a.H ? 0 : i + 1};
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int k = ks[t];
        const float c = fl_dcoef(k, i, a.H, a.ihx, a.fix_x);
        if (c != 0.f) acc += c * (fl_dapply(x + j, a.W, k, a.H, a.ihx, a.fix_x) - fl_dapply(y + j, a.W, k, a.H, a.ihx, a.fix_x));
      }
      int kt[3] = {j == 0 ? a.W - 1 : j - 1, j, j + 1 == a.W ? 0 : j + 1};
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int k = kt[t];
        const float c = fl_dcoef(k, j, a.W, a.ihy, a.fix_y);
        const float* xr = x + (long long)i * a.W;
        const float* yr = y + (long long)i * a.W;
        if (c != 0.f) acc += c * (fl_dapply(xr, 1, k, a.W, a.ihy, a.fix_y) - fl_dapply(yr, 1, k, a.W, a.ihy, a.fix_y));
      }
    }
    a.gx[idx] = a.rowcoef[r] * acc;
  }
}

static int fl_check(int rows, int H, int W, int order, const void* x, const void* y) {
  if (rows < 1 || H < 1 || W < 1 || order < 0 || order > 2 || !x || !y || (order == 1 && (H < 3 || W < 3))) {
    ppsci_set_error("field_loss: invalid argument (order 1 needs at least 3 samples per axis)");
    return PPSCI_E_INVALID;
  }
  return PPSCI_OK;
}

extern "C" int ppsci_field_loss_sums(int rows, int H, int W, int order, float ihx, float ihy, int fix_x, int fix_y, const float* x,
                                     const float* y, float* sums, void* stream) {
  if (fl_check(rows, H, W, order, x, y) != PPSCI_OK || !sums) {
    ppsci_set_error("field_loss_sums: invalid argument");
    return PPSCI_E_INVALID;
  }
  FieldArgs a{x, y, sums, nullptr, nullptr, rows, H, W, order, fix_x, fix_y, ihx, ihy};
  PPSCI_LAUNCH(field_sums_kernel, FieldArgs, rows, 256, 0, stream, a);
  if (PPSCI_LAST_LAUNCH_ERROR() != 0) {
    ppsci_set_error("field_loss_sums: launch failed");
    return PPSCI_E_LAUNCH;
  }
  return PPSCI_OK;
}

extern "C" int ppsci_field_loss_finish(int rows, int mode, float abs_const, float coef, const float* sums, float* loss,
                                       float* rowcoef, void* stream) {
  if (rows < 1 || mode < 0 || mode > 4 || !sums || !loss) {
    ppsci_set_error("field_loss_finish: invalid argument");
    return PPSCI_E_INVALID;
  }
  FieldFinishArgs a{sums, loss, rowcoef, rows, mode, abs_const, coef};
  PPSCI_LAUNCH(field_finish_kernel, FieldFinishArgs, 1, 64, 0, stream, a);
  if (PPSCI_LAST_LAUNCH_ERROR() != 0) {
    ppsci_set_error("field_loss_finish: launch failed");
    return PPSCI_E_LAUNCH;
  }
  return PPSCI_OK;
}

extern "C" int ppsci_field_loss_adjoint(int rows, int H, int W, int order, float ihx, float ihy, int fix_x, int fix_y,
                                        const float* x, const float* y, const float* rowcoef, float* gx, void* stream) {
  if (fl_check(rows, H, W, order, x, y) != PPSCI_OK || !rowcoef || !gx) {
    ppsci_set_error("field_loss_adjoint: invalid argument");
    return PPSCI_E_INVALID;
  }
  FieldArgs a{x, y, nullptr, rowcoef, gx, rows, H, W, order, fix_x, fix_y, ihx, ihy};
  long long grid = ((long long)rows * H * W + 255) / 256;
  if (grid > 4096) grid = 4096;
  PPSCI_LAUNCH(field_adjoint_kernel, FieldArgs, (int)grid, 256, 0, stream, a);
  if (PPSCI_LAST_LAUNCH_ERROR() != 0) {
    ppsci_set_error("field_loss_adjoint: launch failed");
    return PPSCI_E_LAUNCH;
  }
  return PPSCI_OK;
}

// ---- FNOBlocks(stabilizer="tanh"): y = tanh(x) before the spectral convolution, out (+)= g (1 - y^2) behind it
struct TanhArgs {
  const float* x;
  const float* g;
  float* y;
  long long n;
  int accumulate;
};
__global__ void __launch_bounds__(256) tanh_fwd_kernel(TanhArgs a) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (long long)gridDim.x * 256) a.y[i] = tanhf(a.x[i]);
}
__global__ void __launch_bounds__(256) tanh_bwd_kernel(TanhArgs a) {  // x: the forward OUTPUT tanh(x)
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (long long)gridDim.x * 256) {
    const float t = a.x[i], v = a.g[i] * (1.f - t * t);
    a.y[i] = a.accumulate ? a.y[i] + v : v;
  }
}
extern "C" int ppsci_tanh_fwd(int64_t n, const float* x, float* y, void* stream) {
  if (n < 1 || !x || !y) {
    ppsci_set_error("tanh_fwd: invalid argument");
    return PPSCI_E_INVALID;
  }
  TanhArgs a{x, nullptr, y, n, 0};
  long long grid = (n + 255) / 256;
  if (grid > 4096) grid = 4096;
  PPSCI_LAUNCH(tanh_fwd_kernel, TanhArgs, (int)grid, 256, 0, stream, a);
  return PPSCI_LAST_LAUNCH_ERROR() != 0 ? PPSCI_E_LAUNCH : PPSCI_OK;
}
extern "C" int ppsci_tanh_bwd(int64_t n, const float* y, const float* g, float* out, int accumulate, void* stream) {
  if (n < 1 || !y || !g || !out) {
    ppsci_set_error("tanh_bwd: invalid argument");
    return PPSCI_E_INVALID;
  }
  TanhArgs a{y, g, out, n, accumulate};
  long long grid = (n + 255) / 256;
  if (grid > 4096) grid = 4096;
  PPSCI_LAUNCH(tanh_bwd_kernel, TanhArgs, (int)grid, 256, 0, stream, a);
  return PPSCI_LAST_LAUNCH_ERROR() != 0 ? PPSCI_E_LAUNCH : PPSCI_OK;
}
