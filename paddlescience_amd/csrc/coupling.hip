// coupling.hip -- the constant linear map of a batch-coupled residual.
//
// The reference evaluates expressions as tensor code, so one may couple the points of a batch:
//   Volterra.compute_volterra_func        /root/reference/ppsci/equation/ide/volterra.py:66-77
//       rhs = paddle.mm(int_mat, u);  volterra = lhs[:len(rhs)] - rhs        (int_mat: [N, N + N Q], quadrature weights x kernel)
// Everything else of such a residual is a per-point program (epilogue_vm.h); what is left is  y = M v  between two launches
// of it, and  vbar = -M^T (2 scale w r)  for the reverse sweep (engine.FusedConstraint._forward_couplings).  M is tiny
// (12 x 252 in examples/ide/volterra_ide.py): one workgroup per output row / one thread per output column, fixed summation
// order, nothing clever.
#include "ppsci_common.h"
#include "ppsci_hip.h"

extern "C" void ppsci_set_error(const char* fmt, ...);

struct MatvecArgs {
  const float* M;
  const float* x;
  const float* rowscale;
  float* y;
  long long rows, cols;
  float alpha;
};

// y[i] = alpha * sum_q M[i][q] x[q]: one workgroup per row i, lanes along q (coalesced), fixed-order tree in LDS
__global__ void __launch_bounds__(256) matvec_rows_kernel(MatvecArgs a) {
  __shared__ float red[256];
  const long long i = blockIdx.x;
  const float* row = a.M + i * a.cols;
  float s = 0.f;
  for (long long q = threadIdx.x; q < a.cols; q += 256) s += row[q] * a.x[q];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) a.y[i] = a.alpha * red[0];
}

// y[q] = alpha * sum_i M[i][q] x[i] (rowscale[i]): one thread per column q (consecutive lanes = consecutive q), rows in order
__global__ void __launch_bounds__(256) matvec_cols_kernel(MatvecArgs a) {
  const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
  if (q >= a.cols) return;
  float s = 0.f;
  for (long long i = 0; i < a.rows; ++i) {
    float xi = a.x[i];
    if (a.rowscale != nullptr) xi *= a.rowscale[i];
    s += a.M[i * a.cols + q] * xi;
  }
  a.y[q] = a.alpha * s;
}

extern "C" int ppsci_dense_matvec(int64_t rows, int64_t cols, const float* M, const float* x, const float* rowscale, float alpha,
                                  int transpose, float* y, void* stream) {
  if (rows <= 0 || cols <= 0 || !M || !x || !y || rows > (1LL << 30) || cols > (1LL << 30)) {
    ppsci_set_error("dense_matvec: invalid argument");
    return PPSCI_E_INVALID;
  }
  if (!transpose && rowscale) {
    ppsci_set_error("dense_matvec: rowscale belongs to the transposed product");
    return PPSCI_E_INVALID;
  }
  MatvecArgs a;
  a.M = M;
  a.x = x;
  a.rowscale = rowscale;
  a.y = y;
  a.rows = rows;
  a.cols = cols;
  a.alpha = alpha;
  if (transpose) PPSCI_LAUNCH(matvec_cols_kernel, MatvecArgs, (int)((cols + 255) / 256), 256, 0, stream, a);
  else PPSCI_LAUNCH(matvec_rows_kernel, MatvecArgs, (int)rows, 256, 0, stream, a);
  const int e = PPSCI_LAST_LAUNCH_ERROR();
  if (e != 0) {
    ppsci_set_error("dense_matvec: launch failed (hip error %d)", e);
    return PPSCI_E_LAUNCH;
  }
  return PPSCI_OK;
}
