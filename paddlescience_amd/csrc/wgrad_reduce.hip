// wgrad_reduce.hip -- fixed-order tree reduction of the per-tile hidden-weight gradient blocks that
// taylor_bwd streams to HBM (see taylor_bwd.inc).  Purely HBM-bound: reads ntiles*(L-1)*HP^2 floats
// once with float4 loads, coalesced.
#include "taylor_tile.h"

#include <string.h>

struct WRedArgs {
  const float* wpart;  // [ntiles][per_tile]
  float* tmp;          // [nchunks][per_tile]
  float* row;          // [P]
  const float* small;  // [nsmall_rows][psmall]: per-workgroup rows of the W0 | biases | W_last | b_last gradients
  float* tmp_small;    // [nchunks][psmall]: their chunk sums
  int nsmall_rows, psmall, nbs;  // nbs = workgroups per chunk for the small part
  int m, d0;
  ppsci_derived q;
  int L, H, ntiles, nchunks, nb4;  // nb4 = workgroups per chunk (each covers 256 float4)
  long long per_tile;              // (L-1)*HP*HP floats
  long long tmp_stride, small_stride, loss_stride;  // row strides of tmp / tmp_small / x.loss_rows (floats)
  ppsci_wred_extras x;             // stage 2: row (+)= / loss terms / Adam in the same launch (taylor_tile.h)
};

// stage 1: tmp[chunk][j] = sum_{tile in chunk} wpart[tile][j]; the workgroups behind those do the same for the
// per-workgroup rows of the small tensors (one launch instead of a separate row reduction)
__global__ void __launch_bounds__(256) wgrad_reduce1_kernel(WRedArgs a) {
  if ((int)blockIdx.x >= a.nchunks * a.nb4) {
    const int bid = (int)blockIdx.x - a.nchunks * a.nb4;
    const int chunk = bid / a.nbs;
    const int j = (bid - chunk * a.nbs) * 256 + threadIdx.x;
    if (j >= a.psmall) return;
    const int r0 = (int)((long long)a.nsmall_rows * chunk / a.nchunks);
    const int r1 = (int)((long long)a.nsmall_rows * (chunk + 1) / a.nchunks);
    float v = 0.f;
#pragma unroll 8
    for (int r = r0; r < r1; ++r) v += a.small[(long long)r * a.psmall + j];
    a.tmp_small[(long long)chunk * a.psmall + j] = v;
    return;
  }
  const int chunk = blockIdx.x / a.nb4;
  const long long j4 = (long long)(blockIdx.x - chunk * a.nb4) * 256 + threadIdx.x;
  if (j4 * 4 >= a.per_tile) return;
  const int t0 = (int)((long long)a.ntiles * chunk / a.nchunks);
  const int t1 = (int)((long long)a.ntiles * (chunk + 1) / a.nchunks);
  f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
  const f32x4* src = (const f32x4*)a.wpart + j4;
  const long long stride4 = a.per_tile / 4;
#pragma unroll 8
  for (int t = t0; t < t1; ++t) s += __builtin_nontemporal_load(&src[(long long)t * stride4]);
  ((f32x4*)a.tmp)[(long long)chunk * stride4 + j4] = s;
}

// stage 2: one thread per parameter of the row
__global__ void __launch_bounds__(256) wgrad_reduce2_kernel(WRedArgs a) {
  if ((int)blockIdx.x >= (a.q.P + 255) / 256) {
    // one more workgroup: the loss terms, rows summed in a fixed order (thread t: rows t, t + 256, ...; then the waves in order)
    __shared__ float red[4];
    for (int k = 0; k < a.x.n_res; ++k) {
      float v = 0.f;
#pragma unroll 4
      for (int r = threadIdx.x; r < a.x.loss_nrows; r += 256) v += a.x.loss_rows[(long long)r * a.loss_stride + k];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
      __syncthreads();
      if (threadIdx.x == 0) a.x.loss_out[k] = (red[0] + red[1]) + (red[2] + red[3]);
      __syncthreads();
    }
    return;
  }
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= a.q.P) return;
  float v = 0.f;
  bool hidden = false;
  // is idx inside a hidden-to-hidden weight matrix?
  for (int l = 1; l < a.L; ++l) {
    const int off = a.q.offW[l];
    if (idx >= off && idx < off + a.H * a.H) {
      hidden = true;
      const int e = idx - off;
      const int in = e / a.H, out = e - in * a.H;
      const int NB = a.q.NB, HP = a.q.HP;
      // block (ib, ob), lane 16g + c, component r  <->  (in = 16ib + 4g + r, out = 16ob + c)
      const long long j = (long long)(l - 1) * HP * HP +
                          ((long long)((in >> 4) * NB + (out >> 4)) * 64 + 16 * ((in & 15) >> 2) + (out & 15)) * 4 + (in & 3);
      // four interleaved chains (chunk c goes to chain c & 3), combined in a fixed order: 16 loads in flight per thread
      float v4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
      for (int c = 0; c < a.nchunks; c += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // clamped index + select: the loads stay unconditional (no branch between them)
          const int cu = c + u < a.nchunks ? c + u : a.nchunks - 1;
          const float t = a.tmp[(long long)cu * a.tmp_stride + j];
          v4[u] += c + u < a.nchunks ? t : 0.f;
        }
      }
      v = (v4[0] + v4[1]) + (v4[2] + v4[3]);
    }
  }
  if (!hidden) {  // W0, a bias, W_last or b_last: compact index into the summed small block
    const int H = a.H, L = a.L;
    int ci;
    if (idx >= a.q.offB[L]) ci = (a.d0 + L + a.m) * H + (idx - a.q.offB[L]);
    else if (idx >= a.q.offW[L]) ci = (a.d0 + L) * H + (idx - a.q.offW[L]);
    else if (idx < a.q.offW[0] + a.d0 * H) ci = idx - a.q.offW[0];
    else {
      int l = 0;
      while (l + 1 < L && idx >= a.q.offW[l + 1]) ++l;  // the bias that follows W_l
      ci = a.d0 * H + l * H + (idx - a.q.offB[l]);
    }
    float v4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int c = 0; c < a.nchunks; c += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int cu = c + u < a.nchunks ? c + u : a.nchunks - 1;
        const float t = a.tmp_small[(long long)cu * a.small_stride + ci];
        v4[u] += c + u < a.nchunks ? t : 0.f;
      }
    }
    v = (v4[0] + v4[1]) + (v4[2] + v4[3]);
  }
  if (a.x.accumulate) v += a.row[idx];
  a.row[idx] = v;
  if (a.x.p != nullptr) {  // == adam_kernel (epilogue_optim.hip)
    const float g = v * a.x.grad_scale;
    const float mm = a.x.beta1 * a.x.m[idx] + (1.f - a.x.beta1) * g;
    const float vv = a.x.beta2 * a.x.v[idx] + (1.f - a.x.beta2) * g * g;
    a.x.m[idx] = mm;
    a.x.v[idx] = vv;
    a.x.p[idx] = a.x.p[idx] - a.x.lr_t * (mm / (sqrtf(vv) + a.x.eps_t));
  }
}

// Stage 2 alone on rows that are chunk sums already (the first level of the in-kernel reduction tree of the fused tile
// kernel, taylor_step_tail.h): `rows` [nrows][rowlen] = hidden-weight blocks | small tensors at off_small | loss terms at
// off_loss.  Total, grad (+)= it, the loss terms and (x.p) the Adam update in ONE launch.
int ppsci_wgrad_reduce_chunks(const ppsci_mlp_desc& d, const ppsci_derived& q, int nrows, const float* rows, long long rowlen,
                              long long off_small, long long off_loss, float* row, const ppsci_wred_extras& x0, void* stream) {
  WRedArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x0;
  a.x.loss_rows = x0.loss_out ? rows + off_loss : nullptr;
  a.x.loss_nrows = nrows;
  a.psmall = ppsci_small_params(d, q);
  a.m = d.d_out;
  a.d0 = q.d0;
  a.tmp = (float*)rows;
  a.tmp_small = (float*)rows + off_small;
  a.row = row;
  a.q = q;
  a.L = d.n_hidden;
  a.H = d.width;
  a.nchunks = nrows;
  a.per_tile = (long long)(d.n_hidden - 1) * q.HP * q.HP;
  a.tmp_stride = a.small_stride = a.loss_stride = rowlen;
  PPSCI_LAUNCH(wgrad_reduce2_kernel, WRedArgs, (q.P + 255) / 256 + (a.x.loss_rows != nullptr ? 1 : 0), 256, 0, stream, a);
  if (PPSCI_LAST_LAUNCH_ERROR() != 0) {
    ppsci_set_error("wgrad_reduce2: launch failed");
    return PPSCI_E_LAUNCH;
  }
  return PPSCI_OK;
}

int ppsci_wgrad_reduce(const ppsci_mlp_desc& d, const ppsci_derived& q, int ntiles, const float* wpart, float* tmp,
                       const float* small_rows, int nsmall_rows, float* tmp_small, float* row, void* stream) {
  ppsci_wred_extras none;
  memset(&none, 0, sizeof(none));
  return ppsci_wgrad_reduce_ex(d, q, ntiles, wpart, tmp, small_rows, nsmall_rows, tmp_small, row, none, stream);
}

int ppsci_wgrad_reduce_ex(const ppsci_mlp_desc& d, const ppsci_derived& q, int ntiles, const float* wpart, float* tmp,
                          const float* small_rows, int nsmall_rows, float* tmp_small, float* row, const ppsci_wred_extras& x,
                          void* stream) {
  WRedArgs a;
  a.x = x;
  a.small = small_rows;
  a.tmp_small = tmp_small;
  a.nsmall_rows = nsmall_rows;
  a.psmall = ppsci_small_params(d, q);
  a.nbs = (a.psmall + 255) / 256;
  a.m = d.d_out;
  a.d0 = q.d0;
  a.wpart = wpart;
  a.tmp = tmp;
  a.row = row;
  a.q = q;
  a.L = d.n_hidden;
  a.H = d.width;
  a.ntiles = ntiles;
  a.nchunks = ntiles < PPSCI_WRED_CHUNKS ? ntiles : PPSCI_WRED_CHUNKS;
  if (a.nchunks < 1) a.nchunks = 1;
  a.per_tile = (long long)(d.n_hidden - 1) * q.HP * q.HP;
  a.tmp_stride = a.per_tile, a.small_stride = a.psmall, a.loss_stride = x.n_res;
  a.nb4 = a.per_tile > 0 ? (int)((a.per_tile / 4 + 255) / 256) : 0;
  {
    PPSCI_LAUNCH(wgrad_reduce1_kernel, WRedArgs, a.nchunks * (a.nb4 + a.nbs), 256, 0, stream, a);
    int e = PPSCI_LAST_LAUNCH_ERROR();
    if (e != 0) {
      ppsci_set_error("wgrad_reduce1: launch failed (hip error %d)", e);
      return PPSCI_E_LAUNCH;
    }
  }
  PPSCI_LAUNCH(wgrad_reduce2_kernel, WRedArgs, (q.P + 255) / 256 + (x.loss_rows != nullptr ? 1 : 0), 256, 0, stream, a);
  int e = PPSCI_LAST_LAUNCH_ERROR();
  if (e != 0) {
    ppsci_set_error("wgrad_reduce2: launch failed (hip error %d)", e);
    return PPSCI_E_LAUNCH;
  }
  return PPSCI_OK;
}
