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

// ---- ONE launch behind the fused tile kernel (taylor_fused.inc): fixed-order sum of the workgroups' rows, grad (+)= it, the
// loss terms, the Adam update AND the bf16 fragments of the updated hidden matrices for the next step's launch (so that a
// step is two launches -- tile kernel + this one -- instead of weight split + tile kernel + two reduction kernels).
//   workgroup b < nblk: one 16 x 16 block of a hidden-to-hidden matrix = 64 float4 columns of the rows' block layout
//     (block (ib, ob), lane 16g + c, component r  <->  in = 16ib + 4g + r, out = 16ob + c).  1024 threads = 16 waves: wave w
//     sums rows w, w + 16, ... (four interleaved chains, 8 loads in flight per lane), the waves' sums meet in LDS and wave 0
//     adds them in wave order; its 64 lanes then own the block's 256 parameters: gradient, Adam, and both fragment
//     layouts (forward: the lane's own four values; backward: the transposed block through a 16 x 17 LDS tile);
//   the workgroups behind those: 64 columns each of the small tensors' compact rows; the last one: the loss terms.
struct WTailArgs {
  const float* rows_w;  // [nrows][per_tile]
  const float* rows_s;  // [nrows][psmall]
  float* row;           // [P] the gradient
  u32x4* frag;          // forward | backward fragments of the hidden matrices (ppsci_presplit2_kernel's layout); null: none
  ppsci_derived q;
  int L, H, m, d0, nrows, psmall, nblk, nsmall;
  long long per_tile, loss_stride;
  ppsci_wred_extras x;
};

__device__ __forceinline__ float wtail_adam(const ppsci_wred_extras& x, int idx, float gsum) {
  // == adam_kernel (epilogue_optim.hip); returns the new parameter
  const float g = gsum * x.grad_scale;
  const float mm = x.beta1 * x.m[idx] + (1.f - x.beta1) * g;
  const float vv = x.beta2 * x.v[idx] + (1.f - x.beta2) * g * g;
  x.m[idx] = mm;
  x.v[idx] = vv;
  const float pn = x.p[idx] - x.lr_t * (mm / (sqrtf(vv) + x.eps_t));
  x.p[idx] = pn;
  return pn;
}

__global__ void __launch_bounds__(1024) wgrad_tail_kernel(WTailArgs a) {
  __shared__ f32x4 red[16][64];
  __shared__ float tile[16][17];
  const int tid = threadIdx.x, lane = tid & 63, rg = tid >> 6;
  const int b = blockIdx.x;
  if (b < a.nblk) {
    const long long stride4 = a.per_tile / 4;
    const f32x4* src = (const f32x4*)a.rows_w + (long long)b * 64 + lane;
    f32x4 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int r0 = rg; r0 < a.nrows; r0 += 64) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {  // clamped index + select: the loads stay unconditional
        const int r = r0 + 16 * u;
        const f32x4 v = __builtin_nontemporal_load(&src[(long long)(r < a.nrows ? r : a.nrows - 1) * stride4]);
        acc[u] += r < a.nrows ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
    red[rg][lane] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    if (rg != 0) return;
    f32x4 tot = red[0][lane];
#pragma unroll
    for (int w = 1; w < 16; ++w) tot += red[w][lane];
    const int NB = a.q.NB, NKP = NB / 2;
    const int l = b / (NB * NB), blk = b - l * NB * NB, ib = blk / NB, ob = blk - ib * NB;
    const int g = lane >> 4, c = lane & 15;
    const int in0 = 16 * ib + 4 * g, out = 16 * ob + c;
    f32x4 pn = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (in0 + r < a.H && out < a.H) {
        const int idx = a.q.offW[l + 1] + (in0 + r) * a.H + out;
        float v = tot[r];
        if (a.x.accumulate) v += a.row[idx];
        a.row[idx] = v;
        if (a.x.p != nullptr) pn[r] = wtail_adam(a.x, idx, v);
      }
    }
    if (a.x.p == nullptr || a.frag == nullptr) return;
    // forward fragment (z = W^T h): row block ob, k-block ib -- the lane's own four values
    const long long per_layer = (long long)NB * NKP * 3 * 64;  // u32x4 units
    {
      const ppsci_split4 sp = ppsci_split(pn);
      u32x2* dst = (u32x2*)(a.frag + (long long)l * per_layer + (long long)((ob * NKP + (ib >> 1)) * 3) * 64 + lane) + (ib & 1);
#pragma unroll
      for (int p = 0; p < 3; ++p) dst[(long long)p * 64 * 2] = sp.p[p];
    }
    // backward fragment (hbar = W zbar): row block ib, k-block ob, lane (g, c) holds W[16 ib + c][16 ob + 4g + r]
#pragma unroll
    for (int r = 0; r < 4; ++r) tile[4 * g + r][c] = pn[r];
    ppsci_wave_sync();
    f32x4 pt;
#pragma unroll
    for (int r = 0; r < 4; ++r) pt[r] = tile[c][4 * g + r];
    {
      const ppsci_split4 sp = ppsci_split(pt);
      u32x2* dst = (u32x2*)(a.frag + (long long)(a.L - 1 + l) * per_layer + (long long)((ib * NKP + (ob >> 1)) * 3) * 64 + lane) + (ob & 1);
#pragma unroll
      for (int p = 0; p < 3; ++p) dst[(long long)p * 64 * 2] = sp.p[p];
    }
    return;
  }
  if (b < a.nblk + a.nsmall) {
    float* redf = (float*)red;  // [16][64]
    const int ci = (b - a.nblk) * 64 + lane;
    const int cc = ci < a.psmall ? ci : 0;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int r0 = rg; r0 < a.nrows; r0 += 64) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = r0 + 16 * u;
        const float v = a.rows_s[(long long)(r < a.nrows ? r : a.nrows - 1) * a.psmall + cc];
        acc[u] += r < a.nrows ? v : 0.f;
      }
    }
    redf[rg * 64 + lane] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    if (rg != 0 || ci >= a.psmall) return;
    float v = redf[lane];
#pragma unroll
    for (int w = 1; w < 16; ++w) v += redf[w * 64 + lane];
    // compact order of ppsci_small_params: W0 | b_0 .. b_{L-1} | W_last | b_last (| activation parameters)
    const int H = a.H, L = a.L;
    int idx;
    if (ci < a.d0 * H) idx = a.q.offW[0] + ci;
    else if (ci < (a.d0 + L) * H) {
      const int lb = (ci - a.d0 * H) / H;
      idx = a.q.offB[lb] + (ci - a.d0 * H - lb * H);
    } else if (ci < (a.d0 + L + a.m) * H) idx = a.q.offW[L] + (ci - (a.d0 + L) * H);
    else idx = a.q.offB[L] + (ci - (a.d0 + L + a.m) * H);
    if (a.x.accumulate) v += a.row[idx];
    a.row[idx] = v;
    if (a.x.p != nullptr) wtail_adam(a.x, idx, v);
    return;
  }
  // the loss terms: rows summed in a fixed order (thread t: rows t, t + 1024, ...; then the waves in order)
  float* redf = (float*)red;
  for (int k = 0; k < a.x.n_res; ++k) {
    float v = 0.f;
#pragma unroll 4
    for (int r = tid; r < a.x.loss_nrows; r += 1024) v += a.x.loss_rows[(long long)r * a.loss_stride + k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) redf[rg] = v;
    __syncthreads();
    if (tid == 0) {
      float s = redf[0];
      for (int w = 1; w < 16; ++w) s += redf[w];
      a.x.loss_out[k] = s;
    }
    __syncthreads();
  }
}

int ppsci_wgrad_tail(const ppsci_mlp_desc& d, const ppsci_derived& q, int nrows, const float* rows_w, const float* rows_s,
                     float* row, const ppsci_wred_extras& x, void* frag, void* stream) {
  WTailArgs a;
  memset(&a, 0, sizeof(a));
  a.rows_w = rows_w;
  a.rows_s = rows_s;
  a.row = row;
  a.frag = (u32x4*)frag;
  a.q = q;
  a.L = d.n_hidden;
  a.H = d.width;
  a.m = d.d_out;
  a.d0 = q.d0;
  a.nrows = nrows;
  a.psmall = ppsci_small_params(d, q);
  a.nblk = (d.n_hidden - 1) * q.NB * q.NB;
  a.nsmall = (a.psmall + 63) / 64;
  a.per_tile = (long long)(d.n_hidden - 1) * q.HP * q.HP;
  a.loss_stride = x.n_res;
  a.x = x;
  PPSCI_LAUNCH(wgrad_tail_kernel, WTailArgs, a.nblk + a.nsmall + (x.loss_rows != nullptr ? 1 : 0), 1024, 0, stream, a);
  if (PPSCI_LAST_LAUNCH_ERROR() != 0) {
    ppsci_set_error("wgrad_tail: launch failed");
    return PPSCI_E_LAUNCH;
  }
  return PPSCI_OK;
}
