// spinn.hip -- separable PINN (BASELINE config 5): branch nets + tensor-product grid contraction.
//
// Replaces, per constraint of /root/reference/examples/spinn/helmholtz3d.py,
//   ModifiedMLP.forward_tensor     /root/reference/ppsci/arch/mlp.py:488-507   (one net per axis)
//   SPINN.forward_tensor           /root/reference/ppsci/arch/spinn.py:140-167 (broadcast products + sum over rank)
//   Helmholtz.helmholtz / hvp_revrev /root/reference/ppsci/equation/pde/helmholtz.py:27-41,78-93
//                                  (three nested-jvp double-backward passes, each materialising [N,N,N,r])
//   MSELoss + backward()           ppsci/loss/mse.py:82-105, ppsci/solver/train.py:158
//
// (1) modmlp_fwd/bwd: a branch net maps ONE coordinate to r*m features; value / first / second derivative
//     streams w.r.t. that coordinate are carried in Taylor mode through
//         u = act(x Wu + bu), v = act(x Wv + bv);   y <- act(y W_l + b_l);  y <- y*u + (1-y)*v;   f = y W_last + b
//     The nets see only N (=128) points per axis, so these are small VALU kernels: one workgroup per point,
//     one thread per feature, activations exchanged through LDS.
// (2) spinn_grid_fwd/bwd: on the Nx*Ny*Nz grid,  q(i,j,k) = sum_r fx[i,r] fy[j,r] fz[k,r]  for q in
//     {u, u_xx, u_yy, u_zz} (second-derivative stream of the corresponding axis), residual
//     res = cu*u + cxx*u_xx + cyy*u_yy + czz*u_zz, fused weighted MSE and its adjoint; the reverse kernel
//     reduces the adjoint over the two other axes.  HBM-bound: the label grid (4 B/pt) is read once, the
//     adjoint grid written once and read three times; nothing of size [N,N,N,r] is ever materialised.
#include "ppsci_common.h"

#ifndef PPSCI_EMU
#include <hip/hip_runtime.h>
#endif
#include <string.h>

extern "C" void ppsci_set_error(const char* fmt, ...);

#define SP_MAXH 128
#define SP_MAXR 64

// activation value / derivatives, same formulas as taylor_tile.h (kept local: this TU has no MFMA code)
__device__ __forceinline__ void sp_act(int act, float z, float& s, float& d1, float& d2, float& d3) {
  if (act == PPSCI_ACT_TANH) {
    s = tanhf(z);
    d1 = 1.f - s * s;
    d2 = -2.f * s * d1;
    d3 = d1 * (6.f * s * s - 2.f);
  } else if (act == PPSCI_ACT_SILU) {
    float g = 1.f / (1.f + expf(-z));
    float g1 = g * (1.f - g);
    float t = 1.f - 2.f * g;
    s = z * g;
    d1 = g + z * g1;
    d2 = g1 * (2.f + z * t);
    d3 = 3.f * g1 * t + z * (g1 * t * t - 2.f * g1 * g1);
  } else {
    s = sinf(z);
    float c = cosf(z);
    d1 = c;
    d2 = -s;
    d3 = -c;
  }
}

// parameter offsets of a 1-input ModifiedMLP in parameters() order:
//   embed_u.W[1,H] embed_u.b[H] embed_v.W[1,H] embed_v.b[H] linears.l.W linears.l.b ... last_fc.W[H,R] last_fc.b[R]
struct ModOff {
  int wu, bu, wv, bv, w[PPSCI_MAX_HIDDEN], b[PPSCI_MAX_HIDDEN], wl, bl, P;
};

static inline void mod_offsets(const ppsci_modmlp_desc& d, ModOff& o) {
  const int H = d.width;
  int off = 0;
  o.wu = off; off += H;
  o.bu = off; off += H;
  o.wv = off; off += H;
  o.bv = off; off += H;
  int fin = 1;
  for (int l = 0; l < d.n_hidden; ++l) {
    o.w[l] = off; off += fin * H;
    o.b[l] = off; off += H;
    fin = H;
  }
  o.wl = off; off += H * d.d_out;
  o.bl = off; off += d.d_out;
  o.P = off;
}

#define SP_MAXBATCH 3
struct ModBranch {     // one network (one SPINN axis) and its points
  const float* params;
  const float* x;      // [N]
  float* F;            // fwd out: [3][N][R]
  const float* Fbar;   // bwd in : [3][N][R]
  float* stash;        // [N][(L+2)][3][H]: zu, zv, z_0 .. z_{L-1}
  float* partials;     // bwd out: [N][P], row stride `pstride` floats
  long long pstride;
  int N;
  const float* fpart;  // tile kernel: dL/dF as the grid kernel's group partials [N][ngrp][2][R] (null: `Fbar` holds the sums)
  int ngrp;
};
struct ModArgs {       // up to SP_MAXBATCH networks of the same shape in one launch: workgroup = (branch, point)
  ppsci_modmlp_desc d;
  ModOff o;
  ModBranch br[SP_MAXBATCH];
  int nbatch;
};

__device__ __forceinline__ int mod_locate(const ModArgs& a, int& pt) {
  int bi = 0;
  while (bi + 1 < a.nbatch && pt >= a.br[bi].N) { pt -= a.br[bi].N; ++bi; }
  return bi;
}

// streams of the gated layer output o = v + a*(u - v) from the activation streams a and the embeddings
__device__ __forceinline__ void sp_gate(const float a[3], const float U[3], const float V[3], float o[3]) {
  const float D0 = U[0] - V[0], D1 = U[1] - V[1], D2 = U[2] - V[2];
  o[0] = V[0] + a[0] * D0;
  o[1] = V[1] + a[1] * D0 + a[0] * D1;
  o[2] = V[2] + a[2] * D0 + 2.f * a[1] * D1 + a[0] * D2;
}

__device__ __forceinline__ void sp_act_streams(int act, const float z[3], float a[3], float& d1, float& d2, float& d3) {
  float s;
  sp_act(act, z[0], s, d1, d2, d3);
  a[0] = s;
  a[1] = d1 * z[1];
  a[2] = d2 * z[1] * z[1] + d1 * z[2];
}

// [rows][cols] row-major global matrix -> LDS with row stride cols + 1 (thread = row reads are then conflict-free).
// 16 coalesced loads are issued before the first LDS write (a one-wave workgroup has nothing else to hide the
// latency behind); (row, col) of the running element are tracked incrementally -- no integer division.
__device__ __forceinline__ void sp_stage_rows(float* ws, const float* src, int rows, int cols, int f, int nthr) {
  const int total = rows * cols;
  int k = 0, j = f;
  while (j >= cols) { j -= cols; ++k; }
  for (int base = 0; base < total; base += 16 * nthr) {
    float tmp[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int idx = base + u * nthr + f;
      tmp[u] = idx < total ? src[idx] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int idx = base + u * nthr + f;
      if (idx < total) ws[k * (cols + 1) + j] = tmp[u];
      j += nthr;
      while (j >= cols) { j -= cols; ++k; }
    }
  }
}

__global__ void __launch_bounds__(SP_MAXH) modmlp_fwd_kernel(ModArgs a) {
  PPSCI_DYN_SMEM(sh);  // [3][H]
  const int f = threadIdx.x, H = a.d.width, L = a.d.n_hidden, R = a.d.d_out, act = a.d.activation;
  int pt = blockIdx.x;
  const ModBranch& B = a.br[mod_locate(a, pt)];
  const float x = B.x[pt];
  const float* P = B.params;
  float* st = B.stash ? B.stash + (long long)pt * (L + 2) * 3 * H : nullptr;
  float U[3] = {0, 0, 0}, V[3] = {0, 0, 0}, o[3] = {0, 0, 0};
  if (f < H) {
    float z[3], d1, d2, d3;
    z[0] = x * P[a.o.wu + f] + P[a.o.bu + f]; z[1] = P[a.o.wu + f]; z[2] = 0.f;
    if (st) { st[0 * H + f] = z[0]; st[1 * H + f] = z[1]; st[2 * H + f] = z[2]; }
    sp_act_streams(act, z, U, d1, d2, d3);
    z[0] = x * P[a.o.wv + f] + P[a.o.bv + f]; z[1] = P[a.o.wv + f]; z[2] = 0.f;
    if (st) { st[(3 + 0) * H + f] = z[0]; st[(3 + 1) * H + f] = z[1]; st[(3 + 2) * H + f] = z[2]; }
    sp_act_streams(act, z, V, d1, d2, d3);
  }
  for (int l = 0; l < L; ++l) {
    float z[3] = {0, 0, 0};
    if (l == 0) {
      if (f < H) {
        const float w = P[a.o.w[0] + f];
        z[0] = x * w + P[a.o.b[0] + f]; z[1] = w; z[2] = 0.f;
      }
    } else {
      __syncthreads();
      if (f < H) { sh[f] = o[0]; sh[H + f] = o[1]; sh[2 * H + f] = o[2]; }
      __syncthreads();
      if (f < H) {
        const float* W = P + a.o.w[l];
        float s0 = P[a.o.b[l] + f], s1 = 0.f, s2 = 0.f;
#pragma unroll 16
        for (int k = 0; k < H; ++k) {
          const float w = W[k * H + f];
          s0 += w * sh[k]; s1 += w * sh[H + k]; s2 += w * sh[2 * H + k];
        }
        z[0] = s0; z[1] = s1; z[2] = s2;
      }
    }
    if (f < H) {
      if (st) { float* q = st + (2 + l) * 3 * H; q[f] = z[0]; q[H + f] = z[1]; q[2 * H + f] = z[2]; }
      float av[3], d1, d2, d3;
      sp_act_streams(act, z, av, d1, d2, d3);
      sp_gate(av, U, V, o);
    }
  }
  __syncthreads();
  if (f < H) { sh[f] = o[0]; sh[H + f] = o[1]; sh[2 * H + f] = o[2]; }
  __syncthreads();
  for (int r = f; r < R; r += blockDim.x) {
    const float* W = P + a.o.wl;
    float s0 = P[a.o.bl + r], s1 = 0.f, s2 = 0.f;
#pragma unroll 16
    for (int k = 0; k < H; ++k) {
      const float w = W[k * R + r];
      s0 += w * sh[k]; s1 += w * sh[H + k]; s2 += w * sh[2 * H + k];
    }
    B.F[((long long)0 * B.N + pt) * R + r] = s0;
    B.F[((long long)1 * B.N + pt) * R + r] = s1;
    B.F[((long long)2 * B.N + pt) * R + r] = s2;
  }
}

// 256 threads per point: the first H own a feature each, all of them move the weight tiles and write the per-point
// weight gradients.
#define MOD_BWD_BLOCK 256
__global__ void __launch_bounds__(MOD_BWD_BLOCK) modmlp_bwd_kernel(ModArgs a) {
  PPSCI_DYN_SMEM(sh);  // [3][H] exchange + [3][R] Fbar + [H][max(H, R) + 1] weight rows
  const int f = threadIdx.x, H = a.d.width, L = a.d.n_hidden, R = a.d.d_out, act = a.d.activation;
  int pt = blockIdx.x;
  const ModBranch& B = a.br[mod_locate(a, pt)];
  const float x = B.x[pt];
  const float* P = B.params;
  const float* st = B.stash + (long long)pt * (L + 2) * 3 * H;
  float* G = B.partials + (long long)pt * B.pstride;
  float* fb = sh + 3 * H;
  for (int i = f; i < 3 * R; i += blockDim.x) fb[i] = B.Fbar[((long long)(i / R) * B.N + pt) * R + (i % R)];
  // embeddings (recomputed)
  float U[3] = {0, 0, 0}, V[3] = {0, 0, 0}, zu[3] = {0, 0, 0}, zv[3] = {0, 0, 0};
  float du1 = 0, du2 = 0, du3 = 0, dv1 = 0, dv2 = 0, dv3 = 0;
  if (f < H) {
    zu[0] = st[f]; zu[1] = st[H + f]; zu[2] = st[2 * H + f];
    zv[0] = st[3 * H + f]; zv[1] = st[4 * H + f]; zv[2] = st[5 * H + f];
    sp_act_streams(act, zu, U, du1, du2, du3);
    sp_act_streams(act, zv, V, dv1, dv2, dv3);
  }
  // gated output of the last hidden layer (needed for the last_fc weight gradient)
  float zl[3] = {0, 0, 0}, al[3] = {0, 0, 0}, ol[3] = {0, 0, 0}, d1 = 0, d2 = 0, d3 = 0;
  if (f < H) {
    const float* q = st + (2 + (L - 1)) * 3 * H;
    zl[0] = q[f]; zl[1] = q[H + f]; zl[2] = q[2 * H + f];
    sp_act_streams(act, zl, al, d1, d2, d3);
    sp_gate(al, U, V, ol);
  }
  // last_fc: obar_s[k] = sum_r WL[k][r] Fbar_s[r];  gWL[k][r] = sum_s o_s[k] Fbar_s[r];  gbL[r] = Fbar_0[r]
  // Weight rows are read by "thread = row": staged through LDS (coalesced global read, row stride + 1 -> conflict-
  // free row reads) instead of 64 different cache lines per load.
  float* shz = fb + 3 * R;  // [3][H] zbar of the layer
  float* ws = shz + 3 * H;  // [H][max(H, R) + 1]
  const int hstep = (int)blockDim.x / H, hr = f / H, hc = f - hr * H;  // blockDim >= H
  if (f < H) { sh[f] = ol[0]; sh[H + f] = ol[1]; sh[2 * H + f] = ol[2]; }
  // (thread -> (row, column) of a [H][R] tile with ONE integer division: a wave has nothing to hide them behind)
  const int rstep = (int)blockDim.x / R;  // rows per trip; 0: a row is wider than the workgroup
  const int fr = rstep > 0 ? f / R : 0, fc = rstep > 0 ? f - fr * R : f;
  const int rinc = rstep > 0 ? rstep : 1, cinc = rstep > 0 ? R : (int)blockDim.x;
  const bool live = rstep == 0 || fr < rstep;
  sp_stage_rows(ws, P + a.o.wl, H, R, f, (int)blockDim.x);
  __syncthreads();
  float ob[3] = {0, 0, 0};
  if (live) {
#pragma unroll 4
    for (int k = fr; k < H; k += rinc)
      for (int r = fc; r < R; r += cinc)
        G[a.o.wl + k * R + r] = sh[k] * fb[r] + sh[H + k] * fb[R + r] + sh[2 * H + k] * fb[2 * R + r];
  }
  if (f < H) {
    const float* W = ws + f * (R + 1);
    for (int r = 0; r < R; ++r) {
      const float w = W[r];
      ob[0] += w * fb[r]; ob[1] += w * fb[R + r]; ob[2] += w * fb[2 * R + r];
    }
  }
  for (int r = f; r < R; r += blockDim.x) G[a.o.bl + r] = fb[r];
  float Ub[3] = {0, 0, 0}, Vb[3] = {0, 0, 0};
  for (int l = L - 1; l >= 0; --l) {
    // adjoint of the gate  o = V + a*D
    float zb[3] = {0, 0, 0};
    if (f < H) {
      const float D0 = U[0] - V[0], D1 = U[1] - V[1], D2 = U[2] - V[2];
      const float ab2 = ob[2] * D0;
      const float ab1 = ob[1] * D0 + 2.f * ob[2] * D1;
      const float ab0 = ob[0] * D0 + ob[1] * D1 + ob[2] * D2;
      const float Db0 = ob[0] * al[0] + ob[1] * al[1] + ob[2] * al[2];
      const float Db1 = ob[1] * al[0] + 2.f * ob[2] * al[1];
      const float Db2 = ob[2] * al[0];
      Ub[0] += Db0; Ub[1] += Db1; Ub[2] += Db2;
      Vb[0] += ob[0] - Db0; Vb[1] += ob[1] - Db1; Vb[2] += ob[2] - Db2;
      // adjoint of the activation streams
      zb[2] = d1 * ab2;
      zb[1] = d1 * ab1 + 2.f * d2 * zl[1] * ab2;
      zb[0] = d1 * ab0 + d2 * zl[1] * ab1 + (d3 * zl[1] * zl[1] + d2 * zl[2]) * ab2;
      G[a.o.b[l] + f] = zb[0];
    }
    if (l == 0) {
      if (f < H) G[a.o.w[0] + f] = x * zb[0] + zb[1];  // input streams (x, 1, 0)
      break;
    }
    // previous layer's gated output (recomputed) -> LDS, this layer's zbar -> LDS
    float zp[3] = {0, 0, 0}, ap[3] = {0, 0, 0}, op[3] = {0, 0, 0}, p1 = 0, p2 = 0, p3 = 0;
    if (f < H) {
      const float* q = st + (2 + (l - 1)) * 3 * H;
      zp[0] = q[f]; zp[1] = q[H + f]; zp[2] = q[2 * H + f];
      sp_act_streams(act, zp, ap, p1, p2, p3);
      sp_gate(ap, U, V, op);
    }
    __syncthreads();
    if (f < H) {
      sh[f] = op[0]; sh[H + f] = op[1]; sh[2 * H + f] = op[2];
      shz[f] = zb[0]; shz[H + f] = zb[1]; shz[2 * H + f] = zb[2];
    }
    sp_stage_rows(ws, P + a.o.w[l], H, H, f, (int)blockDim.x);
    __syncthreads();
    if (hr < hstep) {  // gW_l[k][j] = sum_s o_prev_s[k] zbar_s[j]   (all threads: thread = (row phase, column j))
      float* gw = G + a.o.w[l];
      const float z0 = shz[hc], z1 = shz[H + hc], z2 = shz[2 * H + hc];
#pragma unroll 4
      for (int k = hr; k < H; k += hstep) gw[k * H + hc] = sh[k] * z0 + sh[H + k] * z1 + sh[2 * H + k] * z2;
    }
    if (f < H) {  // obar_prev_s[k=f] = sum_j W_l[f][j] zbar_s[j]
      const float* W = ws + f * (H + 1);
      float s0 = 0.f, s1 = 0.f, s2 = 0.f;
      for (int j = 0; j < H; ++j) {
        const float w = W[j];
        s0 += w * shz[j]; s1 += w * shz[H + j]; s2 += w * shz[2 * H + j];
      }
      ob[0] = s0; ob[1] = s1; ob[2] = s2;
      zl[0] = zp[0]; zl[1] = zp[1]; zl[2] = zp[2];
      al[0] = ap[0]; al[1] = ap[1]; al[2] = ap[2];
      d1 = p1; d2 = p2; d3 = p3;
    }
  }
  // embeddings: U = act(zu), V = act(zv) with streams (x w + b, w, 0)
  if (f < H) {
    float t0 = du1 * Ub[0] + du2 * zu[1] * Ub[1] + (du3 * zu[1] * zu[1] + du2 * zu[2]) * Ub[2];
    float t1 = du1 * Ub[1] + 2.f * du2 * zu[1] * Ub[2];
    G[a.o.bu + f] = t0;
    G[a.o.wu + f] = x * t0 + t1;
    t0 = dv1 * Vb[0] + dv2 * zv[1] * Vb[1] + (dv3 * zv[1] * zv[1] + dv2 * zv[2]) * Vb[2];
    t1 = dv1 * Vb[1] + 2.f * dv2 * zv[1] * Vb[2];
    G[a.o.bv + f] = t0;
    G[a.o.wv + f] = x * t0 + t1;
  }
}

// ---- the reverse sweep of the branch nets, 16 points per workgroup on the matrix cores -------------------------------
// modmlp_bwd_kernel above gives every POINT a workgroup: each one stages every weight matrix for itself and writes a full
// gradient row of its own (Helmholtz3D, 3 x 128 points on 4 x 64 nets: 26 MB of per-point rows per step, summed again by the
// next launch; 34 us of a 100 us step).  Here a workgroup owns a 16-point TILE of one branch, in the tile model of the
// Taylor kernels (taylor_tile.h): wave w holds feature block w of every quantity as float4 registers per stream in the C/D
// layout of v_mfma_f32_16x16x4_f32 (lane (g, c): features 16w + 4g + r, point c), so that
//   obar_{l-1} = W_l zbar_l          A = W_l[16w + c][16jb + 4g + r] (one float4 load per block jb, straight from L2),
//                                    B = zbar's block jb (k-step (jb, r): every lane supplies its own register r);
//   gW_l      = o_{l-1} zbar_l^T     contraction over (stream, point): both operands read back TRANSPOSED from two LDS
//                                    planes [stream][feature][point] (row stride 17), 12 MFMAs per 16 x 16 block;
// and the activation / gate adjoints are elementwise on the registers.  Bias and first-layer gradients are sums over the
// tile's points: 4 xor-shuffles inside a 16-lane row.  One gradient row per TILE (1/16 of the rows to write and to sum).
// Needs width and rank to be multiples of 16, at most 64; other shapes keep the per-point kernel.
#define MODT_BLOCK 256
#define MODT_LD 17
#define FSUM_PARTS 8  // spinn_fbar_sum_kernel's interleaved subsets of the group partials (its summation order is kept here)

__device__ __forceinline__ float sp_sum16(float v) {  // over the 16 lanes c of a lane group g
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  return v;
}

__device__ __forceinline__ void sp_act_streams4(int act, const f32x4 z[3], f32x4 a[3], f32x4& d1, f32x4& d2, f32x4& d3) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float zz[3] = {z[0][r], z[1][r], z[2][r]};
    float aa[3], e1, e2, e3;
    sp_act_streams(act, zz, aa, e1, e2, e3);
    a[0][r] = aa[0]; a[1][r] = aa[1]; a[2][r] = aa[2];
    d1[r] = e1; d2[r] = e2; d3[r] = e3;
  }
}

__device__ __forceinline__ void sp_gate4(const f32x4 a[3], const f32x4 U[3], const f32x4 V[3], f32x4 o[3]) {
  const f32x4 D0 = U[0] - V[0], D1 = U[1] - V[1], D2 = U[2] - V[2];
  o[0] = V[0] + a[0] * D0;
  o[1] = V[1] + a[1] * D0 + a[0] * D1;
  o[2] = V[2] + a[2] * D0 + 2.f * a[1] * D1 + a[0] * D2;
}

__global__ void __launch_bounds__(MODT_BLOCK) modmlp_bwd_tile_kernel(ModArgs a) {
  PPSCI_DYN_SMEM(sh);  // To[3][H][MODT_LD] | Tz[3][H][MODT_LD]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int H = a.d.width, L = a.d.n_hidden, R = a.d.d_out, act = a.d.activation, NB = H / 16, RB = R / 16;
  int tile = blockIdx.x, bi = 0;
  while (bi + 1 < a.nbatch && tile >= (a.br[bi].N + 15) / 16) { tile -= (a.br[bi].N + 15) / 16; ++bi; }
  const ModBranch& B = a.br[bi];
  const int N = B.N, pt0 = tile * 16, pt = pt0 + c;
  const bool valid = pt < N;
  const int ptc = valid ? pt : N - 1;
  const float x = B.x[ptc];
  const float* P = B.params;
  const float* st = B.stash + (long long)ptc * (L + 2) * 3 * H;
  float* G = B.partials + (long long)tile * B.pstride;
  float* To = sh;
  float* Tz = sh + 3 * H * MODT_LD;
  float* Fb = Tz + 3 * H * MODT_LD;       // [3][16][R]: dL/dF of the tile's points
  const bool active = wave < NB;          // this wave owns feature block `wave`
  const int f4 = 16 * wave + 4 * g;       // its lane's first feature
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // dL/dF of the tile into LDS: either the finished sums, or -- B.fpart -- the grid kernel's per-group partials summed here
  // in spinn_fbar_sum_kernel's order (FSUM_PARTS interleaved subsets, then the subsets in order): that launch is saved
  for (int e = tid; e < 16 * 2 * R; e += MODT_BLOCK) {
    const int p = e / (2 * R), col = e - p * 2 * R;
    const int i = pt0 + p;
    float tot = 0.f;
    if (i < N) {
      if (B.fpart != nullptr) {
        const float* q = B.fpart + (long long)i * B.ngrp * 2 * R + col;
        float part[FSUM_PARTS];
#pragma unroll
        for (int k = 0; k < FSUM_PARTS; ++k) {
          float sk = 0.f;
#pragma unroll 4
          for (int gi = k; gi < B.ngrp; gi += FSUM_PARTS) sk += q[(long long)gi * 2 * R];
          part[k] = sk;
        }
#pragma unroll
        for (int k = 0; k < FSUM_PARTS; ++k) tot += part[k];
      } else {
        tot = B.Fbar[((long long)(col < R ? 0 : 2) * N + i) * R + (col < R ? col : col - R)];
      }
    }
    Fb[((col < R ? 0 : 2) * 16 + p) * R + (col < R ? col : col - R)] = tot;
  }
  for (int e = tid; e < 16 * R; e += MODT_BLOCK) {  // the first-derivative stream
    const int p = e / R, r = e - p * R;
    Fb[(16 + p) * R + r] = (B.fpart == nullptr && pt0 + p < N) ? B.Fbar[((long long)N + pt0 + p) * R + r] : 0.f;
  }

  f32x4 U[3], V[3], zu[3], zv[3], du1 = zero4, du2 = zero4, du3 = zero4, dv1 = zero4, dv2 = zero4, dv3 = zero4;
  f32x4 zl[3], al[3], ol[3], d1 = zero4, d2 = zero4, d3 = zero4, ob[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) U[s] = V[s] = zu[s] = zv[s] = zl[s] = al[s] = ol[s] = ob[s] = zero4;
  // every operand of the last_fc stage is requested here, in front of the activation arithmetic: a tile is one workgroup's
  // latency chain (24 workgroups for 3 x 128 points), a load waited for inside a loop costs a memory round trip per trip
  f32x4 awl[4];
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
    awl[rb] = (active && rb < RB) ? *(const f32x4*)&P[a.o.wl + (16 * wave + c) * R + 16 * rb + 4 * g] : zero4;
  if (active) {
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      zu[s] = *(const f32x4*)&st[(0 * 3 + s) * H + f4];
      zv[s] = *(const f32x4*)&st[(1 * 3 + s) * H + f4];
      zl[s] = *(const f32x4*)&st[((2 + L - 1) * 3 + s) * H + f4];
    }
    sp_act_streams4(act, zu, U, du1, du2, du3);
    sp_act_streams4(act, zv, V, dv1, dv2, dv3);
    sp_act_streams4(act, zl, al, d1, d2, d3);
    sp_gate4(al, U, V, ol);
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int r = 0; r < 4; ++r) To[(s * H + f4 + r) * MODT_LD + c] = ol[s][r];
  }
  __syncthreads();
  // ---- last_fc:  gWL[k][r] = sum_{s,p} o_s[k][p] Fbar_s[r][p]   (blocks (kb, rb) over the waves)
  for (int b = wave; b < NB * RB; b += MODT_BLOCK / 64) {
    const int kb = b / RB, rb = b - kb * RB;
    f32x4 acc = zero4;
    float bv[3][4];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        bv[s][t] = Fb[(s * 16 + 4 * t + g) * R + 16 * rb + c];  // (zero beyond the batch)
      }
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(To[(s * H + 16 * kb + c) * MODT_LD + 4 * t + g], bv[s][t], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) G[a.o.wl + (16 * kb + 4 * g + r) * R + 16 * rb + c] = acc[r];
  }
  for (int r = tid; r < R; r += MODT_BLOCK) {  // gbL[r] = sum_p Fbar_0[r][p]
    float sum = 0.f;
    for (int p = 0; p < 16; ++p) sum += Fb[p * R + r];
    G[a.o.bl + r] = sum;
  }
  // ---- obar_s[k][p] = sum_r WL[k][r] Fbar_s[r][p]   (this wave's feature block; k-step (rb, r); operands requested above)
  if (active) {
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      f32x4 acc = zero4;
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
        if (rb < RB) {
          const f32x4 fb = *(const f32x4*)&Fb[(s * 16 + c) * R + 16 * rb + 4 * g];
#pragma unroll
          for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(awl[rb][r], fb[r], acc, 0, 0, 0);
        }
      ob[s] = acc;
    }
  }
  f32x4 Ub[3] = {zero4, zero4, zero4}, Vb[3] = {zero4, zero4, zero4};
  for (int l = L - 1; l >= 0; --l) {
    f32x4 zb[3] = {zero4, zero4, zero4};
    // this layer's operands from memory, requested in front of the pointwise arithmetic that hides them: the A fragments of
    // obar_{l-1} = W_l zbar_l and the stash of layer l - 1
    f32x4 aw[4], zp[3] = {zero4, zero4, zero4};
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
      aw[jb] = (active && l > 0 && jb < NB) ? *(const f32x4*)&P[a.o.w[l] + (16 * wave + c) * H + 16 * jb + 4 * g] : zero4;
    if (active && l > 0) {
#pragma unroll
      for (int s = 0; s < 3; ++s) zp[s] = *(const f32x4*)&st[((2 + l - 1) * 3 + s) * H + f4];
    }
    if (active) {
      // adjoint of the gate  o = V + a (U - V)  and of the activation streams (the per-point kernel's formulas on float4)
      const f32x4 D0 = U[0] - V[0], D1 = U[1] - V[1], D2 = U[2] - V[2];
      const f32x4 ab2 = ob[2] * D0;
      const f32x4 ab1 = ob[1] * D0 + 2.f * ob[2] * D1;
      const f32x4 ab0 = ob[0] * D0 + ob[1] * D1 + ob[2] * D2;
      const f32x4 Db0 = ob[0] * al[0] + ob[1] * al[1] + ob[2] * al[2];
      const f32x4 Db1 = ob[1] * al[0] + 2.f * ob[2] * al[1];
      const f32x4 Db2 = ob[2] * al[0];
      Ub[0] += Db0; Ub[1] += Db1; Ub[2] += Db2;
      Vb[0] += ob[0] - Db0; Vb[1] += ob[1] - Db1; Vb[2] += ob[2] - Db2;
      zb[2] = d1 * ab2;
      zb[1] = d1 * ab1 + 2.f * d2 * zl[1] * ab2;
      zb[0] = d1 * ab0 + d2 * zl[1] * ab1 + (d3 * zl[1] * zl[1] + d2 * zl[2]) * ab2;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float sb = sp_sum16(zb[0][r]);  // bias gradient: the sum over the tile's points
        if (c == 0) G[a.o.b[l] + f4 + r] = sb;
      }
      if (l == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float sw = sp_sum16(x * zb[0][r] + zb[1][r]);  // input streams (x, 1, 0)
          if (c == 0) G[a.o.w[0] + f4 + r] = sw;
        }
      }
    }
    if (l == 0) break;
    f32x4 ap[3] = {zero4, zero4, zero4}, op[3] = {zero4, zero4, zero4};
    f32x4 p1 = zero4, p2 = zero4, p3 = zero4;
    if (active) {
      sp_act_streams4(act, zp, ap, p1, p2, p3);
      sp_gate4(ap, U, V, op);
    }
    __syncthreads();  // the previous readers of the two planes are done
    if (active) {
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          To[(s * H + f4 + r) * MODT_LD + c] = op[s][r];
          Tz[(s * H + f4 + r) * MODT_LD + c] = zb[s][r];
        }
    }
    __syncthreads();
    // gW_l[k][j] = sum_{s,p} o_prev_s[k][p] zbar_s[j][p]   (blocks (kb, jb) over the waves)
    for (int b = wave; b < NB * NB; b += MODT_BLOCK / 64) {
      const int kb = b / NB, jb = b - kb * NB;
      f32x4 acc = zero4;
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(To[(s * H + 16 * kb + c) * MODT_LD + 4 * t + g],
                                                     Tz[(s * H + 16 * jb + c) * MODT_LD + 4 * t + g], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) G[a.o.w[l] + (16 * kb + 4 * g + r) * H + 16 * jb + c] = acc[r];
    }
    // obar_prev_s[k][p] = sum_j W_l[k][j] zbar_s[j][p]   (this wave's block k; zbar of block jb from the plane)
    if (active) {
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        f32x4 acc = zero4;
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
          if (jb < NB) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[jb][r], Tz[(s * H + 16 * jb + 4 * g + r) * MODT_LD + c], acc, 0, 0, 0);
          }
        ob[s] = acc;
      }
#pragma unroll
      for (int s = 0; s < 3; ++s) { zl[s] = zp[s]; al[s] = ap[s]; }
      d1 = p1; d2 = p2; d3 = p3;
    }
  }
  // embeddings: U = act(zu), V = act(zv) with streams (x w + b, w, 0)
  if (active) {
    f32x4 t0 = du1 * Ub[0] + du2 * zu[1] * Ub[1] + (du3 * zu[1] * zu[1] + du2 * zu[2]) * Ub[2];
    f32x4 t1 = du1 * Ub[1] + 2.f * du2 * zu[1] * Ub[2];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float sb = sp_sum16(t0[r]), sw = sp_sum16(x * t0[r] + t1[r]);
      if (c == 0) { G[a.o.bu + f4 + r] = sb; G[a.o.wu + f4 + r] = sw; }
    }
    t0 = dv1 * Vb[0] + dv2 * zv[1] * Vb[1] + (dv3 * zv[1] * zv[1] + dv2 * zv[2]) * Vb[2];
    t1 = dv1 * Vb[1] + 2.f * dv2 * zv[1] * Vb[2];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float sb = sp_sum16(t0[r]), sw = sp_sum16(x * t0[r] + t1[r]);
      if (c == 0) { G[a.o.bv + f4 + r] = sb; G[a.o.wv + f4 + r] = sw; }
    }
  }
}

// The forward sweep in the same tile model: z_l = W_l^T o_{l-1} + b_l with the previous layer's gated output read back from
// one LDS plane as the B operand (k-step (kb, r): feature 16kb + 4g + r of point c), A = W_l[16kb + 4g + r][16w + c] from L2.
__global__ void __launch_bounds__(MODT_BLOCK) modmlp_fwd_tile_kernel(ModArgs a) {
  PPSCI_DYN_SMEM(sh);  // To[3][H][MODT_LD]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int H = a.d.width, L = a.d.n_hidden, R = a.d.d_out, act = a.d.activation, NB = H / 16, RB = R / 16;
  int tile = blockIdx.x, bi = 0;
  while (bi + 1 < a.nbatch && tile >= (a.br[bi].N + 15) / 16) { tile -= (a.br[bi].N + 15) / 16; ++bi; }
  const ModBranch& B = a.br[bi];
  const int N = B.N, pt = tile * 16 + c;
  const bool valid = pt < N;
  const float x = B.x[valid ? pt : N - 1];
  const float* P = B.params;
  float* st = (B.stash && valid) ? B.stash + (long long)pt * (L + 2) * 3 * H : nullptr;
  float* To = sh;
  const bool active = wave < NB;
  const int f4 = 16 * wave + 4 * g;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 U[3] = {zero4, zero4, zero4}, V[3] = {zero4, zero4, zero4}, o[3] = {zero4, zero4, zero4};
  f32x4 e1, e2, e3;
  if (active) {
    f32x4 z[3];
    const f32x4 wu = *(const f32x4*)&P[a.o.wu + f4], bu = *(const f32x4*)&P[a.o.bu + f4];
    z[0] = x * wu + bu; z[1] = wu; z[2] = zero4;
    if (st) { *(f32x4*)&st[0 * H + f4] = z[0]; *(f32x4*)&st[1 * H + f4] = z[1]; *(f32x4*)&st[2 * H + f4] = z[2]; }
    sp_act_streams4(act, z, U, e1, e2, e3);
    const f32x4 wv = *(const f32x4*)&P[a.o.wv + f4], bv = *(const f32x4*)&P[a.o.bv + f4];
    z[0] = x * wv + bv; z[1] = wv; z[2] = zero4;
    if (st) { *(f32x4*)&st[3 * H + f4] = z[0]; *(f32x4*)&st[4 * H + f4] = z[1]; *(f32x4*)&st[5 * H + f4] = z[2]; }
    sp_act_streams4(act, z, V, e1, e2, e3);
  }
  for (int l = 0; l < L; ++l) {
    f32x4 z[3] = {zero4, zero4, zero4};
    if (l == 0) {
      if (active) {
        const f32x4 w0 = *(const f32x4*)&P[a.o.w[0] + f4], b0 = *(const f32x4*)&P[a.o.b[0] + f4];
        z[0] = x * w0 + b0; z[1] = w0;
      }
    } else {
      // (the layer's weights are requested in front of the two barriers that hide their latency)
      float aw[4][4];
      f32x4 bl4 = zero4;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          aw[kb][r] = (active && kb < NB) ? P[a.o.w[l] + (16 * kb + 4 * g + r) * H + 16 * wave + c] : 0.f;
      if (active) bl4 = *(const f32x4*)&P[a.o.b[l] + f4];
      __syncthreads();  // the previous layer's readers are done
      if (active) {
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
          for (int r = 0; r < 4; ++r) To[(s * H + f4 + r) * MODT_LD + c] = o[s][r];
      }
      __syncthreads();
      if (active) {
        z[0] = bl4;  // bias: every point (column) of the block gets b[feature]
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
          if (kb < NB) {
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                z[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[kb][r], To[(s * H + 16 * kb + 4 * g + r) * MODT_LD + c], z[s], 0, 0, 0);
          }
      }
    }
    if (active) {
      if (st) {
        float* q = st + (2 + l) * 3 * H;
        *(f32x4*)&q[f4] = z[0]; *(f32x4*)&q[H + f4] = z[1]; *(f32x4*)&q[2 * H + f4] = z[2];
      }
      f32x4 av[3];
      sp_act_streams4(act, z, av, e1, e2, e3);
      sp_gate4(av, U, V, o);
    }
  }
  __syncthreads();
  if (active) {
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int r = 0; r < 4; ++r) To[(s * H + f4 + r) * MODT_LD + c] = o[s][r];
  }
  __syncthreads();
  // last_fc: F_s[r][p] = sum_k WL[k][r] o_s[k][p] (+ bl for the value stream): rank blocks over the waves
  for (int rb = wave; rb < RB; rb += MODT_BLOCK / 64) {
    f32x4 acc[3] = {*(const f32x4*)&P[a.o.bl + 16 * rb + 4 * g], zero4, zero4};
    float aw[4][4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) aw[kb][r] = kb < NB ? P[a.o.wl + (16 * kb + 4 * g + r) * R + 16 * rb + c] : 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
      if (kb < NB) {
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[kb][r], To[(s * H + 16 * kb + 4 * g + r) * MODT_LD + c], acc[s], 0, 0, 0);
      }
    if (valid) {
#pragma unroll
      for (int s = 0; s < 3; ++s) *(f32x4*)&B.F[((long long)s * N + pt) * R + 16 * rb + 4 * g] = acc[s];
    }
  }
}

// ------------------------------------------------------------------------------------ grid kernels
struct GridArgs {
  ppsci_spinn_grid_desc d;
  const float* F[3];     // per axis [3][n_a][R]
  const float* label;    // [nx*ny*nz] or null
  float* resid;          // optional [nx*ny*nz]
  float* gadj;           // [nx*ny*nz] adjoint of the residual (train) or null
  float* loss_partials;  // [gridDim.x]
  float* Fbar;           // bwd: [3][n_a][R] of `axis`
  float* Fpart;          // bwd scratch: [n_a][groups][2][R]
  int axis;
  int iters;
  int ngrp;  // fbar_sum: partial groups per index
  // axis < 0: all three axes in one launch -- workgroups [0, wg3[0]) axis 0, the next wg3[1] axis 1, ...
  float* Fbar3[3];
  long long poff3[3];  // offset of the axis' partial rows in Fpart
  int ngrp3[3];
  int wg3[3];
};

__device__ __forceinline__ int grid_locate(const GridArgs& a, int& bx, int& ngrp, float*& Fbar, float*& Fpart) {
  int ax = a.axis;
  ngrp = a.ngrp; Fbar = a.Fbar; Fpart = a.Fpart;
  if (ax < 0) {
    ax = 0;
    while (ax < 2 && bx >= a.wg3[ax]) { bx -= a.wg3[ax]; ++ax; }
    ngrp = a.ngrp3[ax]; Fbar = a.Fbar3[ax]; Fpart = a.Fpart + a.poff3[ax];
  }
  return ax;
}

#define GRID_BLOCK 256
#define GRID_JT 8  // j rows per forward workgroup

// Forward: one workgroup per (i, tile of GRID_JT j's); thread = k.  Per (j, r) the x/y factors are folded into
//   p_r = (cu*fx + cxx*fx'')*fy + cyy*fx*fy''   and   q_r = czz*fx*fy     (LDS, broadcast reads)
// so that  res(i,j,k) = sum_r p_r*fz[k,r] + q_r*fz''[k,r]:  two FMAs per rank and point, fz / fz'' read from an
// LDS copy transposed to [r][k] (conflict-free, consecutive lanes = consecutive k).
__global__ void __launch_bounds__(GRID_BLOCK) spinn_grid_fwd_kernel(GridArgs a) {
  PPSCI_DYN_SMEM(sm);
  const int R = a.d.rank, nx = a.d.n[0], ny = a.d.n[1], nz = a.d.n[2];
  const int njt = (ny + GRID_JT - 1) / GRID_JT;
  const int i = blockIdx.x / njt, j0 = (blockIdx.x % njt) * GRID_JT;
  const int KC = a.iters;  // k-chunk held in LDS
  float* zT0 = sm;                   // [R][KC]
  float* zT2 = zT0 + R * KC;         // [R][KC]
  float* pq = zT2 + R * KC;          // [GRID_JT][2][R]
  float* red = pq + GRID_JT * 2 * R; // [GRID_BLOCK]
  const long long sx = (long long)nx * R, sy = (long long)ny * R, sz = (long long)nz * R;
  (void)sx;
  for (int t = threadIdx.x; t < GRID_JT * R; t += GRID_BLOCK) {
    const int jj = t / R, r = t % R, j = j0 + jj;
    float p = 0.f, q = 0.f;
    if (j < ny) {
      const float x0 = a.F[0][(long long)i * R + r], x2 = a.F[0][2 * (long long)nx * R + (long long)i * R + r];
      const float y0 = a.F[1][(long long)j * R + r], y2 = a.F[1][2 * sy + (long long)j * R + r];
      p = (a.d.cu * x0 + a.d.cxx * x2) * y0 + a.d.cyy * x0 * y2;
      q = a.d.czz * x0 * y0;
    }
    pq[(jj * 2 + 0) * R + r] = p;
    pq[(jj * 2 + 1) * R + r] = q;
  }
  float lsum = 0.f;
  for (int k0 = 0; k0 < nz; k0 += KC) {
    __syncthreads();
    for (int t = threadIdx.x; t < KC * R; t += GRID_BLOCK) {  // coalesced global read, transposed LDS write
      const int kk = t / R, r = t % R, k = k0 + kk;
      zT0[r * KC + kk] = k < nz ? a.F[2][(long long)k * R + r] : 0.f;
      zT2[r * KC + kk] = k < nz ? a.F[2][2 * sz + (long long)k * R + r] : 0.f;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < GRID_JT * KC; t += GRID_BLOCK) {
      const int jj = t / KC, kk = t % KC, j = j0 + jj, k = k0 + kk;
      if (j < ny && k < nz) {
        float res = 0.f;
        const float* pj = pq + jj * 2 * R;
        for (int r = 0; r < R; ++r) res += pj[r] * zT0[r * KC + kk] + pj[R + r] * zT2[r * KC + kk];
        const long long pidx = ((long long)i * ny + j) * nz + k;
        if (a.resid) a.resid[pidx] = res;
        const float diff = res - (a.label ? a.label[pidx] : 0.f);
        lsum += a.d.scale * diff * diff;
        if (a.gadj) a.gadj[pidx] = 2.f * a.d.scale * diff;
      }
    }
  }
  __syncthreads();
  red[threadIdx.x] = lsum;
  __syncthreads();
  for (int s = GRID_BLOCK / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) a.loss_partials[blockIdx.x] = red[0];
}

// Reverse, axis `ax` with the two other axes (b, c), c the faster-varying one.  One workgroup per (index i,
// group of GRID_JG rows jb); thread = (rank r, row slot).  fc / fc'' ([nc][R]) and the group's adjoint rows
// g(i, jb, :) are staged in LDS; for each row the inner sum over kc
//     t0[r] = sum_kc g(i,jb,kc) fc[kc,r],   t2[r] = sum_kc g(i,jb,kc) fc''[kc,r]
// is two FMAs per element, then
//     value  += fb[jb,r]*(cu*t0 + coef_c*t2) + fb''[jb,r]*coef_b*t0,     second += fb[jb,r]*t0
// Row slots are summed through LDS in a fixed order; the per-group partials go to `Fpart` and are summed by
// spinn_fbar_sum_kernel (fixed order as well).
#define GRID_JG 16
__global__ void __launch_bounds__(GRID_BLOCK) spinn_grid_bwd_kernel(GridArgs a) {
  PPSCI_DYN_SMEM(sm);
  const int R = a.d.rank, ax = a.axis;
  const int b = ax == 0 ? 1 : 0, c = ax == 2 ? 1 : 2;
  const int nb = a.d.n[b], nc = a.d.n[c];
  const int ngrp = (nb + GRID_JG - 1) / GRID_JG;
  const int i = blockIdx.x / ngrp, grp = blockIdx.x % ngrp, jb0 = grp * GRID_JG;
  const float coef[3] = {a.d.cxx, a.d.cyy, a.d.czz};
  long long stride[3];
  stride[2] = 1; stride[1] = a.d.n[2]; stride[0] = (long long)a.d.n[1] * a.d.n[2];
  const long long sb = (long long)nb * R, sc = (long long)nc * R;
  const int slots = GRID_BLOCK / R > 0 ? GRID_BLOCK / R : 1;
  float* f0 = sm;                     // [nc][R]
  float* f2 = f0 + (long long)nc * R; // [nc][R]
  float* gs = f2 + (long long)nc * R; // [GRID_JG][nc]
  float* red = gs + GRID_JG * nc;     // [2][slots][R]
  for (int t = threadIdx.x; t < nc * R; t += GRID_BLOCK) {
    f0[t] = a.F[c][t];
    f2[t] = a.F[c][2 * sc + t];
  }
  for (int t = threadIdx.x; t < GRID_JG * nc; t += GRID_BLOCK) {
    const int jj = t / nc, kc = t % nc, jb = jb0 + jj;
    gs[t] = jb < nb ? a.gadj[i * stride[ax] + jb * stride[b] + kc * stride[c]] : 0.f;
  }
  __syncthreads();
  const int r = threadIdx.x % R, slot = threadIdx.x / R;
  float v0 = 0.f, v2 = 0.f;
  if (slot < slots) {
    for (int jj = slot; jj < GRID_JG && jb0 + jj < nb; jj += slots) {
      const int jb = jb0 + jj;
      const float* g = gs + jj * nc;
      float t0 = 0.f, t2 = 0.f;
      for (int kc = 0; kc < nc; ++kc) {
        const float gv = g[kc];
        t0 += gv * f0[kc * R + r];
        t2 += gv * f2[kc * R + r];
      }
      const float y0 = a.F[b][(long long)jb * R + r], y2 = a.F[b][2 * sb + (long long)jb * R + r];
      v0 += y0 * (a.d.cu * t0 + coef[c] * t2) + y2 * coef[b] * t0;
      v2 += y0 * t0;
    }
    red[slot * R + r] = v0;
    red[(slots + slot) * R + r] = v2;
  }
  __syncthreads();
  if (threadIdx.x < (unsigned)R) {
    float s0 = 0.f, s2 = 0.f;
    for (int q = 0; q < slots; ++q) {
      s0 += red[q * R + r];
      s2 += red[(slots + q) * R + r];
    }
    float* out = a.Fpart + ((long long)i * ngrp + grp) * 2 * R;
    out[r] = s0;
    out[R + r] = coef[ax] * s2;
  }
}

// ---- MFMA forms of the two grid contractions (rank <= 64, rank % 4 == 0); the scalar kernels above stay for the rest.
//
// v_mfma_f32_16x16x4_f32: lane (g = l>>4, c = l&15) supplies A[row c][k g], B[k g][col c] and holds D[row 4g+rr][col c].
// The order in which the contraction index is fed to the k-steps is free, and both kernels use it to make every
// operand load a 16-byte one: k-step (q, e) of a 16-wide slab q takes index 16q + 4g + e from lane group g, i.e. each
// lane loads ONE float4 per slab and operand (4 lane groups = 64 contiguous bytes of one row).
//
// Forward as a GEMM per i:  res[(j), k] = sum_{r'} A[j][r'] B[r'][k],  r' over the 2R columns (p_r | q_r):
// one wave = (i, 16 rows j, every GRID_CS-th 16-column tile of k); the wave computes its A operand (16 x 2R) once
// into registers; B[r'][k] = fz[k,r] / fz''[k,r] from the (L1/L2-resident) factor table.  Consecutive lanes c =
// consecutive k: label read and adjoint write in 64 B runs.  The kernel is latency-bound (8 MB of labels, a few
// hundred MFMAs per wave): the label loads are issued with the B-operand loads, ahead of the MFMA chain, and the
// column split puts 4 waves on every SIMD.
#define GRID_CS 4
template <int NQ>
__global__ void __launch_bounds__(GRID_BLOCK) spinn_grid_fwd_mfma_kernel(GridArgs a) {
  const int R = a.d.rank, nx = a.d.n[0], ny = a.d.n[1], nz = a.d.n[2];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const int njb = (ny + 15) / 16;
  const int wid = blockIdx.x * (GRID_BLOCK / 64) + wave;  // ((i, j block), column phase)
  if (wid >= nx * njb * GRID_CS) return;
  const int cs = wid % GRID_CS, ij = wid / GRID_CS;
  const int i = ij / njb, j0 = (ij % njb) * 16;
  const long long sy = (long long)ny * R, sz = (long long)nz * R;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 Ap[NQ], Aq[NQ];
  {
    const int j = j0 + c;
    f32x4 x0[NQ], x2[NQ], y0[NQ], y2[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int r4 = 16 * q + 4 * g;
      const bool ok = j < ny && r4 < R;
      x0[q] = ok ? *(const f32x4*)&a.F[0][(long long)i * R + r4] : zero4;
      x2[q] = ok ? *(const f32x4*)&a.F[0][2 * (long long)nx * R + (long long)i * R + r4] : zero4;
      y0[q] = ok ? *(const f32x4*)&a.F[1][(long long)j * R + r4] : zero4;
      y2[q] = ok ? *(const f32x4*)&a.F[1][2 * sy + (long long)j * R + r4] : zero4;
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      Ap[q] = (a.d.cu * x0[q] + a.d.cxx * x2[q]) * y0[q] + a.d.cyy * x0[q] * y2[q];
      Aq[q] = a.d.czz * x0[q] * y0[q];
    }
  }
  float lsum = 0.f;
  for (int k0 = 16 * cs; k0 < nz; k0 += 16 * GRID_CS) {
    const int k = k0 + c;
    f32x4 B0[NQ], B2[NQ];
    float lab[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int j = j0 + 4 * g + rr;
      lab[rr] = (a.label && j < ny && k < nz) ? a.label[((long long)i * ny + j) * nz + k] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int r4 = 16 * q + 4 * g;
      const bool ok = k < nz && r4 < R;
      B0[q] = ok ? *(const f32x4*)&a.F[2][(long long)k * R + r4] : zero4;
      B2[q] = ok ? *(const f32x4*)&a.F[2][2 * sz + (long long)k * R + r4] : zero4;
    }
    f32x4 acc = zero4;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(Ap[q][e], B0[q][e], acc, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(Aq[q][e], B2[q][e], acc, 0, 0, 0);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int j = j0 + 4 * g + rr;
      if (j < ny && k < nz) {
        const long long pidx = ((long long)i * ny + j) * nz + k;
        const float res = acc[rr];
        if (a.resid) a.resid[pidx] = res;
        const float diff = res - lab[rr];
        lsum += a.d.scale * diff * diff;
        if (a.gadj) a.gadj[pidx] = 2.f * a.d.scale * diff;
      }
    }
  }
  lsum += __shfl_xor(lsum, 1, 64);
  lsum += __shfl_xor(lsum, 2, 64);
  lsum += __shfl_xor(lsum, 4, 64);
  lsum += __shfl_xor(lsum, 8, 64);
  lsum += __shfl_xor(lsum, 16, 64);
  lsum += __shfl_xor(lsum, 32, 64);
  if (lane == 0) a.loss_partials[wid] = lsum;
}

// Reverse for axis `ax`, other axes (b, c), K = the c axis:  one wave = (16 consecutive indices i on `ax`, a chunk of
// GRID_JC indices jb on b); one workgroup = 4 such chunks of the same i block.  Per jb a GEMM
//     T[i, n] = sum_kc g(i,jb,kc) * [fc | fc''][kc, n]     (M = 16 i, N = 2R)
// then -- elementwise in the D layout, no cross-lane traffic --
//     value  += fb[jb,r]*(cu*t0 + coef_c*t2) + fb''[jb,r]*coef_b*t0,     second += fb[jb,r]*t0
// accumulated over the chunk in registers.  fc / fc'' are staged once per workgroup in LDS with a row stride = 4
// mod 32 floats (B operand reads of the (q, e) k-order conflict-free).  The adjoint rows are read as float4 along kc
// where kc is the contiguous axis (ax = 0, 1; nz % 4 == 0) and as 64 B runs along i otherwise (ax = 2); the loads of a
// whole 128-wide slab are issued together, ahead of the MFMA chain.  The 4 waves' sums are added through LDS in
// wave order; the workgroup partials -> `Fpart`, summed by spinn_fbar_sum_kernel.
#define GRID_JC 1
#define GRID_QSLAB 8  // 16-wide slabs of kc per trip
template <int NT>
__global__ void __launch_bounds__(GRID_BLOCK) spinn_grid_bwd_mfma_kernel(GridArgs a) {
  PPSCI_DYN_SMEM(sm);
  int bx = blockIdx.x, ngrp;  // ngrp: workgroup partials per index
  float *Fbar_, *Fpart;
  const int R = a.d.rank, ax = grid_locate(a, bx, ngrp, Fbar_, Fpart);
  (void)Fbar_;
  const int b = ax == 0 ? 1 : 0, c = ax == 2 ? 1 : 2;
  const int na = a.d.n[ax], nb = a.d.n[b], nc = a.d.n[c];
  const int RP = a.iters;   // padded LDS row stride
  const int ncp = (nc + 15) & ~15;
  const float coef[3] = {a.d.cxx, a.d.cyy, a.d.czz};
  long long stride[3];
  stride[2] = 1; stride[1] = a.d.n[2]; stride[0] = (long long)a.d.n[1] * a.d.n[2];
  const long long sb = (long long)nb * R, sc = (long long)nc * R;
  const bool vec = stride[c] == 1 && (a.d.n[2] & 3) == 0;
  float* f0 = sm;             // [ncp][RP]
  float* f2 = f0 + ncp * RP;  // [ncp][RP]
  float* ex = f2 + ncp * RP;  // [4 waves][2][NT][4][64] exchange
  for (int t = threadIdx.x; t < ncp * RP; t += GRID_BLOCK) {
    const int kc = t / RP, r = t % RP;
    const bool ok = kc < nc && r < R;
    f0[t] = ok ? a.F[c][(long long)kc * R + r] : 0.f;
    f2[t] = ok ? a.F[c][2 * sc + (long long)kc * R + r] : 0.f;
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, cl = lane & 15;
  const int i0 = (bx / ngrp) * 16, grp = bx % ngrp;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 v0[NT], v2[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    v0[nt] = zero4;
    v2[nt] = zero4;
  }
  const int irow = i0 + cl;
  for (int jj = 0; jj < GRID_JC; ++jj) {
    const int jb = (grp * (GRID_BLOCK / 64) + wave) * GRID_JC + jj;
    if (jb >= nb) break;
    f32x4 t0[NT], t2[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      t0[nt] = zero4;
      t2[nt] = zero4;
    }
    const float* gp = a.gadj + (long long)irow * stride[ax] + (long long)jb * stride[b];
    for (int k0 = 0; k0 < ncp; k0 += 16 * GRID_QSLAB) {
      f32x4 av[GRID_QSLAB];
#pragma unroll
      for (int q = 0; q < GRID_QSLAB; ++q) {
        const int kc = k0 + 16 * q + 4 * g;
        if (vec) {
          av[q] = (irow < na && kc < nc) ? *(const f32x4*)&gp[kc] : zero4;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) av[q][e] = (irow < na && kc + e < nc) ? gp[(long long)(kc + e) * stride[c]] : 0.f;
        }
      }
#pragma unroll
      for (int q = 0; q < GRID_QSLAB; ++q) {
        if (k0 + 16 * q < ncp) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int kr = k0 + 16 * q + 4 * g + e;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              t0[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q][e], f0[kr * RP + 16 * nt + cl], t0[nt], 0, 0, 0);
              t2[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q][e], f2[kr * RP + 16 * nt + cl], t2[nt], 0, 0, 0);
            }
          }
        }
      }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int r = 16 * nt + cl;
      const float y0 = r < R ? a.F[b][(long long)jb * R + r] : 0.f;
      const float y2 = r < R ? a.F[b][2 * sb + (long long)jb * R + r] : 0.f;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        v0[nt][rr] += y0 * (a.d.cu * t0[nt][rr] + coef[c] * t2[nt][rr]) + y2 * coef[b] * t0[nt][rr];
        v2[nt][rr] += y0 * t0[nt][rr];
      }
    }
  }
  // the 4 waves' sums, added in wave order
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      ex[((wave * 2 + 0) * NT * 4 + nt * 4 + rr) * 64 + lane] = v0[nt][rr];
      ex[((wave * 2 + 1) * NT * 4 + nt * 4 + rr) * 64 + lane] = v2[nt][rr];
    }
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * NT * 4 * 64; e += GRID_BLOCK) {
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < GRID_BLOCK / 64; ++w) sum += ex[w * 2 * NT * 4 * 64 + e];
    const int l = e & 63, q = e >> 6, rr = q & 3, nt = (q >> 2) % NT, which = q / (4 * NT);
    const int i = i0 + 4 * (l >> 4) + rr, r = 16 * nt + (l & 15);
    if (i < na && r < R) {
      float* out = Fpart + ((long long)i * ngrp + grp) * 2 * R;
      if (which == 0) out[r] = sum;
      else out[R + r] = coef[ax] * sum;
    }
  }
}

// One workgroup per index i: thread = (column of the [2][R] partial row, one of FSUM_PARTS interleaved group
// subsets); the subsets' sums are combined through LDS in a fixed order (the loads of one thread are independent:
// one round of memory latency instead of one per group).
__global__ void __launch_bounds__(GRID_BLOCK) spinn_fbar_sum_kernel(GridArgs a) {
  PPSCI_DYN_SMEM(red);  // [FSUM_PARTS][2R]
  int bx = blockIdx.x, ngrp;
  float *Fbar, *Fpart;
  const int R = a.d.rank, ax = grid_locate(a, bx, ngrp, Fbar, Fpart), na = a.d.n[ax];
  const int i = bx, C2 = 2 * R;
  for (int e = threadIdx.x; e < C2 * FSUM_PARTS; e += GRID_BLOCK) {
    const int part = e / C2, col = e % C2;
    const float* p = Fpart + (long long)i * ngrp * C2 + col;
    float s = 0.f;
#pragma unroll 4
    for (int g = part; g < ngrp; g += FSUM_PARTS) s += p[(long long)g * C2];
    red[e] = s;
  }
  __syncthreads();
  const long long sa = (long long)na * R;
  for (int col = threadIdx.x; col < C2; col += GRID_BLOCK) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < FSUM_PARTS; ++q) s += red[q * C2 + col];
    const long long t = (long long)i * R + (col % R);
    if (col < R) {
      Fbar[t] = s;
      Fbar[sa + t] = 0.f;  // first-derivative stream is not used by these residuals
    } else {
      Fbar[2 * sa + t] = s;
    }
  }
}

// ------------------------------------------------------------------------------------ C ABI
static int mod_check(const ppsci_modmlp_desc* d) {
  if (!d || d->width < 1 || d->width > SP_MAXH || d->n_hidden < 1 || d->n_hidden > PPSCI_MAX_HIDDEN || d->d_out < 1 ||
      d->d_out > SP_MAXR * 4) {
    ppsci_set_error("modmlp: invalid descriptor (width <= %d)", SP_MAXH);
    return PPSCI_E_INVALID;
  }
  return PPSCI_OK;
}

extern "C" int64_t ppsci_modmlp_param_count(const ppsci_modmlp_desc* d) {
  if (mod_check(d) != PPSCI_OK) return -1;
  ModOff o;
  mod_offsets(*d, o);
  return o.P;
}

extern "C" int64_t ppsci_modmlp_stash_floats(const ppsci_modmlp_desc* d, int64_t n) {
  if (mod_check(d) != PPSCI_OK) return 0;
  return n * (d->n_hidden + 2) * 3 * d->width;
}

static int block_for(int H) { return ((H + 63) / 64) * 64; }

// the reverse sweep by 16-point tiles (modmlp_bwd_tile_kernel): width and rank multiples of 16, at most 64
// Measured on MI355X (Helmholtz3D, 3 x 128 points, 4 x 64 nets, rank 32; profiles/r06_spinn_*): reverse 30.6 us by tiles against
// 34.2 us by points, forward 17.7 us by tiles against 12.7 us by points -- 24 workgroups of four waves are a longer latency
// chain per layer than 384 of them, and only the reverse sweep has traffic to save (26 MB of per-point gradient rows).
// Default 1: tiles in the reverse sweep only; 2: both sweeps; 0: neither.
static int g_mod_tile = 1;
extern "C" void ppsci_set_modmlp_tile(int mode) { g_mod_tile = mode < 0 ? 0 : (mode > 2 ? 2 : mode); }
static bool mod_tile_shape(const ppsci_modmlp_desc* d) {
  return d->width % 16 == 0 && d->width <= 64 && d->d_out % 16 == 0 && d->d_out <= 64;
}
static bool mod_tiled(const ppsci_modmlp_desc* d) { return g_mod_tile >= 1 && mod_tile_shape(d); }      // reverse sweep
static bool mod_tiled_fwd(const ppsci_modmlp_desc* d) { return g_mod_tile >= 2 && mod_tile_shape(d); }  // forward sweep

extern "C" int64_t ppsci_modmlp_bwd_rows(const ppsci_modmlp_desc* d, int64_t n) {
  if (mod_check(d) != PPSCI_OK || n < 1) return 0;
  return mod_tiled(d) ? (n + 15) / 16 : n;
}

extern "C" int ppsci_modmlp_fwd_batch(const ppsci_modmlp_desc* d, int nbatch, const float* const* params, const int64_t* n,
                                      const float* const* x, float* const* F, float* const* stash, void* stream) {
  if (mod_check(d) != PPSCI_OK || nbatch < 1 || nbatch > SP_MAXBATCH || !params || !n || !x || !F) {
    ppsci_set_error("modmlp_fwd: invalid argument");
    return PPSCI_E_INVALID;
  }
  ModArgs a;
  memset(&a, 0, sizeof(a));
  a.d = *d;
  mod_offsets(*d, a.o);
  a.nbatch = nbatch;
  long long total = 0;
  for (int b = 0; b < nbatch; ++b) {
    if (!params[b] || !x[b] || !F[b] || n[b] < 1) {
      ppsci_set_error("modmlp_fwd: invalid argument (branch %d)", b);
      return PPSCI_E_INVALID;
    }
    a.br[b].params = params[b]; a.br[b].x = x[b]; a.br[b].F = F[b]; a.br[b].stash = stash ? stash[b] : nullptr;
    a.br[b].N = (int)n[b];
    total += n[b];
  }
  if (mod_tiled_fwd(d)) {
    int tiles = 0;
    for (int b = 0; b < nbatch; ++b) tiles += (int)((n[b] + 15) / 16);
    PPSCI_LAUNCH(modmlp_fwd_tile_kernel, ModArgs, tiles, MODT_BLOCK, (size_t)3 * d->width * MODT_LD * sizeof(float), stream, a);
  } else {
    PPSCI_LAUNCH(modmlp_fwd_kernel, ModArgs, (int)total, block_for(d->width), 3 * d->width * sizeof(float), stream, a);
  }
  int e = PPSCI_LAST_LAUNCH_ERROR();
  if (e != 0) { ppsci_set_error("modmlp_fwd: launch failed (%d)", e); return PPSCI_E_LAUNCH; }
  return PPSCI_OK;
}

extern "C" int ppsci_modmlp_fwd(const ppsci_modmlp_desc* d, const float* params, int64_t n, const float* x, float* F,
                                float* stash, void* stream) {
  return ppsci_modmlp_fwd_batch(d, 1, &params, &n, &x, &F, stash ? &stash : nullptr, stream);
}

static int modmlp_bwd_batch_impl(const ppsci_modmlp_desc* d, int nbatch, const float* const* params, const int64_t* n,
                                 const float* const* x, const float* const* Fbar, const float* const* fpart, const int* ngrp,
                                 const float* const* stash, float* const* grad_partials, int64_t partial_stride, void* stream);

extern "C" int ppsci_modmlp_bwd_batch(const ppsci_modmlp_desc* d, int nbatch, const float* const* params, const int64_t* n,
                                      const float* const* x, const float* const* Fbar, const float* const* stash,
                                      float* const* grad_partials, int64_t partial_stride, void* stream) {
  return modmlp_bwd_batch_impl(d, nbatch, params, n, x, Fbar, nullptr, nullptr, stash, grad_partials, partial_stride, stream);
}

static int modmlp_bwd_batch_impl(const ppsci_modmlp_desc* d, int nbatch, const float* const* params, const int64_t* n,
                                 const float* const* x, const float* const* Fbar, const float* const* fpart, const int* ngrp,
                                 const float* const* stash, float* const* grad_partials, int64_t partial_stride, void* stream) {
  if (mod_check(d) != PPSCI_OK || nbatch < 1 || nbatch > SP_MAXBATCH || !params || !n || !x || (!Fbar && !fpart) || !stash ||
      !grad_partials || (fpart && (!ngrp || !mod_tiled(d)))) {
    ppsci_set_error("modmlp_bwd: invalid argument");
    return PPSCI_E_INVALID;
  }
  ModArgs a;
  memset(&a, 0, sizeof(a));
  a.d = *d;
  mod_offsets(*d, a.o);
  a.nbatch = nbatch;
  long long total = 0;
  for (int b = 0; b < nbatch; ++b) {
    if (!params[b] || !x[b] || (fpart ? !fpart[b] || ngrp[b] < 1 : !Fbar[b]) || !stash[b] || !grad_partials[b] || n[b] < 1) {
      ppsci_set_error("modmlp_bwd: invalid argument (branch %d)", b);
      return PPSCI_E_INVALID;
    }
    a.br[b].params = params[b]; a.br[b].x = x[b]; a.br[b].Fbar = fpart ? nullptr : Fbar[b]; a.br[b].stash = (float*)stash[b];
    a.br[b].fpart = fpart ? fpart[b] : nullptr; a.br[b].ngrp = fpart ? ngrp[b] : 0;
    a.br[b].partials = grad_partials[b]; a.br[b].N = (int)n[b];
    a.br[b].pstride = partial_stride > 0 ? partial_stride : a.o.P;
    total += n[b];
  }
  if (mod_tiled(d)) {  // one workgroup per 16-point tile (MFMA), one gradient row per tile
    int tiles = 0;
    for (int b = 0; b < nbatch; ++b) tiles += (int)((n[b] + 15) / 16);
    PPSCI_LAUNCH(modmlp_bwd_tile_kernel, ModArgs, tiles, MODT_BLOCK,
                 ((size_t)6 * d->width * MODT_LD + (size_t)48 * d->d_out) * sizeof(float), stream, a);
    int et = PPSCI_LAST_LAUNCH_ERROR();
    if (et != 0) { ppsci_set_error("modmlp_bwd: launch failed (%d)", et); return PPSCI_E_LAUNCH; }
    return PPSCI_OK;
  }
  const int wcols = (d->width > d->d_out ? d->width : d->d_out) + 1;
  PPSCI_LAUNCH(modmlp_bwd_kernel, ModArgs, (int)total, MOD_BWD_BLOCK,
               (6 * d->width + 3 * d->d_out + (size_t)d->width * wcols) * sizeof(float), stream, a);
  int e = PPSCI_LAST_LAUNCH_ERROR();
  if (e != 0) { ppsci_set_error("modmlp_bwd: launch failed (%d)", e); return PPSCI_E_LAUNCH; }
  return PPSCI_OK;
}

extern "C" int ppsci_modmlp_bwd(const ppsci_modmlp_desc* d, const float* params, int64_t n, const float* x,
                                const float* Fbar, const float* stash, float* grad_partials, void* stream) {
  return ppsci_modmlp_bwd_batch(d, 1, &params, &n, &x, &Fbar, &stash, &grad_partials, 0, stream);
}

#define GRID_MFMA_MAX_RANK 64
static bool grid_mfma(const ppsci_spinn_grid_desc* d) { return d->rank <= GRID_MFMA_MAX_RANK && d->rank % 4 == 0; }

static int grid_fwd_blocks(const ppsci_spinn_grid_desc* d) {  // = loss partial rows
  if (grid_mfma(d)) return d->n[0] * ((d->n[1] + 15) / 16) * GRID_CS;
  return d->n[0] * ((d->n[1] + GRID_JT - 1) / GRID_JT);
}

static int grid_bwd_groups(const ppsci_spinn_grid_desc* d, int ax) {
  const int b = ax == 0 ? 1 : 0;
  const int per_wg = GRID_JC * (GRID_BLOCK / 64);
  return grid_mfma(d) ? (d->n[b] + per_wg - 1) / per_wg : (d->n[b] + GRID_JG - 1) / GRID_JG;
}

static int grid_fwd_kc(const ppsci_spinn_grid_desc* d) {
  int kc = 8192 / d->rank;  // 2*R*KC floats <= 64 KiB
  if (kc > 256) kc = 256;
  if (kc > d->n[2]) kc = d->n[2];
  return kc < 1 ? 1 : kc;
}

extern "C" int64_t ppsci_spinn_grid_partial_rows(const ppsci_spinn_grid_desc* d) {
  if (!d || d->rank < 1) return 0;
  return grid_fwd_blocks(d);
}

extern "C" int ppsci_spinn_grid_fwd(const ppsci_spinn_grid_desc* d, const float* Fx, const float* Fy, const float* Fz,
                                    const float* label, float* resid, float* gadj, float* loss_partials, void* stream) {
  if (!d || !Fx || !Fy || !Fz || !loss_partials || d->rank < 1 || d->rank > GRID_BLOCK || d->n[0] < 1 || d->n[1] < 1 ||
      d->n[2] < 1) {
    ppsci_set_error("spinn_grid_fwd: invalid argument");
    return PPSCI_E_INVALID;
  }
  GridArgs a;
  memset(&a, 0, sizeof(a));
  a.d = *d;
  a.F[0] = Fx; a.F[1] = Fy; a.F[2] = Fz;
  a.label = label; a.resid = resid; a.gadj = gadj; a.loss_partials = loss_partials;
  if (grid_mfma(d)) {
    const int waves = grid_fwd_blocks(d), wg = (waves + GRID_BLOCK / 64 - 1) / (GRID_BLOCK / 64);
    if (d->rank <= 16) PPSCI_LAUNCH(spinn_grid_fwd_mfma_kernel<1>, GridArgs, wg, GRID_BLOCK, 0, stream, a);
    else if (d->rank <= 32) PPSCI_LAUNCH(spinn_grid_fwd_mfma_kernel<2>, GridArgs, wg, GRID_BLOCK, 0, stream, a);
    else PPSCI_LAUNCH(spinn_grid_fwd_mfma_kernel<4>, GridArgs, wg, GRID_BLOCK, 0, stream, a);
    int e = PPSCI_LAST_LAUNCH_ERROR();
    if (e != 0) { ppsci_set_error("spinn_grid_fwd: launch failed (%d)", e); return PPSCI_E_LAUNCH; }
    return PPSCI_OK;
  }
  const int grid = grid_fwd_blocks(d);
  a.iters = grid_fwd_kc(d);
  const size_t lds = (size_t)(2 * d->rank * a.iters + GRID_JT * 2 * d->rank + GRID_BLOCK) * sizeof(float);
  if (PPSCI_SET_MAX_LDS(spinn_grid_fwd_kernel, lds) != 0) {
    ppsci_set_error("spinn_grid_fwd: cannot raise dynamic LDS to %zu B", lds);
    return PPSCI_E_LAUNCH;
  }
  PPSCI_LAUNCH(spinn_grid_fwd_kernel, GridArgs, grid, GRID_BLOCK, lds, stream, a);
  int e = PPSCI_LAST_LAUNCH_ERROR();
  if (e != 0) { ppsci_set_error("spinn_grid_fwd: launch failed (%d)", e); return PPSCI_E_LAUNCH; }
  return PPSCI_OK;
}

extern "C" int64_t ppsci_spinn_grid_bwd_scratch_floats(const ppsci_spinn_grid_desc* d) {
  if (!d || d->rank < 1) return 0;
  long long m = 0;
  for (int ax = 0; ax < 3; ++ax) m += (long long)d->n[ax] * grid_bwd_groups(d, ax) * 2 * d->rank;  // one region per axis
  return m;
}

extern "C" int ppsci_spinn_grid_bwd(const ppsci_spinn_grid_desc* d, const float* Fx, const float* Fy, const float* Fz,
                                    const float* gadj, float* scratch, float* Fbar_x, float* Fbar_y, float* Fbar_z,
                                    void* stream) {
  const bool nosum = !Fbar_x && !Fbar_y && !Fbar_z;  // the group partials stay in `scratch` (ppsci_modmlp_bwd_batch_parts sums them)
  if (!d || !Fx || !Fy || !Fz || !gadj || !scratch || (!nosum && (!Fbar_x || !Fbar_y || !Fbar_z)) || d->rank > GRID_BLOCK ||
      (nosum && !grid_mfma(d))) {
    ppsci_set_error("spinn_grid_bwd: invalid argument");
    return PPSCI_E_INVALID;
  }
  float* outs[3] = {Fbar_x, Fbar_y, Fbar_z};
  if (grid_mfma(d)) {  // the three axes in one launch each for the contraction and for the partial sums
    GridArgs a;
    memset(&a, 0, sizeof(a));
    a.d = *d;
    a.F[0] = Fx; a.F[1] = Fy; a.F[2] = Fz;
    a.gadj = (float*)gadj;
    a.Fpart = scratch;
    a.axis = -1;
    const int rp = ((d->rank + 15) / 16) * 16, RP = (rp % 32 == 0) ? rp + 4 : rp + 20;  // row stride = 4 mod 32
    a.iters = RP;
    const int nt = d->rank <= 16 ? 1 : d->rank <= 32 ? 2 : 4;
    int wg = 0, nsum = 0, ncmax = 0;
    long long off = 0;
    for (int ax = 0; ax < 3; ++ax) {
      const int c = ax == 2 ? 1 : 2;
      a.Fbar3[ax] = outs[ax];
      a.ngrp3[ax] = grid_bwd_groups(d, ax);
      a.wg3[ax] = ((d->n[ax] + 15) / 16) * a.ngrp3[ax];
      a.poff3[ax] = off;
      off += (long long)d->n[ax] * a.ngrp3[ax] * 2 * d->rank;
      wg += a.wg3[ax];
      nsum += d->n[ax];
      if (d->n[c] > ncmax) ncmax = d->n[c];
    }
    const int ncp = (ncmax + 15) & ~15;
    const size_t lds = ((size_t)2 * ncp * RP + (size_t)(GRID_BLOCK / 64) * 2 * nt * 4 * 64) * sizeof(float);
    if (lds > (size_t)PPSCI_LDS_LIMIT_BYTES) {
      ppsci_set_error("spinn_grid_bwd: axis of %d points x rank %d does not fit LDS", ncmax, d->rank);
      return PPSCI_E_UNSUPPORTED;
    }
    int se;
    if (d->rank <= 16) {
      se = PPSCI_SET_MAX_LDS(spinn_grid_bwd_mfma_kernel<1>, lds);
      if (se == 0) PPSCI_LAUNCH(spinn_grid_bwd_mfma_kernel<1>, GridArgs, wg, GRID_BLOCK, lds, stream, a);
    } else if (d->rank <= 32) {
      se = PPSCI_SET_MAX_LDS(spinn_grid_bwd_mfma_kernel<2>, lds);
      if (se == 0) PPSCI_LAUNCH(spinn_grid_bwd_mfma_kernel<2>, GridArgs, wg, GRID_BLOCK, lds, stream, a);
    } else {
      se = PPSCI_SET_MAX_LDS(spinn_grid_bwd_mfma_kernel<4>, lds);
      if (se == 0) PPSCI_LAUNCH(spinn_grid_bwd_mfma_kernel<4>, GridArgs, wg, GRID_BLOCK, lds, stream, a);
    }
    if (se != 0) {
      ppsci_set_error("spinn_grid_bwd: cannot raise dynamic LDS to %zu B", lds);
      return PPSCI_E_LAUNCH;
    }
    int e = PPSCI_LAST_LAUNCH_ERROR();
    if (e != 0) { ppsci_set_error("spinn_grid_bwd: launch failed (%d)", e); return PPSCI_E_LAUNCH; }
    if (nosum) return PPSCI_OK;
    a.wg3[0] = d->n[0]; a.wg3[1] = d->n[1]; a.wg3[2] = d->n[2];
    PPSCI_LAUNCH(spinn_fbar_sum_kernel, GridArgs, nsum, GRID_BLOCK, (size_t)FSUM_PARTS * 2 * d->rank * sizeof(float), stream, a);
    e = PPSCI_LAST_LAUNCH_ERROR();
    if (e != 0) { ppsci_set_error("spinn_fbar_sum: launch failed (%d)", e); return PPSCI_E_LAUNCH; }
    return PPSCI_OK;
  }
  for (int ax = 0; ax < 3; ++ax) {
    GridArgs a;
    memset(&a, 0, sizeof(a));
    a.d = *d;
    a.F[0] = Fx; a.F[1] = Fy; a.F[2] = Fz;
    a.gadj = (float*)gadj;
    a.Fbar = outs[ax];
    a.Fpart = scratch;
    a.axis = ax;
    const int c = ax == 2 ? 1 : 2;
    const int ngrp = grid_bwd_groups(d, ax);
    a.ngrp = ngrp;
    int e;
    {
      const size_t lds = ((size_t)2 * d->n[c] * d->rank + (size_t)GRID_JG * d->n[c] + 2 * GRID_BLOCK) * sizeof(float);
      if (lds > (size_t)PPSCI_LDS_LIMIT_BYTES) {
        ppsci_set_error("spinn_grid_bwd: axis of %d points x rank %d does not fit LDS", d->n[c], d->rank);
        return PPSCI_E_UNSUPPORTED;
      }
      if (PPSCI_SET_MAX_LDS(spinn_grid_bwd_kernel, lds) != 0) {
        ppsci_set_error("spinn_grid_bwd: cannot raise dynamic LDS to %zu B", lds);
        return PPSCI_E_LAUNCH;
      }
      PPSCI_LAUNCH(spinn_grid_bwd_kernel, GridArgs, d->n[ax] * ngrp, GRID_BLOCK, lds, stream, a);
    }
    e = PPSCI_LAST_LAUNCH_ERROR();
    if (e != 0) { ppsci_set_error("spinn_grid_bwd: launch failed (%d)", e); return PPSCI_E_LAUNCH; }
    PPSCI_LAUNCH(spinn_fbar_sum_kernel, GridArgs, d->n[ax], GRID_BLOCK, (size_t)FSUM_PARTS * 2 * d->rank * sizeof(float), stream, a);
    e = PPSCI_LAST_LAUNCH_ERROR();
    if (e != 0) { ppsci_set_error("spinn_fbar_sum: launch failed (%d)", e); return PPSCI_E_LAUNCH; }
  }
  return PPSCI_OK;
}

// The reverse sweep of the three branch nets reading dL/dF as the group partials ppsci_spinn_grid_bwd left in `scratch` (called with
// its three Fbar pointers NULL): the tile kernel sums them on load -- spinn_fbar_sum_kernel's launch is saved.
extern "C" int ppsci_modmlp_bwd_parts_supported(const ppsci_modmlp_desc* d, const ppsci_spinn_grid_desc* gd) {
  return (d && gd && mod_check(d) == PPSCI_OK && mod_tiled(d) && grid_mfma(gd) && gd->rank == d->d_out) ? 1 : 0;
}

extern "C" int ppsci_modmlp_bwd_batch_parts(const ppsci_modmlp_desc* d, const ppsci_spinn_grid_desc* gd, const float* const* params,
                                            const float* const* x, const float* scratch, const float* const* stash,
                                            float* const* grad_partials, int64_t partial_stride, void* stream) {
  if (!ppsci_modmlp_bwd_parts_supported(d, gd) || !scratch) {
    ppsci_set_error("modmlp_bwd_parts: unsupported shape (tile kernel + MFMA grid kernels only)");
    return PPSCI_E_UNSUPPORTED;
  }
  const float* fpart[3];
  int ngrp[3];
  int64_t n[3];
  long long off = 0;
  for (int ax = 0; ax < 3; ++ax) {  // the layout ppsci_spinn_grid_bwd writes: one region per axis, [n_ax][groups][2][R]
    ngrp[ax] = grid_bwd_groups(gd, ax);
    fpart[ax] = scratch + off;
    off += (long long)gd->n[ax] * ngrp[ax] * 2 * gd->rank;
    n[ax] = gd->n[ax];
  }
  return modmlp_bwd_batch_impl(d, 3, params, n, x, nullptr, fpart, ngrp, stash, grad_partials, partial_stride, stream);
}
