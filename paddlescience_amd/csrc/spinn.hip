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

struct ModArgs {
  ppsci_modmlp_desc d;
  ModOff o;
  const float* params;
  const float* x;      // [N]
  float* F;            // fwd out: [3][N][R]
  const float* Fbar;   // bwd in : [3][N][R]
  float* stash;        // [N][(L+2)][3][H]: zu, zv, z_0 .. z_{L-1}
  float* partials;     // bwd out: [N][P]
  int N;
};

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

__global__ void __launch_bounds__(SP_MAXH) modmlp_fwd_kernel(ModArgs a) {
  PPSCI_DYN_SMEM(sh);  // [3][H]
  const int f = threadIdx.x, H = a.d.width, L = a.d.n_hidden, R = a.d.d_out, act = a.d.activation;
  const int pt = blockIdx.x;
  const float x = a.x[pt];
  const float* P = a.params;
  float* st = a.stash ? a.stash + (long long)pt * (L + 2) * 3 * H : nullptr;
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
    for (int k = 0; k < H; ++k) {
      const float w = W[k * R + r];
      s0 += w * sh[k]; s1 += w * sh[H + k]; s2 += w * sh[2 * H + k];
    }
    a.F[((long long)0 * a.N + pt) * R + r] = s0;
    a.F[((long long)1 * a.N + pt) * R + r] = s1;
    a.F[((long long)2 * a.N + pt) * R + r] = s2;
  }
}

__global__ void __launch_bounds__(SP_MAXH) modmlp_bwd_kernel(ModArgs a) {
  PPSCI_DYN_SMEM(sh);  // [3][H] exchange + [3][R] Fbar
  const int f = threadIdx.x, H = a.d.width, L = a.d.n_hidden, R = a.d.d_out, act = a.d.activation;
  const int pt = blockIdx.x;
  const float x = a.x[pt];
  const float* P = a.params;
  const float* st = a.stash + (long long)pt * (L + 2) * 3 * H;
  float* G = a.partials + (long long)pt * a.o.P;
  float* fb = sh + 3 * H;
  for (int i = f; i < 3 * R; i += blockDim.x) fb[i] = a.Fbar[((long long)(i / R) * a.N + pt) * R + (i % R)];
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
  __syncthreads();
  // last_fc: obar_s[k] = sum_r WL[k][r] Fbar_s[r];  gWL[k][r] = sum_s o_s[k] Fbar_s[r];  gbL[r] = Fbar_0[r]
  float ob[3] = {0, 0, 0};
  if (f < H) {
    const float* W = P + a.o.wl + f * R;
    for (int r = 0; r < R; ++r) {
      const float w = W[r];
      ob[0] += w * fb[r]; ob[1] += w * fb[R + r]; ob[2] += w * fb[2 * R + r];
      G[a.o.wl + f * R + r] = ol[0] * fb[r] + ol[1] * fb[R + r] + ol[2] * fb[2 * R + r];
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
    if (f < H) { sh[f] = op[0]; sh[H + f] = op[1]; sh[2 * H + f] = op[2]; }
    __syncthreads();
    if (f < H) {  // gW_l[k][f] = sum_s o_prev_s[k] zbar_s[f]
      float* gw = G + a.o.w[l];
      for (int k = 0; k < H; ++k) gw[k * H + f] = sh[k] * zb[0] + sh[H + k] * zb[1] + sh[2 * H + k] * zb[2];
    }
    __syncthreads();
    if (f < H) { sh[f] = zb[0]; sh[H + f] = zb[1]; sh[2 * H + f] = zb[2]; }
    __syncthreads();
    if (f < H) {  // obar_prev_s[k=f] = sum_j W_l[f][j] zbar_s[j]
      const float* W = P + a.o.w[l] + f * H;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f;
      for (int j = 0; j < H; ++j) {
        const float w = W[j];
        s0 += w * sh[j]; s1 += w * sh[H + j]; s2 += w * sh[2 * H + j];
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
};

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

__global__ void __launch_bounds__(GRID_BLOCK) spinn_fbar_sum_kernel(GridArgs a) {
  const int R = a.d.rank, ax = a.axis, na = a.d.n[ax];
  const int b = ax == 0 ? 1 : 0;
  const int ngrp = (a.d.n[b] + GRID_JG - 1) / GRID_JG;
  const long long t = (long long)blockIdx.x * GRID_BLOCK + threadIdx.x;
  if (t >= (long long)na * R) return;
  const int i = (int)(t / R), r = (int)(t % R);
  float s0 = 0.f, s2 = 0.f;
  for (int g = 0; g < ngrp; ++g) {
    const float* p = a.Fpart + ((long long)i * ngrp + g) * 2 * R;
    s0 += p[r];
    s2 += p[R + r];
  }
  const long long sa = (long long)na * R;
  a.Fbar[t] = s0;
  a.Fbar[sa + t] = 0.f;  // first-derivative stream is not used by these residuals
  a.Fbar[2 * sa + t] = s2;
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

extern "C" int ppsci_modmlp_fwd(const ppsci_modmlp_desc* d, const float* params, int64_t n, const float* x, float* F,
                                float* stash, void* stream) {
  if (mod_check(d) != PPSCI_OK || !params || !x || !F || n < 1) {
    ppsci_set_error("modmlp_fwd: invalid argument");
    return PPSCI_E_INVALID;
  }
  ModArgs a;
  memset(&a, 0, sizeof(a));
  a.d = *d;
  mod_offsets(*d, a.o);
  a.params = params; a.x = x; a.F = F; a.stash = stash; a.N = (int)n;
  PPSCI_LAUNCH(modmlp_fwd_kernel, ModArgs, (int)n, block_for(d->width), 3 * d->width * sizeof(float), stream, a);
  int e = PPSCI_LAST_LAUNCH_ERROR();
  if (e != 0) { ppsci_set_error("modmlp_fwd: launch failed (%d)", e); return PPSCI_E_LAUNCH; }
  return PPSCI_OK;
}

extern "C" int ppsci_modmlp_bwd(const ppsci_modmlp_desc* d, const float* params, int64_t n, const float* x,
                                const float* Fbar, const float* stash, float* grad_partials, void* stream) {
  if (mod_check(d) != PPSCI_OK || !params || !x || !Fbar || !stash || !grad_partials || n < 1) {
    ppsci_set_error("modmlp_bwd: invalid argument");
    return PPSCI_E_INVALID;
  }
  ModArgs a;
  memset(&a, 0, sizeof(a));
  a.d = *d;
  mod_offsets(*d, a.o);
  a.params = params; a.x = x; a.Fbar = Fbar; a.stash = (float*)stash; a.partials = grad_partials; a.N = (int)n;
  PPSCI_LAUNCH(modmlp_bwd_kernel, ModArgs, (int)n, block_for(d->width),
               (3 * d->width + 3 * d->d_out) * sizeof(float), stream, a);
  int e = PPSCI_LAST_LAUNCH_ERROR();
  if (e != 0) { ppsci_set_error("modmlp_bwd: launch failed (%d)", e); return PPSCI_E_LAUNCH; }
  return PPSCI_OK;
}

static int grid_fwd_blocks(const ppsci_spinn_grid_desc* d) {
  return d->n[0] * ((d->n[1] + GRID_JT - 1) / GRID_JT);
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
  for (int ax = 0; ax < 3; ++ax) {
    const int b = ax == 0 ? 1 : 0;
    const long long v = (long long)d->n[ax] * ((d->n[b] + GRID_JG - 1) / GRID_JG) * 2 * d->rank;
    if (v > m) m = v;
  }
  return m;
}

extern "C" int ppsci_spinn_grid_bwd(const ppsci_spinn_grid_desc* d, const float* Fx, const float* Fy, const float* Fz,
                                    const float* gadj, float* scratch, float* Fbar_x, float* Fbar_y, float* Fbar_z,
                                    void* stream) {
  if (!d || !Fx || !Fy || !Fz || !gadj || !scratch || !Fbar_x || !Fbar_y || !Fbar_z || d->rank > GRID_BLOCK) {
    ppsci_set_error("spinn_grid_bwd: invalid argument");
    return PPSCI_E_INVALID;
  }
  float* outs[3] = {Fbar_x, Fbar_y, Fbar_z};
  for (int ax = 0; ax < 3; ++ax) {
    GridArgs a;
    memset(&a, 0, sizeof(a));
    a.d = *d;
    a.F[0] = Fx; a.F[1] = Fy; a.F[2] = Fz;
    a.gadj = (float*)gadj;
    a.Fbar = outs[ax];
    a.Fpart = scratch;
    a.axis = ax;
    const int b = ax == 0 ? 1 : 0, c = ax == 2 ? 1 : 2;
    const int ngrp = (d->n[b] + GRID_JG - 1) / GRID_JG;
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
    int e = PPSCI_LAST_LAUNCH_ERROR();
    if (e != 0) { ppsci_set_error("spinn_grid_bwd: launch failed (%d)", e); return PPSCI_E_LAUNCH; }
    PPSCI_LAUNCH(spinn_fbar_sum_kernel, GridArgs, (d->n[ax] * d->rank + GRID_BLOCK - 1) / GRID_BLOCK, GRID_BLOCK, 0, stream, a);
    e = PPSCI_LAST_LAUNCH_ERROR();
    if (e != 0) { ppsci_set_error("spinn_fbar_sum: launch failed (%d)", e); return PPSCI_E_LAUNCH; }
  }
  return PPSCI_OK;
}
