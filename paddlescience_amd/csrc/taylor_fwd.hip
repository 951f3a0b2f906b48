// taylor_fwd.hip -- MLP forward with Taylor-mode derivative streams, fused per 16-point tile.
//
// Replaces, for one constraint's batch, the reference's eager chain
//   MLP.forward            /root/reference/ppsci/arch/mlp.py:298-315  (L+1 GEMMs + L activations)
//   jacobian()/hessian()   /root/reference/ppsci/autodiff/ad.py:56-77,181-236 (one full reverse
//                          sweep of the net per call, recorded for higher order)
// with ONE kernel that carries S = 1+n1+n2 streams (value, first and pure-second directional
// derivatives) through the net per point.
//
// Mapping to gfx950: see taylor_tile.h.  Hidden-layer weights are staged in LDS as A-fragments
// (one ds_read_b128 = 4 k-steps, shared by all S streams); activations never leave registers;
// the S*NB*NB*4 v_mfma_f32_16x16x4_f32 per layer and tile are the only heavy work.
#include "taylor_tile.h"

struct FwdArgs {
  ppsci_mlp_desc d;
  ppsci_derived q;
  const float* params;
  const float* x[PPSCI_MAX_IN];
  float* U;
  f32x4* stash;  // may be null
  long long N;
  int ntiles;
  int iters;     // tile iterations per wave (uniform over the grid)
  int resident;  // 1: all hidden-layer fragments stay in LDS; 0: re-staged per layer (lock-step)
};

// LDS carve (floats).  small = W0s[d0*HP] + Bs[L*HP] + WLs[m*HP] + BLs[4*ceil(m/4)]
static inline int fwd_small_floats(const ppsci_mlp_desc& d, const ppsci_derived& q) {
  return (q.d0 + d.n_hidden + d.d_out) * q.HP + ((d.d_out + 3) / 4) * 4;
}

template <int NB, int N1, int N2>
__global__ void __launch_bounds__(PPSCI_BLOCK) taylor_fwd_kernel(FwdArgs a) {
  constexpr int S = 1 + N1 + N2;
  constexpr int HP = 16 * NB;
  PPSCI_DYN_SMEM(smem);
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, c = lane & 15;
  const int L = a.d.n_hidden, H = a.d.width, m = a.d.d_out, d0 = a.q.d0;
  const int act = a.d.activation;

  float* W0s = smem;
  float* Bs = W0s + d0 * HP;
  float* WLs = Bs + L * HP;
  float* BLs = WLs + m * HP;
  float* frag = BLs + ((m + 3) / 4) * 4;

  // ---- stage the small operands once
  for (int idx = tid; idx < d0 * HP; idx += nthr) {
    int k = idx / HP, f = idx - k * HP;
    W0s[idx] = (f < H) ? a.params[a.q.offW[0] + k * H + f] : 0.f;
  }
  for (int idx = tid; idx < L * HP; idx += nthr) {
    int l = idx / HP, f = idx - l * HP;
    Bs[idx] = (f < H) ? a.params[a.q.offB[l] + f] : 0.f;
  }
  for (int idx = tid; idx < m * HP; idx += nthr) {
    int cc = idx / HP, f = idx - cc * HP;
    WLs[idx] = (f < H) ? a.params[a.q.offW[L] + f * m + cc] : 0.f;
  }
  for (int idx = tid; idx < m; idx += nthr) BLs[idx] = a.params[a.q.offB[L] + idx];
  if (a.resident) {
    for (int l = 1; l < L; ++l) ppsci_stage_fragF(frag + (l - 1) * HP * HP, a.params + a.q.offW[l], H, NB, tid, nthr);
  }
  __syncthreads();

  for (int it = 0; it < a.iters; ++it) {
    const int tile = (it * (int)gridDim.x + (int)blockIdx.x) * PPSCI_WAVES_PER_BLOCK + wave;
    const bool tile_ok = tile < a.ntiles;
    const long long p = (long long)tile * PPSCI_TILE + c;
    const bool valid = tile_ok && p < a.N;

    // ------------------------------------------------------------------ layer 0 (K = d0, VALU)
    f32x4 h[S][NB];
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int b = 0; b < NB; ++b) h[s][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
      int k = 0;
      for (int j = 0; j < a.d.d_raw; ++j) {
        const float xj = valid ? a.x[j][p] : 0.f;
        const int e = a.d.embed[j];
        const int ncol = (e == PPSCI_EMBED_PERIOD) ? 2 : 1;
        float sn = 0.f, cs = 0.f;
        const float w = a.d.omega[j];
        if (e == PPSCI_EMBED_PERIOD) {
          sn = sinf(w * xj);
          cs = cosf(w * xj);
        }
        for (int qq = 0; qq < ncol; ++qq, ++k) {
          float val, d1v[N1 > 0 ? N1 : 1], d2v[N2 > 0 ? N2 : 1];
          if (e != PPSCI_EMBED_PERIOD) {
            val = xj;
#pragma unroll
            for (int i = 0; i < N1; ++i) d1v[i] = a.d.dirs[i][j];
#pragma unroll
            for (int i = 0; i < N2; ++i) d2v[i] = 0.f;
          } else {
            val = (qq == 0) ? cs : sn;
            const float dv = (qq == 0) ? -w * sn : w * cs;
            const float ddv = (qq == 0) ? -w * w * cs : -w * w * sn;
#pragma unroll
            for (int i = 0; i < N1; ++i) d1v[i] = dv * a.d.dirs[i][j];
#pragma unroll
            for (int i = 0; i < N2; ++i) d2v[i] = ddv * a.d.dirs[i][j] * a.d.dirs[i][j];
          }
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            const f32x4 w4 = *(const f32x4*)&W0s[k * HP + 16 * b + 4 * g];
            h[0][b] += w4 * val;
#pragma unroll
            for (int i = 0; i < N1; ++i) h[1 + i][b] += w4 * d1v[i];
#pragma unroll
            for (int i = 0; i < N2; ++i) h[1 + N1 + i][b] += w4 * d2v[i];
          }
        }
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) h[0][b] += *(const f32x4*)&Bs[16 * b + 4 * g];
    }

    for (int l = 0; l < L; ++l) {
      if (l > 0) {
        // -------------------------------------------------------------- hidden layer l (MFMA)
        const float* fr;
        if (a.resident) {
          fr = frag + (l - 1) * HP * HP;
        } else {
          __syncthreads();
          ppsci_stage_fragF(frag, a.params + a.q.offW[l], H, NB, tid, nthr);
          __syncthreads();
          fr = frag;
        }
        f32x4 acc[S][NB];
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) {
          acc[0][ob] = *(const f32x4*)&Bs[l * HP + 16 * ob + 4 * g];
#pragma unroll
          for (int s = 1; s < S; ++s) acc[s][ob] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) {
#pragma unroll
          for (int kb = 0; kb < NB; ++kb) {
            const f32x4 a4 = *(const f32x4*)&fr[((ob * NB + kb) * 64 + lane) * 4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
              for (int s = 0; s < S; ++s)
                acc[s][ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[r], h[s][kb][r], acc[s][ob], 0, 0, 0);
            }
          }
        }
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
          for (int b = 0; b < NB; ++b) h[s][b] = acc[s][b];
      }
      // ---------------------------------------------------------------- skip quirk, stash, act
      const float zs = ppsci_zscale(a.d, l);
      if (zs != 1.f) {
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
          for (int b = 0; b < NB; ++b) h[s][b] *= zs;
      }
      if (a.stash != nullptr && tile_ok) {
        f32x4* st = a.stash + ((long long)tile * L + l) * (S * NB * 64);
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
          for (int b = 0; b < NB; ++b) st[(s * NB + b) * 64 + lane] = h[s][b];
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float sv, d1, d2, d3;
          ppsci_act_eval(act, h[0][b][r], sv, d1, d2, d3);
          h[0][b][r] = sv;
#pragma unroll
          for (int i = 0; i < N1; ++i) {
            const float zi = h[1 + i][b][r];
            h[1 + i][b][r] = d1 * zi;
            if (i < N2) {
              const float zii = h[1 + N1 + i][b][r];
              h[1 + N1 + i][b][r] = d2 * zi * zi + d1 * zii;
            }
          }
        }
      }
    }

    // ------------------------------------------------------------------ last linear (N = m, VALU)
    for (int cc = 0; cc < m; ++cc) {
#pragma unroll
      for (int s = 0; s < S; ++s) {
        float part = 0.f;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const f32x4 w4 = *(const f32x4*)&WLs[cc * HP + 16 * b + 4 * g];
          const f32x4 t = w4 * h[s][b];
          part += (t[0] + t[1]) + (t[2] + t[3]);
        }
        part = ppsci_group_sum4(part);
        if (s == 0) part += BLs[cc];
        if (g == 0 && valid) a.U[((long long)cc * S + s) * a.N + p] = part;
      }
    }
  }
}

// ------------------------------------------------------------------------------------ host side
#include <stdio.h>
#include <string.h>
extern "C" void ppsci_set_error(const char* fmt, ...);

template <int NB, int N1, int N2>
static int launch_fwd(FwdArgs& a, void* stream) {
  const int HP = 16 * NB, L = a.d.n_hidden;
  const int small = fwd_small_floats(a.d, a.q);
  long long res_floats = (long long)small + (long long)(L - 1) * HP * HP;
  a.resident = (res_floats * 4 <= PPSCI_LDS_LIMIT_BYTES - 1024) ? 1 : 0;
  long long lds_floats = a.resident ? res_floats : (long long)small + (long long)HP * HP;
  if (lds_floats * 4 > PPSCI_LDS_LIMIT_BYTES) {
    ppsci_set_error("taylor_fwd: LDS need %lld B exceeds %d B", lds_floats * 4, PPSCI_LDS_LIMIT_BYTES);
    return PPSCI_E_UNSUPPORTED;
  }
  const int lds = (int)(lds_floats * 4);
  const int blocks_needed = (a.ntiles + PPSCI_WAVES_PER_BLOCK - 1) / PPSCI_WAVES_PER_BLOCK;
  int per_cu = PPSCI_LDS_LIMIT_BYTES / (lds + 256);
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 4) per_cu = 4;
  int grid = blocks_needed < 256 * per_cu ? blocks_needed : 256 * per_cu;
  if (ppsci_get_max_grid() > 0 && grid > ppsci_get_max_grid()) grid = ppsci_get_max_grid();
  if (grid < 1) grid = 1;
  a.iters = (blocks_needed + grid - 1) / grid;
  if (PPSCI_SET_MAX_LDS((taylor_fwd_kernel<NB, N1, N2>), lds) != 0) {
    ppsci_set_error("taylor_fwd: cannot raise dynamic LDS to %d B", lds);
    return PPSCI_E_LAUNCH;
  }
  PPSCI_LAUNCH((taylor_fwd_kernel<NB, N1, N2>), FwdArgs, grid, PPSCI_BLOCK, lds, stream, a);
  int e = PPSCI_LAST_LAUNCH_ERROR();
  if (e != 0) {
    ppsci_set_error("taylor_fwd: launch failed (hip error %d)", e);
    return PPSCI_E_LAUNCH;
  }
  return PPSCI_OK;
}

#define PPSCI_FWD_CASE(NB_, N1_, N2_) \
  if (q.NB == NB_ && d->n1 == N1_ && d->n2 == N2_) return launch_fwd<NB_, N1_, N2_>(a, stream);

extern "C" int ppsci_taylor_fwd(const ppsci_mlp_desc* d, const float* params, int64_t n_points,
                                const float* const* inputs_host, float* U, void* stash, void* stream) {
  ppsci_derived q;
  if (!d || !params || !inputs_host || !U || n_points < 0 || ppsci_derive(d, &q) != PPSCI_OK) {
    ppsci_set_error("taylor_fwd: invalid argument");
    return PPSCI_E_INVALID;
  }
  if (n_points == 0) return PPSCI_OK;
  FwdArgs a;
  memset(&a, 0, sizeof(a));
  a.d = *d;
  a.q = q;
  a.params = params;
  for (int j = 0; j < d->d_raw; ++j) a.x[j] = inputs_host[j];
  a.U = U;
  a.stash = (f32x4*)stash;
  a.N = n_points;
  a.ntiles = (int)((n_points + PPSCI_TILE - 1) / PPSCI_TILE);
  // instantiated (padded width / 16, n1, n2) combinations (ppsci_derive rounds NB up to 2/4/8)
  PPSCI_FWD_CASE(2, 0, 0) PPSCI_FWD_CASE(4, 0, 0) PPSCI_FWD_CASE(8, 0, 0)
  PPSCI_FWD_CASE(2, 1, 1) PPSCI_FWD_CASE(4, 1, 1) PPSCI_FWD_CASE(8, 1, 1)
  PPSCI_FWD_CASE(2, 2, 0) PPSCI_FWD_CASE(4, 2, 0) PPSCI_FWD_CASE(8, 2, 0)
  PPSCI_FWD_CASE(2, 2, 1) PPSCI_FWD_CASE(4, 2, 1) PPSCI_FWD_CASE(8, 2, 1)
  PPSCI_FWD_CASE(2, 2, 2) PPSCI_FWD_CASE(4, 2, 2) PPSCI_FWD_CASE(8, 2, 2)
  PPSCI_FWD_CASE(2, 3, 3) PPSCI_FWD_CASE(4, 3, 3)
  ppsci_set_error("taylor_fwd: unsupported (width=%d -> NB=%d, n1=%d, n2=%d)", d->width, q.NB, d->n1, d->n2);
  return PPSCI_E_UNSUPPORTED;
}
