// taylor_bwd.hip -- reverse sweep through the Taylor-mode forward: dL/dparams from dL/dU.
//
// Replaces `total_loss.backward()` (/root/reference/ppsci/solver/train.py:158), which in the
// reference differentiates through the whole double-backward graph built by
// ppsci/autodiff/ad.py:56-77.  Math: SURVEY.md Appendix A "Backward".
//
// Per 16-point tile (one wave), for hidden layer l = L-1 .. 0 with pre-activation streams Z_l
// read back from the stash written by taylor_fwd:
//   h_l       = act-streams(Z_l)                         (recomputed, T layout)
//   Wbar_{l+1}+= h_l (x) zbar_{l+1}   contraction over points: both operands transposed to the
//                N layout through a per-wave LDS scratch, then 16x16x4 MFMAs; the 16x16 result
//                blocks are added into a per-block LDS accumulator with ds_add_f32
//   zbar_l    = pointwise(hbar_l, Z_l)                   (in place, T layout)
//   hbar_{l-1}= W_l zbar_l            MFMA with W_l staged in LDS as "B-fragments"
// Layer 0 (K = d0) and the last linear (N = m) are VALU + 16-lane row reductions.
// Each block flushes its LDS accumulators to its own row of grad_partials; ppsci_reduce_rows
// sums the rows in a fixed order.
#include "taylor_tile.h"

struct BwdArgs {
  ppsci_mlp_desc d;
  ppsci_derived q;
  const float* params;
  const float* x[PPSCI_MAX_IN];
  const float* Ubar;
  const f32x4* stash;
  float* partials;  // [gridDim.x, P]
  long long N;
  int ntiles;
  int iters;
  int resident;
};

// LDS carve (floats): WLs[m*HP] gW0[d0*HP] gB[L*HP] gWL[m*HP] gBL[4*ceil(m/4)]
//                     scratch[WAVES*SCR] fragB[(L-1 or 1)*HP*HP] gWh[(L-1 or 1)*HP*HP]
static inline int bwd_small_floats(const ppsci_mlp_desc& d, const ppsci_derived& q) {
  return (2 * d.d_out + q.d0 + d.n_hidden) * q.HP + ((d.d_out + 3) / 4) * 4 +
         PPSCI_WAVES_PER_BLOCK * PPSCI_SCR_FLOATS;
}

template <int NB, int N1, int N2>
__global__ void __launch_bounds__(PPSCI_BLOCK) taylor_bwd_kernel(BwdArgs a) {
  constexpr int S = 1 + N1 + N2;
  constexpr int HP = 16 * NB;
  PPSCI_DYN_SMEM(smem);
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, c = lane & 15;
  const int L = a.d.n_hidden, H = a.d.width, m = a.d.d_out, d0 = a.q.d0;
  const int act = a.d.activation;
  const int nslot = a.resident ? (L - 1) : 1;

  float* WLs = smem;
  float* gW0 = WLs + m * HP;
  float* gB = gW0 + d0 * HP;
  float* gWL = gB + L * HP;
  float* gBL = gWL + m * HP;
  float* scr = gBL + ((m + 3) / 4) * 4 + wave * PPSCI_SCR_FLOATS;
  float* fragB = gBL + ((m + 3) / 4) * 4 + PPSCI_WAVES_PER_BLOCK * PPSCI_SCR_FLOATS;
  float* gWh = fragB + nslot * HP * HP;
  float* prow = a.partials + (long long)blockIdx.x * a.q.P;

  for (int idx = tid; idx < m * HP; idx += nthr) {
    int cc = idx / HP, f = idx - cc * HP;
    WLs[idx] = (f < H) ? a.params[a.q.offW[L] + f * m + cc] : 0.f;
  }
  const int nacc_small = (d0 + L + m) * HP + ((m + 3) / 4) * 4;  // gW0 .. gBL are contiguous
  for (int idx = tid; idx < nacc_small; idx += nthr) gW0[idx] = 0.f;
  for (int idx = tid; idx < nslot * HP * HP; idx += nthr) gWh[idx] = 0.f;
  if (a.resident) {
    for (int l = 1; l < L; ++l) ppsci_stage_fragB(fragB + (l - 1) * HP * HP, a.params + a.q.offW[l], H, NB, tid, nthr);
  }
  __syncthreads();

  for (int it = 0; it < a.iters; ++it) {
    const int tile = (it * (int)gridDim.x + (int)blockIdx.x) * PPSCI_WAVES_PER_BLOCK + wave;
    const bool tile_ok = tile < a.ntiles;
    const long long p = (long long)tile * PPSCI_TILE + c;
    const bool valid = tile_ok && p < a.N;
    const f32x4* st_tile = a.stash + (long long)(tile_ok ? tile : 0) * L * (S * NB * 64);

    f32x4 hb[S][NB];  // adjoint of h_l streams, turned into zbar_l in place (T layout)
    f32x4 zN[S][NB];  // zbar_{l+1} in N layout
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        hb[s][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        zN[s][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }

    // ---- adjoint of the last linear: hbar_{L-1,s}[f] = sum_c W_last[f][c] * ubar[c][s]
    for (int cc = 0; cc < m; ++cc) {
      float ub[S];
#pragma unroll
      for (int s = 0; s < S; ++s) ub[s] = valid ? a.Ubar[((long long)cc * S + s) * a.N + p] : 0.f;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const f32x4 w4 = *(const f32x4*)&WLs[cc * HP + 16 * b + 4 * g];
#pragma unroll
        for (int s = 0; s < S; ++s) hb[s][b] += w4 * ub[s];
      }
      // bias of the last linear: sum over points of the value-stream adjoint
      const float sb = ppsci_row_sum16(ub[0]);
      if (lane == 0) atomicAdd(&gBL[cc], sb);
    }

    for (int l = L - 1; l >= 0; --l) {
      const f32x4* st = st_tile + (long long)l * (S * NB * 64);
      const float zs = ppsci_zscale(a.d, l);
      int slot_next = a.resident ? l : 0;  // gWh slot of layer l+1 (index l+1-1)
      if (!a.resident) {
        // lock-step per layer: stage W_l fragments (needed below for hbar_{l-1}); gWh was
        // zeroed by the flush at the end of the previous layer iteration
        __syncthreads();
        if (l > 0) ppsci_stage_fragB(fragB, a.params + a.q.offW[l], H, NB, tid, nthr);
        __syncthreads();
      }
#pragma unroll
      for (int ib = 0; ib < NB; ++ib) {
        f32x4 z[S];
#pragma unroll
        for (int s = 0; s < S; ++s) z[s] = tile_ok ? st[(s * NB + ib) * 64 + lane] : (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 hc[S];  // h_l streams of this block (T layout)
        f32x4 D1, D2, D3;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float sv, d1, d2, d3;
          ppsci_act_eval(act, z[0][r], sv, d1, d2, d3);
          D1[r] = d1;
          D2[r] = d2;
          D3[r] = d3;
          hc[0][r] = sv;
#pragma unroll
          for (int i = 0; i < N1; ++i) {
            hc[1 + i][r] = d1 * z[1 + i][r];
            if (i < N2) hc[1 + N1 + i][r] = d2 * z[1 + i][r] * z[1 + i][r] + d1 * z[1 + N1 + i][r];
          }
        }
        if (l == L - 1) {
          // Wbar_last[f][c] += sum_pts sum_s h_s[f] * ubar[c][s]
          for (int cc = 0; cc < m; ++cc) {
            f32x4 t = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < S; ++s) {
              const float ub = valid ? a.Ubar[((long long)cc * S + s) * a.N + p] : 0.f;
              t += hc[s] * ub;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float v = ppsci_row_sum16(t[r]);
              if (c == 0) atomicAdd(&gWL[cc * HP + 16 * ib + 4 * g + r], v);
            }
          }
        } else {
          // Wbar_{l+1}[16ib.., :] += sum_s sum_pts h_s[in][pt] * zbar_{l+1,s}[out][pt]
          f32x4 hN[S];
#pragma unroll
          for (int s = 0; s < S; ++s) hN[s] = ppsci_t2n(hc[s], scr, g, c);
          float* gw = gWh + slot_next * HP * HP;
#pragma unroll
          for (int ob = 0; ob < NB; ++ob) {
            f32x4 D = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
              for (int step = 0; step < 4; ++step)
                D = __builtin_amdgcn_mfma_f32_16x16x4f32(hN[s][step], zN[s][ob][step], D, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(&gw[(16 * ib + 4 * g + r) * HP + 16 * ob + c], D[r]);
          }
        }
        // zbar_l for this block, in place of hbar_l
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d1 = D1[r], d2 = D2[r], d3 = D3[r];
          float accv = d1 * hb[0][ib][r];
#pragma unroll
          for (int i = 0; i < N1; ++i) {
            const float zi = z[1 + i][r];
            const float hbi = hb[1 + i][ib][r];
            float zbi = d1 * hbi;
            accv += d2 * zi * hbi;
            if (i < N2) {
              const float zii = z[1 + N1 + i][r];
              const float hb2 = hb[1 + N1 + i][ib][r];
              hb[1 + N1 + i][ib][r] = zs * d1 * hb2;
              zbi += 2.f * d2 * zi * hb2;
              accv += (d3 * zi * zi + d2 * zii) * hb2;
            }
            hb[1 + i][ib][r] = zs * zbi;
          }
          hb[0][ib][r] = zs * accv;
          const float sb = ppsci_row_sum16(hb[0][ib][r]);
          if (c == 0) atomicAdd(&gB[l * HP + 16 * ib + 4 * g + r], sb);
        }
      }

      if (l > 0) {
        // hbar_{l-1} = W_l zbar_l
        const float* fr = a.resident ? fragB + (l - 1) * HP * HP : fragB;
        f32x4 acc[S][NB];
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
          for (int b = 0; b < NB; ++b) acc[s][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ib = 0; ib < NB; ++ib) {
#pragma unroll
          for (int kb = 0; kb < NB; ++kb) {
            const f32x4 a4 = *(const f32x4*)&fr[((ib * NB + kb) * 64 + lane) * 4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
              for (int s = 0; s < S; ++s)
                acc[s][ib] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[r], hb[s][kb][r], acc[s][ib], 0, 0, 0);
            }
          }
        }
        // zbar_l -> N layout for the next iteration's weight-gradient GEMM; hbar_{l-1} takes over hb
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            zN[s][b] = ppsci_t2n(hb[s][b], scr, g, c);
            hb[s][b] = acc[s][b];
          }
      } else {
        // layer 0: Wbar_0[k][f] += sum_pts sum_s h0_s[k] * zbar_0,s[f]   (h0 = embedded input streams)
        int k = 0;
        for (int j = 0; j < a.d.d_raw; ++j) {
          const float xj = valid ? a.x[j][p] : 0.f;
          const int e = a.d.embed[j];
          const int ncol = (e == PPSCI_EMBED_PERIOD) ? 2 : 1;
          const float w = a.d.omega[j];
          float sn = 0.f, cs = 0.f;
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
            if (!valid) val = 0.f;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
              f32x4 t = hb[0][b] * val;
#pragma unroll
              for (int i = 0; i < N1; ++i) t += hb[1 + i][b] * d1v[i];
#pragma unroll
              for (int i = 0; i < N2; ++i) t += hb[1 + N1 + i][b] * d2v[i];
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float v = ppsci_row_sum16(t[r]);
                if (c == 0) atomicAdd(&gW0[k * HP + 16 * b + 4 * g + r], v);
              }
            }
          }
        }
      }

      if (!a.resident && l < L - 1) {
        // flush Wbar_{l+1} of this iteration into the block's partial row and re-zero
        __syncthreads();
        const int off = a.q.offW[l + 1];
        for (int idx = tid; idx < H * H; idx += nthr) {
          int in = idx / H, out = idx - in * H;
          float v = gWh[in * HP + out];
          if (it == 0) prow[off + idx] = v; else prow[off + idx] += v;
        }
        __syncthreads();
        for (int idx = tid; idx < HP * HP; idx += nthr) gWh[idx] = 0.f;
      }
    }
  }

  // ---- flush the block's accumulators to its partial row (canonical parameter layout)
  __syncthreads();
  for (int idx = tid; idx < d0 * H; idx += nthr) {
    int k = idx / H, f = idx - k * H;
    prow[a.q.offW[0] + idx] = gW0[k * HP + f];
  }
  for (int idx = tid; idx < L * H; idx += nthr) {
    int l = idx / H, f = idx - l * H;
    prow[a.q.offB[l] + f] = gB[l * HP + f];
  }
  for (int idx = tid; idx < H * m; idx += nthr) {
    int f = idx / m, cc = idx - f * m;
    prow[a.q.offW[L] + idx] = gWL[cc * HP + f];
  }
  for (int idx = tid; idx < m; idx += nthr) prow[a.q.offB[L] + idx] = gBL[idx];
  if (a.resident) {
    for (int l = 1; l < L; ++l) {
      const float* gw = gWh + (l - 1) * HP * HP;
      for (int idx = tid; idx < H * H; idx += nthr) {
        int in = idx / H, out = idx - in * H;
        prow[a.q.offW[l] + idx] = gw[in * HP + out];
      }
    }
  }
}

// ------------------------------------------------------------------------------------ host side
#include <stdio.h>
#include <string.h>
extern "C" void ppsci_set_error(const char* fmt, ...);

static int bwd_plan(const ppsci_mlp_desc& d, const ppsci_derived& q, int ntiles, int* resident, int* lds_bytes, int* grid, int* iters) {
  const long long HP2 = (long long)q.HP * q.HP;
  const long long small = bwd_small_floats(d, q);
  const long long res = small + 2LL * (d.n_hidden - 1) * HP2;
  *resident = (res * 4 <= PPSCI_LDS_LIMIT_BYTES - 1024) ? 1 : 0;
  const long long fl = *resident ? res : small + 2 * HP2;
  if (fl * 4 > PPSCI_LDS_LIMIT_BYTES) return PPSCI_E_UNSUPPORTED;
  *lds_bytes = (int)(fl * 4);
  const int blocks_needed = (ntiles + PPSCI_WAVES_PER_BLOCK - 1) / PPSCI_WAVES_PER_BLOCK;
  int per_cu = PPSCI_LDS_LIMIT_BYTES / (*lds_bytes + 256);
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 2) per_cu = 2;
  int gr = blocks_needed < 256 * per_cu ? blocks_needed : 256 * per_cu;
  if (ppsci_get_max_grid() > 0 && gr > ppsci_get_max_grid()) gr = ppsci_get_max_grid();
  if (gr < 1) gr = 1;
  *grid = gr;
  *iters = (blocks_needed + gr - 1) / gr;
  return PPSCI_OK;
}

extern "C" int64_t ppsci_bwd_partial_rows(const ppsci_mlp_desc* d, int64_t n_points) {
  ppsci_derived q;
  if (!d || n_points <= 0 || ppsci_derive(d, &q) != PPSCI_OK) return 0;
  int resident, lds, grid, iters;
  const int ntiles = (int)((n_points + PPSCI_TILE - 1) / PPSCI_TILE);
  if (bwd_plan(*d, q, ntiles, &resident, &lds, &grid, &iters) != PPSCI_OK) return 0;
  return grid;
}

template <int NB, int N1, int N2>
static int launch_bwd(BwdArgs& a, void* stream) {
  int lds, grid;
  if (bwd_plan(a.d, a.q, a.ntiles, &a.resident, &lds, &grid, &a.iters) != PPSCI_OK) {
    ppsci_set_error("taylor_bwd: LDS need exceeds %d B (width %d)", PPSCI_LDS_LIMIT_BYTES, a.d.width);
    return PPSCI_E_UNSUPPORTED;
  }
  if (PPSCI_SET_MAX_LDS((taylor_bwd_kernel<NB, N1, N2>), lds) != 0) {
    ppsci_set_error("taylor_bwd: cannot raise dynamic LDS to %d B", lds);
    return PPSCI_E_LAUNCH;
  }
  PPSCI_LAUNCH((taylor_bwd_kernel<NB, N1, N2>), BwdArgs, grid, PPSCI_BLOCK, lds, stream, a);
  int e = PPSCI_LAST_LAUNCH_ERROR();
  if (e != 0) {
    ppsci_set_error("taylor_bwd: launch failed (hip error %d)", e);
    return PPSCI_E_LAUNCH;
  }
  return PPSCI_OK;
}

#define PPSCI_BWD_CASE(NB_, N1_, N2_) \
  if (q.NB == NB_ && d->n1 == N1_ && d->n2 == N2_) return launch_bwd<NB_, N1_, N2_>(a, stream);

extern "C" int ppsci_taylor_bwd(const ppsci_mlp_desc* d, const float* params, int64_t n_points,
                                const float* const* inputs_host, const float* Ubar, const void* stash,
                                float* grad_partials, void* stream) {
  ppsci_derived q;
  if (!d || !params || !inputs_host || !Ubar || !stash || !grad_partials || n_points <= 0 ||
      ppsci_derive(d, &q) != PPSCI_OK) {
    ppsci_set_error("taylor_bwd: invalid argument");
    return PPSCI_E_INVALID;
  }
  BwdArgs a;
  memset(&a, 0, sizeof(a));
  a.d = *d;
  a.q = q;
  a.params = params;
  for (int j = 0; j < d->d_raw; ++j) a.x[j] = inputs_host[j];
  a.Ubar = Ubar;
  a.stash = (const f32x4*)stash;
  a.partials = grad_partials;
  a.N = n_points;
  a.ntiles = (int)((n_points + PPSCI_TILE - 1) / PPSCI_TILE);
  PPSCI_BWD_CASE(2, 0, 0) PPSCI_BWD_CASE(4, 0, 0) PPSCI_BWD_CASE(8, 0, 0)
  PPSCI_BWD_CASE(2, 1, 1) PPSCI_BWD_CASE(4, 1, 1) PPSCI_BWD_CASE(8, 1, 1)
  PPSCI_BWD_CASE(2, 2, 0) PPSCI_BWD_CASE(4, 2, 0) PPSCI_BWD_CASE(8, 2, 0)
  PPSCI_BWD_CASE(2, 2, 1) PPSCI_BWD_CASE(4, 2, 1) PPSCI_BWD_CASE(8, 2, 1)
  PPSCI_BWD_CASE(2, 2, 2) PPSCI_BWD_CASE(4, 2, 2) PPSCI_BWD_CASE(8, 2, 2)
  PPSCI_BWD_CASE(2, 3, 3) PPSCI_BWD_CASE(4, 3, 3)
  ppsci_set_error("taylor_bwd: unsupported (width=%d -> NB=%d, n1=%d, n2=%d)", d->width, q.NB, d->n1, d->n2);
  return PPSCI_E_UNSUPPORTED;
}
