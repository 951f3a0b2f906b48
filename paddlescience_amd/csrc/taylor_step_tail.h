// taylor_step_tail.h -- the tree of "last one out" reductions that ends a one-launch step kernel (taylor_step.inc: the
// single-wave kernels of padded width 32; taylor_fused.inc: the feature-split tile kernels of padded width 64): the
// workgroups' rows of partial sums -> gradient in the canonical parameter layout, loss terms, and (when asked) the Adam
// update, by the workgroup that finishes last.  See taylor_step.inc for the protocol.
#pragma once
#include "taylor_step.h"

// A row of partial sums has three segments: the hidden-weight blocks (per_tile floats, read as float4 columns), the small
// tensors (psmall floats) and the loss terms (n_res floats).  Level 0 keeps them in three arrays with their own row
// strides (what the reverse sweep and the epilogue write); the levels above in rows of `tree`.
struct StepSrc {
  const float *w, *s, *l;
  long long stw, sts, stl;
};

// Sums rows first .. first + nrows - 1 (nrows <= PPSCI_STEP_FAN) of `src`, column by column, in row order.
//   TOP == false: the sums become row `dst` of the next level (agent-scope stores: another workgroup reads them);
//   TOP == true : they are the totals -- each one is scattered to its place in the canonical parameter layout:
//                 grad (+)= total, the Adam update of that parameter, loss_terms.
// Per pass every thread takes two float4 columns of the hidden-weight blocks and one scalar column (small tensors, then
// loss terms) and requests ALL their rows -- and, TOP, the Adam operands of the parameters they map to -- before the
// first sum: one memory round trip (3-4 us: the rows come from memory, not from this XCD's L2) per pass, and one pass
// for a 3-hidden-layer net of padded width 32, instead of a round trip per row.
template <bool TOP>
__device__ __forceinline__ void ppsci_step_reduce(const StepTail& t, const ppsci_mlp_desc& d, const ppsci_derived& q,
                                                  const StepSrc& src, int first, int nrows, float* dst) {
  constexpr int FAN = PPSCI_STEP_FAN;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int ncw = t.per_tile >> 2, nsc = t.psmall + t.n_res;
  const int off_s = t.per_tile, off_l = t.per_tile + ((t.psmall + 3) & ~3);
  const int L = d.n_hidden, H = d.width, NB = q.NB, HP = q.HP, m = d.d_out, d0 = q.d0;
  for (int pass = 0; pass * 2 * nthr < ncw || pass * nthr < nsc; ++pass) {
    // ---- requests.  Slots 0, 1: float4 columns; slot 2: the scalar column (in .x).  A missing column repeats column 0.
    int col[3];
    col[0] = pass * 2 * nthr + tid, col[1] = col[0] + nthr, col[2] = pass * nthr + tid;
    const bool ok[3] = {col[0] < ncw, col[1] < ncw, col[2] < nsc};
    f32x4 v4[2][FAN];
    float v1[FAN];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float* p = src.w + 4LL * (ok[u] ? col[u] : 0) + first * src.stw;
#pragma unroll
      for (int r = 0; r < FAN; ++r) v4[u][r] = *(const f32x4*)(p + (long long)(r < nrows ? r : nrows - 1) * src.stw);
    }
    {
      const int c = ok[2] ? col[2] : 0;
      const bool small = c < t.psmall;
      const long long st = small ? src.sts : src.stl;
      const float* p = (small ? src.s + c : src.l + (c - t.psmall)) + first * st;
#pragma unroll
      for (int r = 0; r < FAN; ++r) v1[r] = p[(long long)(r < nrows ? r : nrows - 1) * st];
    }
    int pidx[3][4];  // TOP: parameter index of each of the column's (up to four) values; -1: padding; -2 - k: loss term k
    float og[3][4], om[3][4], ov[3][4], op[3][4];
    if (TOP) {
#pragma unroll
      for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) pidx[u][r] = -1;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (ok[u]) {
          // block (ib, ob), lane 16g + c, component r  <->  (in = 16ib + 4g + r, out = 16ob + c) of hidden matrix l
          const int e = 4 * col[u], l = 1 + e / (HP * HP), rem = e - (l - 1) * HP * HP;
          const int blk = rem >> 8, lane = (rem >> 2) & 63;
          const int in0 = 16 * (blk / NB) + 4 * (lane >> 4), out = 16 * (blk % NB) + (lane & 15);
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (in0 + r < H && out < H) pidx[u][r] = q.offW[l] + (in0 + r) * H + out;
        }
      }
      if (ok[2]) {
        const int ci = col[2];
        if (ci >= t.psmall) pidx[2][0] = -2 - (ci - t.psmall);
        // compact order of ppsci_small_params: W0 | b_0 .. b_{L-1} | W_last | b_last (| activation parameters)
        else if (ci < d0 * H) pidx[2][0] = q.offW[0] + ci;
        else if (ci < (d0 + L) * H) {
          const int lb = (ci - d0 * H) / H;
          pidx[2][0] = q.offB[lb] + (ci - d0 * H - lb * H);
        } else if (ci < (d0 + L + m) * H) pidx[2][0] = q.offW[L] + (ci - (d0 + L) * H);
        else pidx[2][0] = q.offB[L] + (ci - (d0 + L + m) * H);
      }
#pragma unroll
      for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int r = 0; r < (u < 2 ? 4 : 1); ++r) {
          const int i = pidx[u][r] >= 0 ? pidx[u][r] : 0;
          og[u][r] = t.accumulate ? t.grad[i] : 0.f;
          if (t.do_adam) om[u][r] = t.m[i], ov[u][r] = t.v[i], op[u][r] = t.p[i];
        }
    }
    // ---- sums in row order, then the consumers
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < FAN; ++r) {
        const f32x4 x = u < 2 ? v4[u < 2 ? u : 0][r] : (f32x4){v1[r], 0.f, 0.f, 0.f};
        acc += r < nrows ? x : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      if (ok[u]) {
        if (!TOP) {
          if (u < 2) ppsci_store_agent4((f32x4*)dst + col[u], acc);
          else if (col[2] < t.psmall) ppsci_store_agent(&dst[off_s + col[2]], acc[0]);
          else ppsci_store_agent(&dst[off_l + (col[2] - t.psmall)], acc[0]);
        } else {
#pragma unroll
          for (int r = 0; r < (u < 2 ? 4 : 1); ++r) {
            const int i = pidx[u][r];
            if (i <= -2) t.loss_terms[-2 - i] = acc[r];
            if (i >= 0) {
              float g = acc[r] + og[u][r];
              t.grad[i] = g;
              if (t.do_adam) {  // == adam_kernel (epilogue_optim.hip)
                g *= t.grad_scale;
                const float mm = t.beta1 * om[u][r] + (1.f - t.beta1) * g;
                const float vv = t.beta2 * ov[u][r] + (1.f - t.beta2) * g * g;
                t.m[i] = mm;
                t.v[i] = vv;
                t.p[i] = op[u][r] - t.lr_t * (mm / (sqrtf(vv) + t.eps_t));
              }
            }
          }
        }
      }
    }
  }
}

__device__ __forceinline__ void ppsci_step_tail(const StepTail& t, const ppsci_mlp_desc& d, const ppsci_derived& q,
                                                float* smem) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int off_s = t.per_tile, off_l = t.per_tile + ((t.psmall + 3) & ~3);
  StepSrc src{t.rows_w, t.rows_s, t.rows_l, t.per_tile, t.psmall, t.n_res};
  int idx = blockIdx.x, n = t.grid, row0 = 0, cnt0 = 0;  // first row / counter of the NEXT level
  for (;;) {
    const int g = idx / PPSCI_STEP_FAN;
    const int first = g * PPSCI_STEP_FAN;
    const int gsize = n - first < PPSCI_STEP_FAN ? n - first : PPSCI_STEP_FAN;
    const int n_next = (n + PPSCI_STEP_FAN - 1) / PPSCI_STEP_FAN;
    // release: this workgroup's row (level 0) / the row it has just summed was written with agent-scope stores; once
    // they have completed (the barrier waits for every wave's outstanding stores) the ticket may tell the others
    ppsci_block_sync_mem();
    if (n > 1) {
      if (tid == 0) ((unsigned*)smem)[0] = atomicAdd(&t.counters[cnt0 + g], 1u);
      __syncthreads();
      const unsigned ticket = ((const unsigned*)smem)[0];
      __syncthreads();
      if ((int)ticket != gsize - 1) return;  // not the last of the group: done
    }
    ppsci_acquire_agent();  // the other members' rows: nothing stale from this CU's / XCD's caches
    if (n_next == 1) {
      // the only group of the top level: its sums are the totals
      for (int k = tid; k < cnt0 + 1; k += nthr) t.counters[k] = 0u;  // ready for the next launch
      ppsci_step_reduce<true>(t, d, q, src, first, gsize, nullptr);
      return;
    }
    float* dst = t.tree + (long long)(row0 + g) * t.rowlen;
    ppsci_step_reduce<false>(t, d, q, src, first, gsize, dst);
    if (t.external == 2) {  // one level only: the kernel behind this launch sums the level-1 rows
      if (tid == 0) t.counters[cnt0 + g] = 0u;  // (every member of the group has taken its ticket: ready for the next launch)
      return;
    }
    const float* base = t.tree + (long long)row0 * t.rowlen;
    src = StepSrc{base, base + off_s, base + off_l, t.rowlen, t.rowlen, t.rowlen};
    idx = g;
    n = n_next;
    row0 += n_next;
    cnt0 += n_next;
  }
}

