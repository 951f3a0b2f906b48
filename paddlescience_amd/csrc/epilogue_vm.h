// epilogue_vm.h -- the pointwise epilogue VM (+ fused loss terms and their adjoints) as a device function, shared by
// the stand-alone epilogue kernel (epilogue_optim.hip) and the one-launch step kernel (taylor_step.inc).
//
// Replaces the chain of tiny elementwise kernels the reference launches for
//   OperatorNode / ConstantNode / DetachNode   /root/reference/ppsci/utils/symbolic.py:184-267,433-468,165-181
//   AllenCahn closure body                     /root/reference/ppsci/equation/pde/allen_cahn.py:62
//   MSELoss.forward                            /root/reference/ppsci/loss/mse.py:82-105
// and the seed of total_loss.backward() (train.py:158) for the pointwise part.
// One lane = one collocation point; all HBM traffic is coalesced SoA ([row][N] arrays).  The
// program is uniform across lanes (no divergence).
#pragma once
#include "taylor_tile.h"

#define EPI_BLOCK 256
#define EPI_LDS_PROG 31  // longest program whose VM register file is kept in LDS (2 x 31 KiB + 1 KiB per workgroup: under the 64 KiB
                         // that need no launch attribute; 2 workgroups per CU): covers NavierStokes 2-D (30 instructions)

struct EpiArgs {
  ppsci_epilogue_desc e;
  const float* x[PPSCI_MAX_IN];
  const float* aux[PPSCI_MAX_AUX];
  const float* U;
  float* resid;     // may be null: [n_res, N]
  float* Ubar;      // may be null: [n_streams, N]
  float* partials;  // [gridDim.x, n_res]
  const float* ep;  // [PPSCI_MAX_EPARAM] learnable equation parameters (may be null)
  float* ep_part;   // [gridDim.x, PPSCI_MAX_EPARAM] (may be null)
  long long N;
  int iters;
  int ntiles;  // EPI_RF_TILED only
  float* loss_out;     // may be null; else [n_res]: the loss terms, summed over all workgroups by the one that finishes last
  unsigned* counter;   // with loss_out: one zero-initialised ticket counter (left at zero)
  int nload;   // the program's load instructions (LD_IN / LD_U / LD_AUX), in program order: epi_fill_loads()
  unsigned char load_idx[PPSCI_MAX_PROG];
  // EPI_RF_FUSED: the pre-decoded arithmetic steps of a program that only adds, subtracts, multiplies, negates and
  // detaches (every BASELINE PDE residual) -- epi_fast_encode(); device memory; null: the generic interpreter
  const unsigned* fast;
  int nfast;
  int nfast_loads;  // entries of the load table: the program's loads AND constants, in program order
  int static_id;    // > 0: the program IS table `static_id` of epi_static_programs.h (epi_static.h): evaluated as compile-time
                    // straight-line code by every wave; `fast` still holds the tables (constants, scales)
};

// ---- pre-decoded programs for epi_point_fast.  One 32-bit word per arithmetic step (loads and constants are not steps):
//   [0:6] a  [7:13] b  [14:20] destination i  [21] product  [22:23] sx  [24:25] sy  [26:27] rx  [28:29] ry  [30] a == b
// value:   r = product ? v[a] v[b] : sx v[a] + sy v[b];   adjoints: abar += g (product ? v[b] : rx), bbar += g (product ? v[a] : ry)
// with the two-bit codes 0 -> 0, 1 -> +1, 2 -> -1 (detach: sx = 1, rx = 0).
static inline unsigned epi_fast_word(int a, int b, int i, int prod, int sx, int sy, int rx, int ry) {
  return (unsigned)a | ((unsigned)b << 7) | ((unsigned)i << 14) | ((unsigned)prod << 21) | ((unsigned)sx << 22) |
         ((unsigned)sy << 24) | ((unsigned)rx << 26) | ((unsigned)ry << 28) | ((a == b ? 1u : 0u) << 30);
}
// The device buffer of a pre-decoded program (EPI_FAST_WORDS dwords): [0, 64) the arithmetic steps, [64, 128) the loads
// and constants --
//   [0:6] destination i  [7:13] index a (input / stream / aux array)  [14:15] 0 input, 1 stream, 2 aux, 3 constant --
// [128, 136) the loss terms: [0:6] value  [7:11] label + 1  [12:16] weight + 1  [17:21] area + 1  (0: none), and
// [192, 256) the constants' values (entry k belongs to load-table entry k).
// A wave keeps the groups in four registers, entry k in lane k, and fetches an entry with v_readlane: no memory access
// (a scalar load per step from the argument block cost ~300 cycles, more than the step itself).
#define EPI_FAST_MAX 64
#define EPI_FAST_WORDS (64 + 64 + 64 + 64)
#define EPI_FAST_NREG 16  // programs of at most this many instructions keep the VM's register file in VGPRs (epi_tile_fast_regs)
// host: fills `out` (EPI_FAST_WORDS dwords), *n_loads (entries of the load table) and returns the number of arithmetic
// steps (<= EPI_FAST_MAX), or -1 when the program (or one of its loss terms) needs the generic interpreter
static inline int epi_fast_encode(const ppsci_epilogue_desc& e, unsigned* out, int* n_loads) {
  if (e.n_instr > 127) return -1;
  {
    for (int k = 0; k < EPI_FAST_WORDS; ++k) out[k] = 0u;
    int nl = 0;
    for (int i = 0; i < e.n_instr; ++i) {
      const ppsci_instr& ins = e.prog[i];
      const int kind = ins.op == PPSCI_OP_LD_IN ? 0 : (ins.op == PPSCI_OP_LD_U ? 1 : (ins.op == PPSCI_OP_LD_AUX ? 2 : (ins.op == PPSCI_OP_CONST ? 3 : -1)));
      if (kind < 0) continue;
      if (nl == 64 || ins.a > 127 || ins.a < 0) return -1;
      union {
        float f;
        unsigned u;
      } cb;
      cb.f = kind == 3 ? ins.c : 0.f;
      const unsigned cbits = cb.u;
      out[192 + nl] = cbits;
      out[64 + nl++] = (unsigned)i | ((unsigned)(kind == 3 ? 0 : ins.a) << 7) | ((unsigned)kind << 14);
    }
    *n_loads = nl;
    for (int k = 0; k < e.n_res; ++k) {
      const ppsci_residual& r = e.res[k];
      if (r.label > 30 || r.weight > 30 || r.area > 30) return -1;
      out[128 + k] = (unsigned)r.value | ((unsigned)(r.label + 1) << 7) | ((unsigned)(r.weight + 1) << 12) | ((unsigned)(r.area + 1) << 17);
    }
  }
  for (int k = 0; k < e.n_res; ++k)
    if (e.res[k].kind != PPSCI_LOSS_MSE || e.res[k].scale_param != 0) return -1;
  int n = 0;
  for (int i = 0; i < e.n_instr; ++i) {
    const ppsci_instr& ins = e.prog[i];
    unsigned w;
    switch (ins.op) {
      case PPSCI_OP_LD_IN:
      case PPSCI_OP_LD_U:
      case PPSCI_OP_LD_AUX:
      case PPSCI_OP_CONST: continue;
      case PPSCI_OP_ADD: w = epi_fast_word(ins.a, ins.b, i, 0, 1, 1, 1, 1); break;
      case PPSCI_OP_SUB: w = epi_fast_word(ins.a, ins.b, i, 0, 1, 2, 1, 2); break;
      case PPSCI_OP_MUL: w = epi_fast_word(ins.a, ins.b, i, 1, 0, 0, 0, 0); break;
      case PPSCI_OP_NEG: w = epi_fast_word(ins.a, ins.a, i, 0, 2, 0, 2, 0); break;     // (a == b: one accumulation of -g)
      case PPSCI_OP_DETACH: w = epi_fast_word(ins.a, ins.a, i, 0, 1, 0, 0, 0); break;  // (no adjoint flows)
      default: return -1;
    }
    if (n == EPI_FAST_MAX) return -1;
    out[n++] = w;
  }
  return n;
}

// host: list the load instructions of a.e (after a.e is set)
static inline void epi_fill_loads(EpiArgs& a) {
  a.nload = 0;
  for (int i = 0; i < a.e.n_instr; ++i) {
    const int op = a.e.prog[i].op;
    if (op == PPSCI_OP_LD_IN || op == PPSCI_OP_LD_U || op == PPSCI_OP_LD_AUX) a.load_idx[a.nload++] = (unsigned char)i;
  }
}

// d/dx lgamma(x) (the adjoint of paddle.lgamma): reflection for x < 0.5, recurrence up to x >= 6, then the
// asymptotic series ln x - 1/(2x) - 1/(12x^2) + 1/(120x^4) - 1/(252x^6)  (truncation < 1e-8 at x = 6).
__device__ __forceinline__ float epi_digamma(float x) {
  float refl = 0.f;
  if (x < 0.5f) {
    refl = -3.14159265358979f / tanf(3.14159265358979f * x);
    x = 1.f - x;
  }
  float acc = 0.f;
  while (x < 6.f) {
    acc -= 1.f / x;
    x += 1.f;
  }
  const float i1 = 1.f / x, i2 = i1 * i1;
  return refl + acc + logf(x) - 0.5f * i1 - i2 * (1.f / 12.f - i2 * (1.f / 120.f - i2 * (1.f / 252.f)));
}

// The VM's register file (values and adjoints of the n instructions), by MODE:
//   EPI_RF_SCRATCH  per-lane scratch memory (dynamic indexing: 1 040 bytes per lane, every access a scratch round trip)
//   EPI_RF_LDS      LDS, [n][EPI_BLOCK] floats each (conflict-free: lane-consecutive), for programs of at most
//                   EPI_LDS_PROG instructions -- the pointwise programs of the BASELINE PDEs (Laplace 3, Allen-Cahn 12, NavierStokes 2-D 30)
//   EPI_RF_TILED    inside the one-launch step kernel (taylor_step.inc): wave w runs the program for the 16 points of
//                   ITS tile on lanes 0..15 (the tile the same wave's forward sweep has just produced and its reverse
//                   sweep consumes next); register file in LDS, [n][16 * waves]
// `red`: EPI_BLOCK floats of LDS for the loss reductions, followed by the register file (LDS modes).
//   EPI_RF_FUSED    inside the fused tile kernel (taylor_fused.inc): ONE 16-point tile per workgroup; lanes 0..15 of wave 0
//                   run the program for it.  The tile's operands never leave the CU: LD_U sums the waves' partial sums
//                   of the last linear layer from LDS, LD_IN reads the tile's inputs from LDS, and the adjoint of every
//                   LD_U goes to the LDS rows the reverse sweep reads (EpiFused); register file in LDS, [n][16]
#define EPI_RF_SCRATCH 0
#define EPI_RF_LDS 1
#define EPI_RF_TILED 2
#define EPI_RF_FUSED 3

// -DPPSCI_FUSED_TIMERS (measurement builds only): cycle stamps inside the pre-decoded program, summed per phase in LDS
#ifdef PPSCI_FUSED_TIMERS
#define EPI_FT(k)                                                         \
  if (threadIdx.x == 0) {                                                 \
    const unsigned now_ = (unsigned)__builtin_amdgcn_s_memtime();          \
    fx->dbg[k] += now_ - fx->dbg[15];                                     \
    fx->dbg[15] = now_;                                                   \
  }
#else
#define EPI_FT(k)
#endif

struct EpiFused {
  unsigned* dbg;      // PPSCI_FUSED_TIMERS: 16 counters in LDS ([15]: the last stamp)
  const float* red;   // [W][mS][16]: per-wave partial sums of the last linear layer (fixed-order sum over w)
  const float* bl;    // [m]: last bias (added to the value stream, s == 0)
  const float* tinx;  // [d_raw][16]: the tile's raw inputs
  float* tin;         // [mS][16]: dL/dU of the tile (rows the program never loads are zeroed here)
  float* rres;        // epi_tile_fast: [n_res][16] residual values of the tile (another wave writes them out); may be null
  int W, mS, S;
};

// One point of the program: forward (memory operands first), loss terms and their seeds, reverse.
//   vp / ap: this point's column of the register file (values / adjoints), row stride RS; n = program length;
//   p / pp / valid: the point, its clamped index for loads, inside the batch; pt: the point's index in its tile (FUSED).
template <int MODE>
__device__ __forceinline__ void epi_point(const EpiArgs& a, float* const vp, float* const ap, const int RS, const int n,
                                          const long long p, const long long pp, const bool valid, const int pt,
                                          float (&lsum)[PPSCI_MAX_RES], float (&padj)[PPSCI_MAX_EPARAM], const EpiFused* fx) {
    // labels / weights / areas of the first two loss terms: requested now, used behind the forward pass (they come
    // from HBM -- a round trip of their own when requested where they are used).  Named scalars: an array indexed by
    // the term number ends up in scratch memory.
#define EPI_PF(k_, L_, W_, A_)                                                   \
  float L_ = 0.f, W_ = 0.f, A_ = 0.f;                                            \
  if (k_ < a.e.n_res) {                                                          \
    const ppsci_residual rs_ = a.e.res[k_];                                      \
    if (rs_.label >= 0) L_ = a.aux[rs_.label][pp];                              \
    if (rs_.weight >= 0) W_ = a.aux[rs_.weight][pp];                             \
    if (rs_.area >= 0) A_ = a.aux[rs_.area][pp];                                 \
  }
    EPI_PF(0, pfl0, pfw0, pfa0)
    EPI_PF(1, pfl1, pfw1, pfa1)
#undef EPI_PF
    // ---- forward, pass 1: every memory operand of the program, eight loads in flight at a time, straight into the
    // register file (issued one per VM step, each load is a full round trip that the next instruction waits for:
    // 5-6 us for the eight loads of a Laplace program with label and weight, against well under 1 us of arithmetic)
    if (MODE == EPI_RF_FUSED) {
      // the tile's streams and inputs are in LDS; only LD_AUX goes to memory
      for (int q = pt; q < fx->mS * PPSCI_TILE; q += PPSCI_TILE) fx->tin[q] = 0.f;
      for (int k = 0; k < a.nload; ++k) {
        const int i = a.load_idx[k];
        const ppsci_instr ins = a.e.prog[i];
        float v;
        if (ins.op == PPSCI_OP_LD_U) {
          v = (ins.a % fx->S == 0) ? fx->bl[ins.a / fx->S] : 0.f;
          for (int w = 0; w < fx->W; ++w) v += fx->red[(w * fx->mS + ins.a) * PPSCI_TILE + pt];
        } else if (ins.op == PPSCI_OP_LD_IN) v = fx->tinx[ins.a * PPSCI_TILE + pt];
        else v = a.aux[ins.a][pp];
        vp[i * RS] = v;
      }
    } else {
    for (int base = 0; base < a.nload; base += 8) {
      float tv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int i = a.load_idx[base + k < a.nload ? base + k : a.nload - 1];
        const ppsci_instr ins = a.e.prog[i];
        const float* src = ins.op == PPSCI_OP_LD_IN ? a.x[ins.a] : (ins.op == PPSCI_OP_LD_U ? a.U + (long long)ins.a * a.N : a.aux[ins.a]);
        tv[k] = src[pp];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (base + k < a.nload) vp[(a.load_idx[base + k]) * RS] = tv[k];
    }
    }
    // ---- forward, pass 2
    for (int i = 0; i < n; ++i) {
      const ppsci_instr ins = a.e.prog[i];
      float r;
      switch (ins.op) {
        case PPSCI_OP_LD_IN:
        case PPSCI_OP_LD_U:
        case PPSCI_OP_LD_AUX: r = vp[(i) * RS]; break;  // pass 1
        case PPSCI_OP_CONST: r = ins.c; break;
        case PPSCI_OP_LD_PARAM: r = a.ep[ins.a]; break;
        case PPSCI_OP_ADD: r = vp[(ins.a) * RS] + vp[(ins.b) * RS]; break;
        case PPSCI_OP_SUB: r = vp[(ins.a) * RS] - vp[(ins.b) * RS]; break;
        case PPSCI_OP_MUL: r = vp[(ins.a) * RS] * vp[(ins.b) * RS]; break;
        case PPSCI_OP_DIV: r = vp[(ins.a) * RS] / vp[(ins.b) * RS]; break;
        case PPSCI_OP_NEG: r = -vp[(ins.a) * RS]; break;
        case PPSCI_OP_POW: r = powf(vp[(ins.a) * RS], vp[(ins.b) * RS]); break;
        case PPSCI_OP_SIN: r = sinf(vp[(ins.a) * RS]); break;
        case PPSCI_OP_COS: r = cosf(vp[(ins.a) * RS]); break;
        case PPSCI_OP_TANH: r = tanhf(vp[(ins.a) * RS]); break;
        case PPSCI_OP_EXP: r = expf(vp[(ins.a) * RS]); break;
        case PPSCI_OP_LOG: r = logf(vp[(ins.a) * RS]); break;
        case PPSCI_OP_SQRT: r = sqrtf(vp[(ins.a) * RS]); break;
        case PPSCI_OP_ABS: r = fabsf(vp[(ins.a) * RS]); break;
        case PPSCI_OP_SINH: r = sinhf(vp[(ins.a) * RS]); break;
        case PPSCI_OP_COSH: r = coshf(vp[(ins.a) * RS]); break;
        case PPSCI_OP_TAN: r = tanf(vp[(ins.a) * RS]); break;
        case PPSCI_OP_MAX: r = fmaxf(vp[(ins.a) * RS], vp[(ins.b) * RS]); break;
        case PPSCI_OP_MIN: r = fminf(vp[(ins.a) * RS], vp[(ins.b) * RS]); break;
        case PPSCI_OP_SIGN: r = (vp[(ins.a) * RS] > 0.f) ? 1.f : ((vp[(ins.a) * RS] < 0.f) ? -1.f : 0.f); break;
        case PPSCI_OP_HEAVISIDE: r = (vp[(ins.a) * RS] > 0.f) ? 1.f : 0.f; break;  // heaviside(x, y=0)
        case PPSCI_OP_DETACH: r = vp[(ins.a) * RS]; break;
        case PPSCI_OP_ASIN: r = asinf(vp[(ins.a) * RS]); break;
        case PPSCI_OP_ACOS: r = acosf(vp[(ins.a) * RS]); break;
        case PPSCI_OP_ATAN: r = atanf(vp[(ins.a) * RS]); break;
        case PPSCI_OP_ATAN2: r = atan2f(vp[(ins.a) * RS], vp[(ins.b) * RS]); break;
        case PPSCI_OP_ASINH: r = asinhf(vp[(ins.a) * RS]); break;
        case PPSCI_OP_ACOSH: r = acoshf(vp[(ins.a) * RS]); break;
        case PPSCI_OP_ATANH: r = atanhf(vp[(ins.a) * RS]); break;
        case PPSCI_OP_ERF: r = erff(vp[(ins.a) * RS]); break;
        case PPSCI_OP_LGAMMA: r = lgammaf(vp[(ins.a) * RS]); break;
        case PPSCI_OP_CEIL: r = ceilf(vp[(ins.a) * RS]); break;
        case PPSCI_OP_FLOOR: r = floorf(vp[(ins.a) * RS]); break;
        default: r = 0.f; break;
      }
      vp[(i) * RS] = r;
      ap[(i) * RS] = 0.f;
    }
    // ---- residuals, loss terms and their seeds
    for (int k = 0; k < a.e.n_res; ++k) {
      const ppsci_residual rs = a.e.res[k];
      float labl, wgtl, arel;
      if (k == 0) labl = pfl0, wgtl = pfw0, arel = pfa0;
      else if (k == 1) labl = pfl1, wgtl = pfw1, arel = pfa1;
      else {  // label, weight and area in ONE round trip: unconditional loads from clamped rows, selected afterwards
        labl = a.aux[rs.label >= 0 ? rs.label : 0] != nullptr ? a.aux[rs.label >= 0 ? rs.label : 0][pp] : 0.f;
        wgtl = a.aux[rs.weight >= 0 ? rs.weight : 0] != nullptr ? a.aux[rs.weight >= 0 ? rs.weight : 0][pp] : 0.f;
        arel = a.aux[rs.area >= 0 ? rs.area : 0] != nullptr ? a.aux[rs.area >= 0 ? rs.area : 0][pp] : 0.f;
      }
      const float rv = vp[(rs.value) * RS];
      if (a.resid != nullptr && valid) a.resid[(long long)k * a.N + p] = rv;
      const float lab = (rs.label >= 0) ? labl : 0.f;
      float w = 1.f;
      if (rs.weight >= 0) w *= wgtl;
      if (rs.area >= 0 && rs.kind != PPSCI_LOSS_ABSREL) w *= arel;
      const float diff = rv - lab;
      if (valid) {
        // (scale_param: a multiplier that lives on the device -- the adjoint of a batch reduction, ppsci_hip.h)
        const float tscale = rs.scale_param > 0 ? rs.scale * a.ep[PPSCI_MAX_EPARAM + rs.scale_param - 1] : rs.scale;
        if (rs.kind == PPSCI_LOSS_MSE) {
          w *= tscale;
          lsum[k] += w * diff * diff;
          ap[(rs.value) * RS] += 2.f * w * diff;
        } else if (rs.kind == PPSCI_LOSS_LINEAR) {
          w *= tscale;
          lsum[k] += w * diff;
          ap[(rs.value) * RS] += w;
        } else {
          float f = tscale * (rs.kind == PPSCI_LOSS_SQRTABS ? sqrtf(w) : w);
          if (rs.kind == PPSCI_LOSS_ABSREL) f /= fabsf(lab);
          lsum[k] += f * fabsf(diff);
          ap[(rs.value) * RS] += diff > 0.f ? f : (diff < 0.f ? -f : 0.f);
        }
      }
    }
    // ---- reverse
    if (MODE == EPI_RF_FUSED || a.Ubar != nullptr) {
      for (int i = n - 1; i >= 0; --i) {
        const ppsci_instr ins = a.e.prog[i];
        const float g = ap[(i) * RS];
        switch (ins.op) {
          case PPSCI_OP_LD_U:
            if (MODE == EPI_RF_FUSED) {
              fx->tin[ins.a * PPSCI_TILE + pt] = g;  // (invalid lanes carry g = 0: their seeds are never set)
              if (valid && a.Ubar != nullptr) a.Ubar[(long long)ins.a * a.N + p] = g;
            } else if (valid) a.Ubar[(long long)ins.a * a.N + p] = g;
            break;
          case PPSCI_OP_LD_PARAM:
#pragma unroll
            for (int k = 0; k < PPSCI_MAX_EPARAM; ++k)
              if (ins.a == k) padj[k] += g;  // invalid lanes carry g = 0 (their seeds are never set)
            break;
          case PPSCI_OP_ADD: ap[(ins.a) * RS] += g; ap[(ins.b) * RS] += g; break;
          case PPSCI_OP_SUB: ap[(ins.a) * RS] += g; ap[(ins.b) * RS] -= g; break;
          case PPSCI_OP_MUL: ap[(ins.a) * RS] += g * vp[(ins.b) * RS]; ap[(ins.b) * RS] += g * vp[(ins.a) * RS]; break;
          case PPSCI_OP_DIV: {
            const float inv = 1.f / vp[(ins.b) * RS];
            ap[(ins.a) * RS] += g * inv;
            ap[(ins.b) * RS] -= g * vp[(i) * RS] * inv;
          } break;
          case PPSCI_OP_NEG: ap[(ins.a) * RS] -= g; break;
          case PPSCI_OP_POW: {
            const float x = vp[(ins.a) * RS], y = vp[(ins.b) * RS];
            ap[(ins.a) * RS] += g * y * powf(x, y - 1.f);
            if (x > 0.f) ap[(ins.b) * RS] += g * vp[(i) * RS] * logf(x);
          } break;
          case PPSCI_OP_SIN: ap[(ins.a) * RS] += g * cosf(vp[(ins.a) * RS]); break;
          case PPSCI_OP_COS: ap[(ins.a) * RS] -= g * sinf(vp[(ins.a) * RS]); break;
          case PPSCI_OP_TANH: ap[(ins.a) * RS] += g * (1.f - vp[(i) * RS] * vp[(i) * RS]); break;
          case PPSCI_OP_EXP: ap[(ins.a) * RS] += g * vp[(i) * RS]; break;
          case PPSCI_OP_LOG: ap[(ins.a) * RS] += g / vp[(ins.a) * RS]; break;
          case PPSCI_OP_SQRT: ap[(ins.a) * RS] += g * 0.5f / vp[(i) * RS]; break;
          case PPSCI_OP_ABS: ap[(ins.a) * RS] += g * ((vp[(ins.a) * RS] > 0.f) ? 1.f : ((vp[(ins.a) * RS] < 0.f) ? -1.f : 0.f)); break;
          case PPSCI_OP_SINH: ap[(ins.a) * RS] += g * coshf(vp[(ins.a) * RS]); break;
          case PPSCI_OP_COSH: ap[(ins.a) * RS] += g * sinhf(vp[(ins.a) * RS]); break;
          case PPSCI_OP_TAN: ap[(ins.a) * RS] += g * (1.f + vp[(i) * RS] * vp[(i) * RS]); break;
          case PPSCI_OP_MAX:
            if (vp[(ins.a) * RS] >= vp[(ins.b) * RS]) ap[(ins.a) * RS] += g; else ap[(ins.b) * RS] += g;
            break;
          case PPSCI_OP_MIN:
            if (vp[(ins.a) * RS] <= vp[(ins.b) * RS]) ap[(ins.a) * RS] += g; else ap[(ins.b) * RS] += g;
            break;
          case PPSCI_OP_ASIN: ap[(ins.a) * RS] += g / sqrtf(1.f - vp[(ins.a) * RS] * vp[(ins.a) * RS]); break;
          case PPSCI_OP_ACOS: ap[(ins.a) * RS] -= g / sqrtf(1.f - vp[(ins.a) * RS] * vp[(ins.a) * RS]); break;
          case PPSCI_OP_ATAN: ap[(ins.a) * RS] += g / (1.f + vp[(ins.a) * RS] * vp[(ins.a) * RS]); break;
          case PPSCI_OP_ATAN2: {
            const float y = vp[(ins.a) * RS], x = vp[(ins.b) * RS], inv = 1.f / (x * x + y * y);
            ap[(ins.a) * RS] += g * x * inv;
            ap[(ins.b) * RS] -= g * y * inv;
          } break;
          case PPSCI_OP_ASINH: ap[(ins.a) * RS] += g / sqrtf(vp[(ins.a) * RS] * vp[(ins.a) * RS] + 1.f); break;
          case PPSCI_OP_ACOSH: ap[(ins.a) * RS] += g / sqrtf(vp[(ins.a) * RS] * vp[(ins.a) * RS] - 1.f); break;
          case PPSCI_OP_ATANH: ap[(ins.a) * RS] += g / (1.f - vp[(ins.a) * RS] * vp[(ins.a) * RS]); break;
          case PPSCI_OP_ERF: ap[(ins.a) * RS] += g * 1.1283791670955126f * expf(-vp[(ins.a) * RS] * vp[(ins.a) * RS]); break;
          case PPSCI_OP_LGAMMA: ap[(ins.a) * RS] += g * epi_digamma(vp[(ins.a) * RS]); break;
          default: break;  // LD_IN, LD_AUX, CONST, SIGN, HEAVISIDE, DETACH, CEIL, FLOOR: no adjoint flows
        }
      }
    }
}

// EPI_RF_FUSED, pre-decoded programs (EpiArgs::fast): the same arithmetic as epi_point, step for step and in the same
// order, without the interpreter -- no opcode dispatch, every step a handful of branch-free VALU instructions between one
// round of LDS reads and its writes.  (The interpreter costs ~600 cycles per step on the 16 lanes it runs on, with the
// workgroup's other waves parked at the barrier behind it: 17 000 of a tile's 52 000 cycles on Allen-Cahn's 12
// instructions.)  The constants of the program sit in the register file from the start of the kernel (epi_fast_init).
__device__ __forceinline__ float epi_fast_code(unsigned c) { return (float)(c & 1u) - (float)(c >> 1); }

// the program's tables, entry k in lane k (loaded once per wave at the start of the kernel)
struct EpiFastRegs {
  unsigned steps, loads, terms;
  float scale;  // lane k: res[k].scale (the argument block's value: ppsci_taylor_step_plan_set_scales changes it)
  float cvals;  // lane k: the value of load-table entry k when it is a constant
};

// kernel start: the tables go to LDS (`tab`: EPI_FAST_WORDS + PPSCI_MAX_RES dwords), the constants into the register file
__device__ __forceinline__ void epi_fast_init(const EpiArgs& a, unsigned* const tab, float* const vp, const int RS, const int tid,
                                              const int nthr, const bool owner) {
  for (int k = tid; k < EPI_FAST_WORDS; k += nthr) tab[k] = a.fast[k];
  for (int k = tid; k < PPSCI_MAX_RES; k += nthr) tab[EPI_FAST_WORDS + k] = __builtin_bit_cast(unsigned, a.e.res[k].scale);
  if (owner) {  // the constants sit in the register file for the whole launch
    for (int i = 0; i < a.e.n_instr; ++i) {
      const ppsci_instr ins = a.e.prog[i];
      if (ins.op == PPSCI_OP_CONST) vp[i * RS] = ins.c;
    }
  }
}
// per tile: the wave that runs the program takes the tables into three registers (one LDS round trip)
__device__ __forceinline__ EpiFastRegs epi_fast_regs(const unsigned* const tab, const int lane) {
  EpiFastRegs R;
  R.steps = tab[lane & 63];
  R.loads = tab[64 + (lane & 63)];
  R.terms = tab[128 + (lane & 63)];
  R.scale = __builtin_bit_cast(float, tab[EPI_FAST_WORDS + (lane & (PPSCI_MAX_RES - 1))]);
  R.cvals = __builtin_bit_cast(float, tab[192 + (lane & 63)]);
  return R;
}

// One tile; called by ALL lanes of the wave (the table look-ups are wave-wide), `act`: this lane runs point `pt`.
// No global stores: dL/dU stays in fx->tin, the residual values go to fx->rres; the caller writes them out from another
// wave (a store here would put an `s_waitcnt vmcnt(0)` -- a full HBM write round trip -- into the step loops).
// `lacc`: this point's column of the running loss sums in LDS ([PPSCI_MAX_RES][PPSCI_TILE], stride PPSCI_TILE).
__device__ __forceinline__ void epi_tile_fast(const EpiArgs& a, const EpiFastRegs& R, float* const vp, float* const ap, const int RS,
                                              const bool act, const long long p, const long long pp, const bool valid,
                                              const int pt, float* const lacc, const EpiFused* fx) {
  // label / weight / area of the first two loss terms: requested first (HBM), used behind the forward steps
#define EPI_PF(k_, L_, W_)                                                             \
  float L_ = 0.f, W_ = 1.f;                                                            \
  if (k_ < a.e.n_res) {                                                                \
    const unsigned t_ = ppsci_readlane(R.terms, k_);                                   \
    const int lb_ = (int)((t_ >> 7) & 31u) - 1, wt_ = (int)((t_ >> 12) & 31u) - 1, ar_ = (int)((t_ >> 17) & 31u) - 1; \
    if (act) {                                                                         \
      if (lb_ >= 0) L_ = a.aux[lb_][pp];                                               \
      if (wt_ >= 0) W_ = a.aux[wt_][pp];                                               \
      if (ar_ >= 0) W_ *= a.aux[ar_][pp];                                              \
    }                                                                                  \
  }
  EPI_PF(0, lab0, w0)
  EPI_PF(1, lab1, w1)
#undef EPI_PF
  // ---- memory operands; the adjoint of everything the reverse pass accumulates into starts at zero
  if (act)
    for (int q = pt; q < fx->mS * PPSCI_TILE; q += PPSCI_TILE) fx->tin[q] = 0.f;
  for (int k = 0; k < a.nfast_loads; ++k) {
    const unsigned w = ppsci_readlane(R.loads, k);
    const int i = w & 127u, ia = (w >> 7) & 127u, kind = (w >> 14) & 3u;
    if (act && kind != 3) {  // (the constants sit in the LDS register file since epi_fast_init)
      float v;
      if (kind == 1) {
        v = (ia % fx->S == 0) ? fx->bl[ia / fx->S] : 0.f;
        for (int wv = 0; wv < fx->W; ++wv) v += fx->red[(wv * fx->mS + ia) * PPSCI_TILE + pt];
      } else if (kind == 0) v = fx->tinx[ia * PPSCI_TILE + pt];
      else v = a.aux[ia][pp];
      vp[i * RS] = v;
      ap[i * RS] = 0.f;
    }
  }
  // ---- forward steps
  for (int k = 0; k < a.nfast; ++k) {
    const unsigned w = ppsci_readlane(R.steps, k);
    const int ia = w & 127u, ib = (w >> 7) & 127u, id = (w >> 14) & 127u;
    if (act) {
      const float x = vp[ia * RS], y = vp[ib * RS];
      const float lin = epi_fast_code((w >> 22) & 3u) * x + epi_fast_code((w >> 24) & 3u) * y;
      vp[id * RS] = ((w >> 21) & 1u) ? x * y : lin;
      ap[id * RS] = 0.f;
    }
  }
  // ---- residuals, MSE terms and their seeds (epi_point, PPSCI_LOSS_MSE)
  for (int k = 0; k < a.e.n_res; ++k) {
    const unsigned t = ppsci_readlane(R.terms, k);
    const float scale = __builtin_bit_cast(float, ppsci_readlane(__builtin_bit_cast(unsigned, R.scale), k));
    const int iv = t & 127u, lb = (int)((t >> 7) & 31u) - 1, wt = (int)((t >> 12) & 31u) - 1, ar = (int)((t >> 17) & 31u) - 1;
    if (act) {
      float lab, wk;
      if (k == 0) lab = lab0, wk = w0;
      else if (k == 1) lab = lab1, wk = w1;
      else {
        lab = lb >= 0 ? a.aux[lb][pp] : 0.f;
        wk = wt >= 0 ? a.aux[wt][pp] : 1.f;
        if (ar >= 0) wk *= a.aux[ar][pp];
      }
      const float rv = vp[iv * RS];
      if (fx->rres != nullptr) fx->rres[k * PPSCI_TILE + pt] = rv;
      const float diff = rv - lab;
      if (valid) {
        const float wgt = wk * scale;
        lacc[k * PPSCI_TILE] += wgt * diff * diff;
        ap[iv * RS] += 2.f * wgt * diff;
      }
    }
  }
  // ---- reverse steps
  for (int k = a.nfast - 1; k >= 0; --k) {
    const unsigned w = ppsci_readlane(R.steps, k);
    const int ia = w & 127u, ib = (w >> 7) & 127u, id = (w >> 14) & 127u;
    if (act) {
      const float g = ap[id * RS], x = vp[ia * RS], y = vp[ib * RS];
      const float ga = ap[ia * RS], gb = ap[ib * RS];
      const bool prod = (w >> 21) & 1u;
      const float ca = prod ? y : epi_fast_code((w >> 26) & 3u), cb = prod ? x : epi_fast_code((w >> 28) & 3u);
      if ((w >> 30) & 1u) ap[ia * RS] = ga + g * ca + g * cb;  // a == b (u * u, -u): ONE accumulation, in the generic order
      else {
        ap[ia * RS] = ga + g * ca;
        ap[ib * RS] = gb + g * cb;
      }
    }
  }
  for (int k = 0; k < a.nfast_loads; ++k) {
    const unsigned w = ppsci_readlane(R.loads, k);
    const int i = w & 127u, ia = (w >> 7) & 127u, kind = (w >> 14) & 3u;
    if (act && kind == 1) fx->tin[ia * PPSCI_TILE + pt] = ap[i * RS];  // (invalid lanes carry 0: their seeds are never set)
  }
}

// The same for programs of at most EPI_FAST_NREG instructions, with the VM's register file (values and adjoints) in VGPRs:
// the operand of a step is a uniformly indexed register (s_set_gpr_idx), so a step has NO memory access at all -- under
// load an LDS round trip per step (the other workgroup of the CU streams GEMM operands through the same LDS) was most
// of the ~350 cycles a step of epi_tile_fast takes.  Lanes 16..63 compute the same values as lanes 0..15 (pt = lane & 15:
// same addresses) and only the stores are predicated.  fx->uls: [mS][16] the tile's stream values (bias included).
// GLOBAL (the one-launch step kernel of padded width 32, EPI_RF_TILED: every wave runs the program of ITS tile): streams,
// inputs and dL/dU are the [row][N] arrays in memory; `lacc` has stride `lstride`.
template <bool GLOBAL>
__device__ __forceinline__ void epi_tile_fast_regs(const EpiArgs& a, const EpiFastRegs& R, const bool act, const long long p,
                                                   const long long pp, const bool valid, const int pt, float* const lacc,
                                                   const int lstride, const EpiFused* fx, const float* const uls) {
  EPI_FT(0)
  float rv[EPI_FAST_NREG], ra[EPI_FAST_NREG];
#pragma unroll
  for (int i = 0; i < EPI_FAST_NREG; ++i) rv[i] = 0.f, ra[i] = 0.f;
#define EPI_PF(k_, L_, W_)                                                             \
  float L_ = 0.f, W_ = 1.f;                                                            \
  if (k_ < a.e.n_res) {                                                                \
    const unsigned t_ = ppsci_readlane(R.terms, k_);                                   \
    const int lb_ = (int)((t_ >> 7) & 31u) - 1, wt_ = (int)((t_ >> 12) & 31u) - 1, ar_ = (int)((t_ >> 17) & 31u) - 1; \
    if (lb_ >= 0) L_ = a.aux[lb_][pp];                                                 \
    if (wt_ >= 0) W_ = a.aux[wt_][pp];                                                 \
    if (ar_ >= 0) W_ *= a.aux[ar_][pp];                                                \
  }
  EPI_PF(0, lab0, w0)
  EPI_PF(1, lab1, w1)
#undef EPI_PF
  if (!GLOBAL && act)
    for (int q = pt; q < fx->mS * PPSCI_TILE; q += PPSCI_TILE) fx->tin[q] = 0.f;
  EPI_FT(1)
  // ---- memory operands and constants, four at a time (one LDS round trip per four)
  for (int k0 = 0; k0 < a.nfast_loads; k0 += 4) {
    unsigned w[4];
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kk = k0 + j < a.nfast_loads ? k0 + j : a.nfast_loads - 1;
      w[j] = ppsci_readlane(R.loads, kk);
      const float cv = __builtin_bit_cast(float, ppsci_readlane(__builtin_bit_cast(unsigned, R.cvals), kk));
      const int ia = (w[j] >> 7) & 127u, kind = (w[j] >> 14) & 3u;
      if (kind == 3) v[j] = cv;
      else if (kind == 2) v[j] = a.aux[ia][pp];
      else if (GLOBAL) v[j] = kind == 1 ? a.U[(long long)ia * a.N + pp] : a.x[ia][pp];
      else v[j] = (kind == 1 ? uls : fx->tinx)[ia * PPSCI_TILE + pt];
    }
    // (unconditional writes: a conditional write to a uniformly indexed register array is compiled as a copy of the whole
    // array + a select per register; the clamped entries past the end rewrite the last entry with its own value)
#pragma unroll
    for (int j = 0; j < 4; ++j) rv[w[j] & 127u] = v[j];
  }
  EPI_FT(2)
  // ---- forward steps
  for (int k = 0; k < a.nfast; ++k) {
    const unsigned w = ppsci_readlane(R.steps, k);
    const int ia = w & 127u, ib = (w >> 7) & 127u, id = (w >> 14) & 127u;
    const float x = rv[ia], y = rv[ib];
    const float lin = epi_fast_code((w >> 22) & 3u) * x + epi_fast_code((w >> 24) & 3u) * y;
    rv[id] = ((w >> 21) & 1u) ? x * y : lin;
  }
  EPI_FT(3)
  // ---- residuals, MSE terms and their seeds (epi_point, PPSCI_LOSS_MSE)
  for (int k = 0; k < a.e.n_res; ++k) {
    const unsigned t = ppsci_readlane(R.terms, k);
    const float scale = __builtin_bit_cast(float, ppsci_readlane(__builtin_bit_cast(unsigned, R.scale), k));
    const int iv = t & 127u, lb = (int)((t >> 7) & 31u) - 1, wt = (int)((t >> 12) & 31u) - 1, ar = (int)((t >> 17) & 31u) - 1;
    float lab, wk;
    if (k == 0) lab = lab0, wk = w0;
    else if (k == 1) lab = lab1, wk = w1;
    else {
      lab = lb >= 0 ? a.aux[lb][pp] : 0.f;
      wk = wt >= 0 ? a.aux[wt][pp] : 1.f;
      if (ar >= 0) wk *= a.aux[ar][pp];
    }
    const float rval = rv[iv];
    if (GLOBAL) {
      if (act && valid && a.resid != nullptr) a.resid[(long long)k * a.N + p] = rval;
    } else if (act && fx->rres != nullptr) fx->rres[k * PPSCI_TILE + pt] = rval;
    const float diff = rval - lab;
    const float wgt = wk * scale;
    if (act && valid) lacc[k * lstride] += wgt * diff * diff;
    ra[iv] += valid ? 2.f * wgt * diff : 0.f;
  }
  EPI_FT(4)
  // ---- reverse steps
  for (int k = a.nfast - 1; k >= 0; --k) {
    const unsigned w = ppsci_readlane(R.steps, k);
    const int ia = w & 127u, ib = (w >> 7) & 127u, id = (w >> 14) & 127u;
    const float g = ra[id], x = rv[ia], y = rv[ib];
    const bool prod = (w >> 21) & 1u;
    const float ca = prod ? y : epi_fast_code((w >> 26) & 3u), cb = prod ? x : epi_fast_code((w >> 28) & 3u);
    // two sequential read-modify-writes, no branch: with a == b (u * u, -u) the second one sees the first -- the generic
    // order (abar + g ca) + g cb
    ra[ia] = ra[ia] + g * ca;
    ra[ib] = ra[ib] + g * cb;
  }
  EPI_FT(5)
  for (int k = 0; k < a.nfast_loads; ++k) {
    const unsigned w = ppsci_readlane(R.loads, k);
    const int i = w & 127u, ia = (w >> 7) & 127u, kind = (w >> 14) & 3u;
    const float g = ra[i];
    if (GLOBAL) {
      if (act && valid && kind == 1 && a.Ubar != nullptr) a.Ubar[(long long)ia * a.N + p] = g;
    } else if (act && kind == 1) fx->tin[ia * PPSCI_TILE + pt] = g;  // (invalid lanes carry 0: their seeds are never set)
  }
  EPI_FT(6)
}

// Block reduction of the loss terms (and equation-parameter adjoints) the lanes have summed over their points: a butterfly
// inside every wave (register shuffles), then the waves' sums in
// wave order -- a fixed shape, so deterministic; two LDS-only barriers in all (a 256-wide LDS tree with a full
// __syncthreads() per level costs ten barriers per term, the first of which also waits for every adjoint store).
// `red`: (PPSCI_MAX_RES + PPSCI_MAX_EPARAM) * 16 floats of LDS.
template <int MODE>
__device__ __forceinline__ void epi_finale(const EpiArgs& a, float* red, float (&lsum)[PPSCI_MAX_RES],
                                           float (&padj)[PPSCI_MAX_EPARAM]) {
  const int tid = threadIdx.x;
  const int wv = tid >> 6, nwv = (int)(blockDim.x >> 6);
  ppsci_block_sync_lds();  // LDS modes: the register file's last reads come first (`red` is its own region, but cheap)
#pragma unroll
  for (int k = 0; k < PPSCI_MAX_RES; ++k) {
    if (k < a.e.n_res) {
      float v = lsum[k];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
      if ((tid & 63) == 0) red[k * 16 + wv] = v;
    }
  }
  if (a.ep_part != nullptr) {
#pragma unroll
    for (int k = 0; k < PPSCI_MAX_EPARAM; ++k) {
      float v = padj[k];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
      if ((tid & 63) == 0) red[(PPSCI_MAX_RES + k) * 16 + wv] = v;
    }
  }
  ppsci_block_sync_lds();
  if (tid < a.e.n_res) {
    float t = 0.f;
    for (int w = 0; w < nwv; ++w) t += red[tid * 16 + w];
    // (inside the one-launch step kernel, or with loss_out, another workgroup reads the row before the launch ends)
    if (MODE == EPI_RF_TILED || MODE == EPI_RF_FUSED || a.loss_out != nullptr) ppsci_store_agent(&a.partials[(long long)blockIdx.x * a.e.n_res + tid], t);
    else a.partials[(long long)blockIdx.x * a.e.n_res + tid] = t;
  }
  if (a.ep_part != nullptr && tid >= 64 && tid < 64 + PPSCI_MAX_EPARAM) {
    const int k = tid - 64;
    float t = 0.f;
    for (int w = 0; w < nwv; ++w) t += red[(PPSCI_MAX_RES + k) * 16 + w];
    a.ep_part[(long long)blockIdx.x * PPSCI_MAX_EPARAM + k] = t;
  }
  if (MODE != EPI_RF_TILED && MODE != EPI_RF_FUSED && a.loss_out != nullptr) {
    // ---- loss terms without a reduction launch: the workgroup that finishes LAST sums all workgroups' rows, in a fixed
    // order (thread t: rows t, t + 256, ...; then the butterfly and the waves in order) -- deterministic whichever it is.
    // Rows were written with agent-scope stores; the ticket is taken after they have completed (ppsci_common.h).
    ppsci_block_sync_mem();
    if (tid == 0) ((unsigned*)red)[0] = atomicAdd(a.counter, 1u);
    ppsci_block_sync_lds();
    const unsigned ticket = ((const unsigned*)red)[0];
    ppsci_block_sync_lds();
    if (ticket != gridDim.x - 1) return;
    ppsci_acquire_agent();
    if (tid == 0) *a.counter = 0u;
    for (int k = 0; k < a.e.n_res; ++k) {
      float v = 0.f;
#pragma unroll 4
      for (int r = tid; r < (int)gridDim.x; r += (int)blockDim.x) v += a.partials[(long long)r * a.e.n_res + k];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
      if ((tid & 63) == 0) red[16 + wv] = v;
      ppsci_block_sync_lds();
      if (tid == 0) {
        float t = 0.f;
        for (int w = 0; w < nwv; ++w) t += red[16 + w];
        a.loss_out[k] = t;
      }
      ppsci_block_sync_lds();
    }
  }
}

// `red`: EPI_BLOCK floats of LDS for the loss reductions, followed by the register file (LDS modes).
template <int MODE>
__device__ __forceinline__ void epilogue_body(const EpiArgs& a, float* red) {
  static_assert(MODE != EPI_RF_FUSED, "the fused tile kernel calls epi_point / epi_finale itself");
  const int tid = threadIdx.x;
  const int n = a.e.n_instr;
  constexpr bool LDSRF = MODE != EPI_RF_SCRATCH;
  float v_s[LDSRF ? 1 : PPSCI_MAX_PROG], adj_s[LDSRF ? 1 : PPSCI_MAX_PROG];
  const int RS = MODE == EPI_RF_SCRATCH ? 1 : (MODE == EPI_RF_LDS ? EPI_BLOCK : (int)(blockDim.x >> 2));
  const int slot = MODE == EPI_RF_TILED ? ((tid >> 6) * PPSCI_TILE + (tid & 15)) : tid;
  float* const vp = LDSRF ? red + EPI_BLOCK + slot : v_s;
  float* const ap = LDSRF ? red + EPI_BLOCK + (long long)n * RS + slot : adj_s;
  float lsum[PPSCI_MAX_RES];
  for (int k = 0; k < PPSCI_MAX_RES; ++k) lsum[k] = 0.f;
  float padj[PPSCI_MAX_EPARAM];  // adjoints of the equation parameters, summed over this lane's points
#pragma unroll
  for (int k = 0; k < PPSCI_MAX_EPARAM; ++k) padj[k] = 0.f;

  if (MODE == EPI_RF_TILED && a.fast != nullptr && n <= EPI_FAST_NREG) {
    // pre-decoded program on a register file in VGPRs (epi_tile_fast_regs): no interpreter, no LDS round trip per step.
    // All 64 lanes of a wave go through the steps (the table look-ups are wave-wide); lanes 0..15 own the tile's points.
    const int lane = tid & 63;
    EpiFastRegs R;
    R.steps = a.fast[lane];
    R.loads = a.fast[64 + lane];
    R.terms = a.fast[128 + lane];
    R.cvals = __builtin_bit_cast(float, a.fast[192 + lane]);
    R.scale = a.e.res[lane & (PPSCI_MAX_RES - 1)].scale;
    float* const lacc = red + EPI_BLOCK + slot;  // [PPSCI_MAX_RES][RS]: the (unused) LDS register file
    if (lane < PPSCI_TILE)
      for (int k = 0; k < PPSCI_MAX_RES; ++k) lacc[k * RS] = 0.f;
    for (int it = 0; it < a.iters; ++it) {
      const int tile = ppsci_tile_index(it, (int)(blockDim.x >> 6));
      const long long p = (long long)tile * PPSCI_TILE + (lane & 15);
      const bool valid = tile < a.ntiles && p < a.N;
      epi_tile_fast_regs<true>(a, R, lane < PPSCI_TILE, p, valid ? p : 0, valid, lane & 15, lacc, RS, nullptr, nullptr);
    }
#pragma unroll
    for (int k = 0; k < PPSCI_MAX_RES; ++k) lsum[k] = lane < PPSCI_TILE ? lacc[k * RS] : 0.f;
    epi_finale<MODE>(a, red, lsum, padj);
    return;
  }
  for (int it = 0; it < a.iters; ++it) {
    if (MODE == EPI_RF_TILED && (tid & 63) >= PPSCI_TILE) break;  // lanes 0..15 of every wave run the tile's points
    long long p;
    bool valid;
    if (MODE == EPI_RF_TILED) {
      const int tile = ppsci_tile_index(it, (int)(blockDim.x >> 6));
      p = (long long)tile * PPSCI_TILE + (tid & 15);
      valid = tile < a.ntiles && p < a.N;
    } else {
      p = ((long long)it * gridDim.x + blockIdx.x) * EPI_BLOCK + tid;
      valid = p < a.N;
    }
    const long long pp = valid ? p : 0;
    epi_point<MODE>(a, vp, ap, RS, n, p, pp, valid, tid & 15, lsum, padj, nullptr);
  }
  epi_finale<MODE>(a, red, lsum, padj);
}
