// taylor_tile.h -- device helpers shared by the Taylor-mode forward and reverse kernels.
//
// Tile model (gfx950, wave64, v_mfma_f32_16x16x4_f32 -- exact fp32 at the fp32 vector rate):
//   one wave owns a tile of 16 collocation points.  Lane l = 16*g + c (g = l>>4, c = l&15).
//   "T layout": a [HP features x 16 points] activation matrix of one stream lives in NB float4
//   registers per lane; register block `blk`, component r of lane (g,c) holds
//       X[feature = 16*blk + 4*g + r][point = c].
//   This is exactly the C/D layout of the 16x16x4 MFMA (row = 4*(l>>4)+r, col = l&15) AND its B
//   operand layout (B[k = l>>4][j = l&15]) for k-step (blk, r), so a layer's output registers
//   feed the next layer's MFMAs without any data movement: Z^T = W^T-fragments x H^T.
//   "N layout" (needed for the weight-gradient GEMM, whose contraction runs over points):
//       component `step` of lane (g,c) holds X[feature = 16*blk + c][point = 4*g + step].
//   T -> N goes through a per-wave LDS scratch (one 16x16 slot per stream).
#pragma once
#include "ppsci_common.h"

#ifndef PPSCI_EMU
#include <hip/hip_runtime.h>
#endif

// tanh(x) = 1 - 2 / (exp(2x) + 1): v_mul, v_exp_f32, v_add, v_rcp_f32, v_fma -- five VALU issue slots (ocml
// tanhf costs ~45; a 2-ulp variant with an odd polynomial below |x| = 0.35 cost 16 and was measurably slower:
// on gfx950 every VALU slot is taken from the fp32 MFMAs).  Absolute error <= 1.2e-7 everywhere (the exp and
// rcp units are 1 ulp); the RELATIVE error grows like 6e-8/|x| near zero, which is irrelevant here: the value
// feeds dot products with O(1) terms, and the derivative factors 1 - s^2, -2 s s' see it damped by s.
// Saturates correctly: exp -> inf gives 1, exp -> 0 gives -1.
__device__ __forceinline__ float ppsci_tanh(float x) {
  return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * x) + 1.f);
}

// Internal template id: a tanh net whose hidden layer 0 is the Fourier embedding (ppsci_mlp_desc.fourier_half).
#define PPSCI_ACT_TANH_FOURIER 16
#define PPSCI_ACT_BASE(ACT) ((ACT) == PPSCI_ACT_TANH_FOURIER ? PPSCI_ACT_TANH : (ACT))

// FourierEmbedding (mlp.py:128-136) as an activation: cos on the first half of the features, sin on the rest.
__device__ __forceinline__ void ppsci_fourier_eval(float z, bool is_cos, float& s, float& d1, float& d2, float& d3) {
  const float sn = sinf(z), cs = cosf(z);
  s = is_cos ? cs : sn;
  d1 = is_cos ? -sn : cs;
  d2 = -s;
  d3 = -d1;
}

// value and first three derivatives of the activation (SURVEY.md Appendix A;
// /root/reference/ppsci/arch/activation.py:77-88 Silu = x*sigmoid(x), :139-154 tanh / sin)
template <int ACT>
__device__ __forceinline__ void ppsci_act_eval(float z, float& s, float& d1, float& d2, float& d3) {
  if (ACT == PPSCI_ACT_TANH) {
    s = ppsci_tanh(z);
    d1 = 1.f - s * s;
    d2 = -2.f * s * d1;
    d3 = d1 * (6.f * s * s - 2.f);
  } else if (ACT == PPSCI_ACT_SILU) {
    float g = 1.f / (1.f + expf(-z));
    float g1 = g * (1.f - g);
    float t = 1.f - 2.f * g;
    s = z * g;
    d1 = g + z * g1;
    d2 = g1 * (2.f + z * t);
    d3 = 3.f * g1 * t + z * (g1 * t * t - 2.f * g1 * g1);
  } else if (ACT == PPSCI_ACT_SIGMOID) {  // nn.Sigmoid
    float g = 1.f / (1.f + expf(-z));
    s = g;
    d1 = g * (1.f - g);
    d2 = d1 * (1.f - 2.f * g);
    d3 = d1 * (1.f - 6.f * d1);
  } else if (ACT == PPSCI_ACT_COS) {  // activation.py Cos
    float sn = sinf(z);
    s = cosf(z);
    d1 = -sn;
    d2 = -s;
    d3 = sn;
  } else if (ACT == PPSCI_ACT_GELU) {  // nn.GELU (exact, erf): x Phi(x)
    const float phi = 0.3989422804014327f * expf(-0.5f * z * z);
    const float Phi = 0.5f * (1.f + erff(z * 0.7071067811865476f));
    s = z * Phi;
    d1 = Phi + z * phi;
    d2 = phi * (2.f - z * z);
    d3 = phi * z * (z * z - 4.f);
  } else if (ACT == PPSCI_ACT_RELU) {  // nn.ReLU (activation.py:141)
    s = z > 0.f ? z : 0.f;
    d1 = z > 0.f ? 1.f : 0.f;
    d2 = 0.f;
    d3 = 0.f;
  } else if (ACT == PPSCI_ACT_LEAKY_RELU) {  // nn.LeakyReLU(), negative_slope = 0.01 (activation.py:144)
    s = z > 0.f ? z : 0.01f * z;
    d1 = z > 0.f ? 1.f : 0.01f;
    d2 = 0.f;
    d3 = 0.f;
  } else if (ACT == PPSCI_ACT_ELU || ACT == PPSCI_ACT_SELU) {
    // nn.ELU(): alpha = 1 (activation.py:140); nn.SELU(): scale * (max(0,x) + min(0, alpha (e^x - 1))) (:142)
    const float scale = ACT == PPSCI_ACT_SELU ? 1.0507009873554804934193349852946f : 1.f;
    const float alpha = ACT == PPSCI_ACT_SELU ? 1.6732632423543772848170429916717f : 1.f;
    const float e = scale * alpha * expf(z);
    s = z > 0.f ? scale * z : e - scale * alpha;
    d1 = z > 0.f ? scale : e;
    d2 = z > 0.f ? 0.f : e;
    d3 = d2;
  } else if (ACT == PPSCI_ACT_IDENTITY) {  // nn.Identity (activation.py:151)
    s = z;
    d1 = 1.f;
    d2 = 0.f;
    d3 = 0.f;
  } else {
    s = sinf(z);
    float c = cosf(z);
    d1 = c;
    d2 = -s;
    d3 = -c;
  }
}

// Activations with a trainable per-feature parameter p: value, three derivatives w.r.t. z, and -- for the reverse
// sweep -- the derivatives of (s, d1, d2) w.r.t. p.
//   Swish  s = z g(pz), g = sigmoid                (activation.py:49-58)
//   Stan   s = tanh(z) (1 + p z)                    (activation.py:28-46)
template <int ACT>
__device__ __forceinline__ void ppsci_act_eval_p(float z, float p, float& s, float& d1, float& d2, float& d3,
                                                 float& sp, float& d1p, float& d2p) {
  if (ACT == PPSCI_ACT_SWISH) {
    const float g = 1.f / (1.f + expf(-p * z));
    const float g1 = g * (1.f - g), g2 = g1 * (1.f - 2.f * g), g3 = g1 * (1.f - 6.f * g1);
    s = z * g;
    d1 = g + p * z * g1;
    d2 = 2.f * p * g1 + p * p * z * g2;
    d3 = 3.f * p * p * g2 + p * p * p * z * g3;
    sp = z * z * g1;
    d1p = 2.f * z * g1 + p * z * z * g2;
    d2p = 2.f * g1 + 4.f * p * z * g2 + p * p * z * z * g3;
  } else {  // Stan
    const float t = tanhf(z);
    const float t1 = 1.f - t * t, t2 = -2.f * t * t1, t3 = t1 * (6.f * t * t - 2.f);
    const float q = 1.f + p * z;
    s = t * q;
    d1 = t1 * q + p * t;
    d2 = t2 * q + 2.f * p * t1;
    d3 = t3 * q + 3.f * p * t2;
    sp = z * t;
    d1p = z * t1 + t;
    d2p = z * t2 + 2.f * t1;
  }
}

// the four parameters of features 16*blk + 4g .. +3 of hidden layer l (zero outside the unpadded width)
__device__ __forceinline__ f32x4 ppsci_act_params4(const float* params, const ppsci_derived& q, int H, int l, int blk, int g) {
  f32x4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int f = 16 * blk + 4 * g + r;
    v[r] = f < H ? params[q.offA + l * H + f] : 0.f;
  }
  return v;
}

// What taylor_fwd stashes for the value stream of a hidden layer and how taylor_bwd turns it back into the
// activation value and its first three derivatives: for tanh the activation VALUE s = tanh(z) is stashed,
// because every derivative is a polynomial in s (1 - s^2, -2 s s', s'(6 s^2 - 2)) and the reverse sweep never
// needs z itself -- this removes the exp/rcp/polynomial evaluation (about 40 % of the reverse kernel's
// pointwise VALU work, which on gfx950 issues on the same pipe as the fp32 MFMAs); silu and sin stash z.
template <int ACT>
__device__ __forceinline__ void ppsci_act_from_stash(float v, float& s, float& d1, float& d2, float& d3) {
  if (ACT == PPSCI_ACT_TANH) {
    s = v;
    d1 = 1.f - s * s;
    d2 = -2.f * s * d1;
    d3 = d1 * (6.f * s * s - 2.f);
  } else {
    ppsci_act_eval<ACT>(v, s, d1, d2, d3);
  }
}

// sigma(z) and the first three derivatives for the four features a lane holds, from what the stash holds.  tanh nets: plain
// float4 arithmetic, so that the packed fp32 instructions the compiler selects pair feature r with r + 1 of ONE quantity.
// The per-feature scalar form let the SLP vectoriser pair two DIFFERENT quantities of one feature -- (second, third derivative)
// x a splat of the first -- as  v_pk_mul_f32 d, a, b op_sel:[0,1]  (low result = a.lo * b.HI): on MI355X, with two workgroups
// on a CU, the low result of that one instruction came out as a.lo * 0 in lanes 48..63 in about 1 of 500 workgroups
// (run-to-run differences of 1e-3 in one feature of the top layer's zbar; tools/det_probe7.py).
// tests/test_isa_lint.py keeps that instruction form out of the library.
template <int ACT>
__device__ __forceinline__ void ppsci_act_from_stash4(const f32x4 v, f32x4& s, f32x4& d1, f32x4& d2, f32x4& d3) {
  if constexpr (ACT == PPSCI_ACT_TANH) {
    const f32x4 one = {1.f, 1.f, 1.f, 1.f}, two = {2.f, 2.f, 2.f, 2.f}, six = {6.f, 6.f, 6.f, 6.f};
    s = v;
    d1 = one - s * s;
    d2 = -two * s * d1;
    d3 = d1 * (six * s * s - two);
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float sv, a, b, c;
      ppsci_act_from_stash<ACT>(v[r], sv, a, b, c);
      s[r] = sv, d1[r] = a, d2[r] = b, d3[r] = c;
    }
  }
}

// Derivatives 1..5 of the activation for the third / fourth-order Taylor streams (Faa di Bruno needs sigma^(4) in the
// forward sweep and sigma^(5) in the reverse one).  `v` is what the stash holds: tanh(z) for tanh nets when
// FROM_STASH, z otherwise.  tanh: polynomials in s; sigmoid family: polynomials in g1 = g(1-g); gelu: Hermite-type
// polynomials times the Gaussian.
template <int ACT, bool FROM_STASH>
__device__ __forceinline__ void ppsci_act_eval5(float v, float& s, float (&d)[5]) {
  if (ACT == PPSCI_ACT_TANH) {
    s = FROM_STASH ? v : ppsci_tanh(v);
    const float s2 = s * s;
    d[0] = 1.f - s2;
    d[1] = -2.f * s * d[0];
    d[2] = d[0] * (6.f * s2 - 2.f);
    d[3] = d[0] * s * (16.f - 24.f * s2);
    d[4] = d[0] * (16.f + s2 * (-120.f + 120.f * s2));
  } else if (ACT == PPSCI_ACT_SILU || ACT == PPSCI_ACT_SIGMOID) {
    const float g = 1.f / (1.f + expf(-v));
    const float g1 = g * (1.f - g), t = 1.f - 2.f * g;
    const float g2 = g1 * t, g3 = g1 * (1.f - 6.f * g1), g4 = g2 * (1.f - 12.f * g1);
    const float g5 = g1 * (1.f + g1 * (-30.f + 120.f * g1));
    if (ACT == PPSCI_ACT_SIGMOID) {
      s = g;
      d[0] = g1, d[1] = g2, d[2] = g3, d[3] = g4, d[4] = g5;
    } else {  // z g(z): k g^(k-1) + z g^(k)
      s = v * g;
      d[0] = g + v * g1, d[1] = 2.f * g1 + v * g2, d[2] = 3.f * g2 + v * g3, d[3] = 4.f * g3 + v * g4, d[4] = 5.f * g4 + v * g5;
    }
  } else if (ACT == PPSCI_ACT_GELU) {
    const float phi = 0.3989422804014327f * expf(-0.5f * v * v);
    const float Phi = 0.5f * (1.f + erff(v * 0.7071067811865476f));
    const float z2 = v * v;
    s = v * Phi;
    d[0] = Phi + v * phi;
    d[1] = phi * (2.f - z2);
    d[2] = phi * v * (z2 - 4.f);
    d[3] = phi * (-4.f + z2 * (7.f - z2));
    d[4] = phi * v * (18.f + z2 * (-11.f + z2));
  } else if (ACT == PPSCI_ACT_COS) {
    const float sn = sinf(v), cs = cosf(v);
    s = cs;
    d[0] = -sn, d[1] = -cs, d[2] = sn, d[3] = cs, d[4] = -sn;
  } else if (ACT == PPSCI_ACT_SIN) {
    const float sn = sinf(v), cs = cosf(v);
    s = sn;
    d[0] = cs, d[1] = -sn, d[2] = -cs, d[3] = sn, d[4] = cs;
  } else {  // piecewise-linear / exponential-linear family: one evaluation serves every order
    float d1, d2, d3;
    ppsci_act_eval<ACT>(v, s, d1, d2, d3);
    d[0] = d1, d[1] = d2, d[2] = d3, d[3] = d3, d[4] = d3;  // elu / selu: c e^z below 0; relu-type: 0
  }
}

// mlp.py:286-291: `skip = y; y = y + skip` on even hidden layers after the first one.
__device__ __forceinline__ float ppsci_zscale(const ppsci_mlp_desc& d, int layer) {
  const float w0 = d.act_scale != 0.f ? d.act_scale : 1.f;  // Siren: act(w0 * z)
  if (d.fourier_half > 0) {  // kernel layer 0 is the embedding; self.linears[i] is kernel layer i + 1
    if (layer == 0) return 1.f;
    layer -= 1;
  }
  return ((d.skip_connection && (layer & 1) == 0 && layer >= 2) ? 2.f : 1.f) * w0;
}

// Sum over the 16 lanes of a DPP row (= over the 16 points of the tile, lanes sharing g).
// The total is valid in the LAST lane of the row (c == 15); VALU-rate (4 DPP adds), no LDS.
__device__ __forceinline__ float ppsci_row_sum16_last(float v) {
#define PPSCI_DPP_ADD(ctrl)                                                                            \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true))
  PPSCI_DPP_ADD(0x111);  // row_shr:1
  PPSCI_DPP_ADD(0x112);  // row_shr:2
  PPSCI_DPP_ADD(0x114);  // row_shr:4
  PPSCI_DPP_ADD(0x118);  // row_shr:8
#undef PPSCI_DPP_ADD
  return v;
}

// sum over g (the four 16-lane groups); every lane gets it.
__device__ __forceinline__ float ppsci_group_sum4(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// T layout -> N layout of NS 16x16 blocks at once through the wave-private scratch (NS slots).
template <int NS>
__device__ __forceinline__ void ppsci_t2n(const f32x4 (&v)[NS], f32x4 (&o)[NS], float* scr, int g, int c) {
#pragma unroll
  for (int s = 0; s < NS; ++s) *(f32x4*)&scr[s * PPSCI_SCR_FLOATS + c * PPSCI_SCR_LD + 4 * g] = v[s];  // scr[point][feature]
  ppsci_wave_sync();
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int step = 0; step < 4; ++step) o[s][step] = scr[s * PPSCI_SCR_FLOATS + (4 * g + step) * PPSCI_SCR_LD + c];
  ppsci_wave_sync();
}

// ---- fp32 GEMMs on the bf16 (XDL) matrix pipe -----------------------------------------------------------------
// v_mfma_f32_16x16x4_f32 runs at the fp32 VECTOR rate (157 TFLOP/s) and, on gfx950, shares its issue slots with the
// pointwise VALU work; the bf16 MFMAs run 16x faster and overlap with VALU.  Every fp32 operand is
// therefore split into three bf16 terms  x = h + m + l  (8 + 8 + 8 significand bits, round-to-nearest at each level,
// so the sum is exact up to the last fp32 bit) and a product is evaluated as the six cross terms that matter,
//      a b ~= a_h b_l + a_l b_h + a_m b_m + a_h b_m + a_m b_h + a_h b_h           (dropped: 2^-24 relative and below)
// each an exact bf16 x bf16 product accumulated in fp32 by v_mfma_f32_16x16x32_bf16.  Measured on MI355X
// (tools/microbench/xdl_split.hip): rms error 1.8e-8 of sum|a||b| at K = 64 against 2.8e-8 for the fp32 MFMA's fmaf
// chain -- at least as accurate as the instruction it replaces -- at 6/16 of its matrix-pipe time.
// Layouts: the K = 32 operand of lane (g, c) is 8 bf16 = the 4 + 4 values that two T-layout (or N-layout) float4
// registers of that lane hold, so a "half operand" (4 bf16, one u32x2) is the split of ONE float4 register and any
// two blocks / streams can be paired; the A halves come from LDS with one ds_read_b64 each.
#ifndef PPSCI_XDL
#define PPSCI_XDL 1  // 0: the fp32-input MFMA everywhere (build option for A/B measurements)
#endif
struct ppsci_split4 {
  u32x2 p[3];  // planes h, m, l: 4 packed bf16 each
};
__device__ __forceinline__ float ppsci_bf16lo_f32(unsigned v) { return __builtin_bit_cast(float, v << 16); }
__device__ __forceinline__ float ppsci_bf16hi_f32(unsigned v) { return __builtin_bit_cast(float, v & 0xffff0000u); }
__device__ __forceinline__ ppsci_split4 ppsci_split(f32x4 x) {
  ppsci_split4 o;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float a = x[2 * i], b = x[2 * i + 1];
    const unsigned h = ppsci_cvt_pk_bf16(a, b);
    a = ppsci_bf16_sub_lo(h, a);  // a - (a as bf16): exact; one v_dot2c_f32_bf16 each (ppsci_common.h)
    b = ppsci_bf16_sub_hi(h, b);
    const unsigned m = ppsci_cvt_pk_bf16(a, b);
    a = ppsci_bf16_sub_lo(m, a);
    b = ppsci_bf16_sub_hi(m, b);
    o.p[0][i] = h;
    o.p[1][i] = m;
    o.p[2][i] = ppsci_cvt_pk_bf16(a, b);
  }
  return o;
}
// the six products, small terms first: (plane of A, plane of B)
#define PPSCI_XDL_NPROD 6
__device__ static constexpr int ppsci_xdl_pa[6] = {0, 2, 1, 0, 1, 0};
__device__ static constexpr int ppsci_xdl_pb[6] = {2, 0, 1, 1, 0, 0};
// acc += A B over one K = 32 step (two paired halves per operand), all six products, one accumulator
__device__ __forceinline__ f32x4 ppsci_xdl6(const ppsci_split4& a0, const ppsci_split4& a1, const ppsci_split4& b0,
                                            const ppsci_split4& b1, f32x4 acc) {
#pragma unroll
  for (int q = 0; q < PPSCI_XDL_NPROD; ++q)
    acc = ppsci_xdl32(a0.p[ppsci_xdl_pa[q]], a1.p[ppsci_xdl_pa[q]], b0.p[ppsci_xdl_pb[q]], b1.p[ppsci_xdl_pb[q]], acc);
  return acc;
}
// K = 16 step (one half per operand): the odd stream of a stream-paired contraction
__device__ __forceinline__ f32x4 ppsci_xdl6_k16(const ppsci_split4& a0, const ppsci_split4& b0, f32x4 acc) {
#pragma unroll
  for (int q = 0; q < PPSCI_XDL_NPROD; ++q) acc = ppsci_xdl16(a0.p[ppsci_xdl_pa[q]], b0.p[ppsci_xdl_pb[q]], acc);
  return acc;
}

// XDL weight fragments in LDS: per (row block rb, k block kb) three planes of 64 lanes x 4 bf16 (u32x2):
//   frag[((rb*NB + kb)*3 + plane)*64 + lane]
// forward  (z = W^T h):    rb = output block, lane (g, c) holds W[in = 16kb + 4g + r][out = 16rb + c], r = 0..3
// backward (hbar = W zbar): rb = input block,  lane (g, c) holds W[in = 16rb + c][out = 16kb + 4g + r]
#define PPSCI_XFRAG_FLOATS(HP) ((HP) * (HP) * 3 / 2)  // size of one layer's fragments in floats (6 bytes per weight)
template <bool BWD>
__device__ __forceinline__ void ppsci_stage_frag_xdl(float* dst_, const float* W, int H, int NB, int tid, int nthr) {
  u32x2* dst = (u32x2*)dst_;
  for (int idx = tid; idx < NB * NB * 64; idx += nthr) {
    const int lane = idx & 63, pair = idx >> 6, rb = pair / NB, kb = pair - rb * NB;
    const int g = lane >> 4, c = lane & 15;
    f32x4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int in = BWD ? 16 * rb + c : 16 * kb + 4 * g + r, out = BWD ? 16 * kb + 4 * g + r : 16 * rb + c;
      v[r] = (in < H && out < H) ? W[in * H + out] : 0.f;
    }
    const ppsci_split4 sp = ppsci_split(v);
#pragma unroll
    for (int q = 0; q < 3; ++q) dst[(pair * 3 + q) * 64 + lane] = sp.p[q];
  }
}
// the three planes of fragment (rb, kb) for this lane
__device__ __forceinline__ ppsci_split4 ppsci_load_frag_xdl(const float* frag, int NB, int rb, int kb, int lane) {
  const u32x2* f = (const u32x2*)frag + ((rb * NB + kb) * 3) * 64 + lane;
  ppsci_split4 o;
  o.p[0] = f[0];
  o.p[1] = f[64];
  o.p[2] = f[128];
  return o;
}

// ---- LDS staging of weights --------------------------------------------------------------
// Forward A-fragments of hidden layer l (W is [H,H] row-major [in,out] in HBM):
//   fragF[((ob*NB + kb)*64 + lane)*4 + r] = W[in = 16kb + 4g + r][out = 16ob + c]
// so one ds_read_b128 per lane yields the A operands (A[i=c][k=g] = W^T[out][in]) of 4 k-steps.
__device__ __forceinline__ void ppsci_stage_fragF(float* dst, const float* W, int H, int NB, int tid, int nthr) {
  const int HP = 16 * NB;
  for (int idx = tid; idx < HP * HP; idx += nthr) {
    int in = idx / HP, out = idx - in * HP;
    float v = (in < H && out < H) ? W[in * H + out] : 0.f;
    int ob = out >> 4, c = out & 15, kb = in >> 4, g = (in & 15) >> 2, r = in & 3;
    dst[((ob * NB + kb) * 64 + 16 * g + c) * 4 + r] = v;
  }
}
// Backward A-fragments (hbar_prev = W zbar):
//   fragB[((ib*NB + kb)*64 + lane)*4 + r] = W[in = 16ib + c][out = 16kb + 4g + r]
__device__ __forceinline__ void ppsci_stage_fragB(float* dst, const float* W, int H, int NB, int tid, int nthr) {
  const int HP = 16 * NB;
  for (int idx = tid; idx < HP * HP; idx += nthr) {
    int in = idx / HP, out = idx - in * HP;
    float v = (in < H && out < H) ? W[in * H + out] : 0.f;
    int ib = in >> 4, c = in & 15, kb = out >> 4, g = (out & 15) >> 2, r = out & 3;
    dst[((ib * NB + kb) * 64 + 16 * g + c) * 4 + r] = v;
  }
}

// ---- feature-split XDL kernels (taylor_fwd_wx.inc, taylor_bwd_wx.inc) ----
// LDS chunk index of (feature group fg = 4 features, point pt) inside one 16x16 bf16 plane (64 chunks of 8 bytes):
// feature-group-major, i.e. lane (g, c) owns chunk ~lane, so the 16 lanes of a group always touch one contiguous
// 128-byte run -- conflict-free for every DS form the compiler picks (ds_read_b64, ds_read2st64_b64, ds_write2st64_b64
// are served per 16-lane group over 32 banks; a point-major layout measured 47 % conflict cycles there).  The points
// of groups 2 and 3 are XOR-ed with 8 so that ds_read_b64_tr_b16, whose 32-lane halves gather chunks (fg = 0..3,
// pt = 8 consecutive points), hits 64 distinct banks as well.
__device__ __forceinline__ int ppsci_xchunk(int fg, int pt) { return fg * 16 + (pt ^ ((fg >> 1) << 3)); }

// Exchange buffers that are read as B operands of K = 32 steps keep the chunks of a block PAIR (2j, 2j+1) side by side, so that
// one ds_read_b128 delivers both halves of the operand (half the LDS instructions and twice the LDS rate of two 8-byte
// reads): index, in 8-byte units, of chunk `ch` of plane `pl` of (stream s, block blk) in a buffer of NB blocks per stream.
__device__ __forceinline__ int ppsci_xpair(int NB, int s, int blk, int pl, int ch) {
  return (((s * (NB / 2) + (blk >> 1)) * 3 + pl) * 64 + ch) * 2 + (blk & 1);
}

// u32x4 units per hidden-to-hidden layer of the pre-split global fragments:
//   gfrag[((rb * (NB/2) + kp) * 3 + plane) * 64 + lane] = the K = 32 A operand of row block rb, k-blocks (2kp | 2kp+1)
#define PPSCI_GFRAG_PER_LAYER(NB) ((NB) * ((NB) / 2) * 3 * 64)

// defined in taylor_api.hip: splits the hidden-to-hidden matrices of `params` into the fragment cache entry of
// (params, bwd) on `stream` and returns the device pointer (nullptr + ppsci_set_error on failure)
const void* ppsci_presplit(const float* params, const ppsci_mlp_desc& d, const ppsci_derived& q, int bwd, void* stream);

// one layer's hidden-weight fragments in LDS (floats) and their staging, for the MFMA flavour this build uses
#define PPSCI_FRAG_FLOATS(HP) (PPSCI_XDL ? PPSCI_XFRAG_FLOATS(HP) : (HP) * (HP))
__device__ __forceinline__ void ppsci_stage_fwd_frag(float* dst, const float* W, int H, int NB, int tid, int nthr) {
  if (PPSCI_XDL) ppsci_stage_frag_xdl<false>(dst, W, H, NB, tid, nthr);
  else ppsci_stage_fragF(dst, W, H, NB, tid, nthr);
}
__device__ __forceinline__ void ppsci_stage_bwd_frag(float* dst, const float* W, int H, int NB, int tid, int nthr) {
  if (PPSCI_XDL) ppsci_stage_frag_xdl<true>(dst, W, H, NB, tid, nthr);
  else ppsci_stage_fragB(dst, W, H, NB, tid, nthr);
}

// Tile owned by (iteration, block, wave): consecutive tiles go to different workgroups first, so a
// partially filled last round is spread over many CUs instead of filling a few of them.
__device__ __forceinline__ int ppsci_tile_index(int it, int waves) {
  const int t = (it * waves + (int)(threadIdx.x >> 6)) * (int)gridDim.x + (int)blockIdx.x;
  // wave-uniform by construction; telling the compiler moves the tile's address arithmetic to the scalar
  // unit and turns every `tile < ntiles` guard into a scalar branch instead of an exec-mask region
  return __builtin_amdgcn_readfirstlane(t);
}

// ---- kernel argument blocks ------------------------------------------------------------------
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
  int tile0;     // first tile of this launch (feature-split kernels; `ntiles` stays the END of the range)
  const void* xfrag;  // feature-split XDL kernels: pre-split hidden-weight fragments (ppsci_presplit)
};

struct BwdArgs {
  ppsci_mlp_desc d;
  ppsci_derived q;
  const float* params;
  const float* x[PPSCI_MAX_IN];
  const float* Ubar;
  const f32x4* stash;
  float* partials;  // [gridDim.x][ppsci_small_params]: per-workgroup sums of W0 | b_0..b_{L-1} | W_last | b_last
  f32x4* wpart;     // [ntiles][L-1][NB*NB][64] float4: per-tile hidden-weight gradient blocks
  long long N;
  int ntiles;
  int iters;
  int resident;
  int tile0;     // first tile of this launch (feature-split kernels; `ntiles` stays the END of the range)
  int accum;     // 1: hidden-weight gradient blocks accumulated per WORKGROUP (wpart has one slot per workgroup)
  int xdl_split; // 1: planned onto the register-accumulating feature-split kernel (always one slot per workgroup)
  const void* xfrag;  // feature-split XDL kernels: pre-split hidden-weight fragments (ppsci_presplit)
  // layer-by-layer kernel of padded width 256 (taylor_bwd_lw.inc): one launch per hidden-to-hidden matrix
  int lw;        // 1: planned onto it (the workspace holds `hbuf`)
  int layer;     // the matrix W_layer of this launch
  f32x4* hbuf;   // [ntiles][S][NB][64]: hbar of the layer below, handed from launch to launch
};

// Parameters that are NOT hidden-to-hidden matrices, in the compact order the reverse kernels flush their LDS
// accumulators in: W0 [d0, H] | b_0 .. b_{L-1} [H each] | W_last [H, m] | b_last [m].
__host__ __device__ inline int ppsci_small_params(const ppsci_mlp_desc& d, const ppsci_derived& q) {
  return q.d0 * d.width + d.n_hidden * d.width + d.width * d.d_out + d.d_out +
         (ppsci_act_has_param(d.activation) ? d.n_hidden * d.width : 0);  // ... | p_0 .. p_{L-1} [H each]
}

// Two-stage, fixed-order reduction of the per-tile hidden-weight gradient partials (wgrad_reduce.hip):
// stage 1 sums chunks of tiles, stage 2 sums the chunks and writes the gradient in the canonical parameter
// layout; the entries that are not hidden-to-hidden weights are taken from `small_sum` (compact order above).
#define PPSCI_WRED_CHUNKS 64
// What stage 2 can do in the same launch (the tail of a fused-tile step, taylor_api.hip): row (+)= the sums instead of
// row = ; the loss terms from the workgroups' rows (one more workgroup); the Adam update of the parameters from the
// finished gradient (p != null; lr_t / eps_t: bias-corrected as in ppsci_adam_step).
struct ppsci_wred_extras {
  int accumulate;
  const float* loss_rows;  // [loss_nrows][n_res] or null
  float* loss_out;         // [n_res]
  int loss_nrows, n_res;
  float *p, *m, *v;        // Adam (p == null: none)
  float lr_t, beta1, beta2, eps_t, grad_scale;
};
int ppsci_wgrad_reduce_ex(const ppsci_mlp_desc& d, const ppsci_derived& q, int ntiles, const float* wpart, float* tmp,
                          const float* small_rows, int nsmall_rows, float* tmp_small, float* row, const ppsci_wred_extras& x,
                          void* stream);
int ppsci_wgrad_reduce(const ppsci_mlp_desc& d, const ppsci_derived& q, int ntiles, const float* wpart, float* tmp,
                       const float* small_rows, int nsmall_rows, float* tmp_small, float* row, void* stream);
// one launch: the sums, grad (+)=, the loss terms, Adam and the fragments of the updated hidden matrices (`frag`: the step
// workspace's fragment buffer or null) -- the tail of a fused-tile step (wgrad_reduce.hip wgrad_tail_kernel)
int ppsci_wgrad_tail(const ppsci_mlp_desc& d, const ppsci_derived& q, int nrows, const float* rows_w, const float* rows_s,
                     float* row, const ppsci_wred_extras& x, void* frag, void* stream);
int ppsci_wgrad_reduce_chunks(const ppsci_mlp_desc& d, const ppsci_derived& q, int nrows, const float* rows, long long rowlen,
                              long long off_small, long long off_loss, float* row, const ppsci_wred_extras& x, void* stream);

// per-activation entry points (one translation unit each, so they compile in parallel).
// launch == 0: only plan (fills a.resident / a.iters and *grid_out); launch == 1: plan + launch.
int ppsci_fwd_run_tanh(FwdArgs& a, void* stream, int launch, int* grid_out);
int ppsci_fwd_run_tanh_fourier(FwdArgs& a, void* stream, int launch, int* grid_out);
int ppsci_fwd_run_silu(FwdArgs& a, void* stream, int launch, int* grid_out);
int ppsci_fwd_run_sin(FwdArgs& a, void* stream, int launch, int* grid_out);
int ppsci_fwd_run_gelu(FwdArgs& a, void* stream, int launch, int* grid_out);
int ppsci_fwd_run_swish(FwdArgs& a, void* stream, int launch, int* grid_out);
int ppsci_fwd_run_stan(FwdArgs& a, void* stream, int launch, int* grid_out);
int ppsci_fwd_run_cos(FwdArgs& a, void* stream, int launch, int* grid_out);
int ppsci_fwd_run_sigmoid(FwdArgs& a, void* stream, int launch, int* grid_out);
int ppsci_fwd_run_relu(FwdArgs& a, void* stream, int launch, int* grid_out);
int ppsci_fwd_run_leaky_relu(FwdArgs& a, void* stream, int launch, int* grid_out);
int ppsci_fwd_run_elu(FwdArgs& a, void* stream, int launch, int* grid_out);
int ppsci_fwd_run_selu(FwdArgs& a, void* stream, int launch, int* grid_out);
int ppsci_fwd_run_identity(FwdArgs& a, void* stream, int launch, int* grid_out);
int ppsci_bwd_run_tanh(BwdArgs& a, void* stream, int launch, int* grid_out);
int ppsci_bwd_run_relu(BwdArgs& a, void* stream, int launch, int* grid_out);
int ppsci_bwd_run_leaky_relu(BwdArgs& a, void* stream, int launch, int* grid_out);
int ppsci_bwd_run_elu(BwdArgs& a, void* stream, int launch, int* grid_out);
int ppsci_bwd_run_selu(BwdArgs& a, void* stream, int launch, int* grid_out);
int ppsci_bwd_run_identity(BwdArgs& a, void* stream, int launch, int* grid_out);
int ppsci_bwd_run_tanh_fourier(BwdArgs& a, void* stream, int launch, int* grid_out);
int ppsci_bwd_run_silu(BwdArgs& a, void* stream, int launch, int* grid_out);
int ppsci_bwd_run_sin(BwdArgs& a, void* stream, int launch, int* grid_out);
int ppsci_bwd_run_gelu(BwdArgs& a, void* stream, int launch, int* grid_out);
int ppsci_bwd_run_swish(BwdArgs& a, void* stream, int launch, int* grid_out);
int ppsci_bwd_run_stan(BwdArgs& a, void* stream, int launch, int* grid_out);
int ppsci_bwd_run_cos(BwdArgs& a, void* stream, int launch, int* grid_out);
int ppsci_bwd_run_sigmoid(BwdArgs& a, void* stream, int launch, int* grid_out);
extern "C" void ppsci_set_error(const char* fmt, ...);

// LDS carve sizes (floats)
#define PPSCI_BWD_TINP_FLOATS (2 * (PPSCI_MAX_OUT * (1 + 4 * PPSCI_MAX_DIRS) + PPSCI_MAX_IN))  // 64-bit row pointers
static inline int ppsci_fwd_small_floats(const ppsci_mlp_desc& d, const ppsci_derived& q) {
  // W0s[d0*HP] + Bs[L*HP] + WLs[m*HP] + BLs[4*ceil(m/4)]
  return (q.d0 + d.n_hidden + d.d_out) * q.HP + ((d.d_out + 3) / 4) * 4;
}
// per-wave accumulators of the small tensors in the reverse kernel: gW0[d0*HP] gB[L*HP] (gP[L*HP]) gWL[m*HP] gBL[4*ceil(m/4)]
__host__ __device__ static inline int ppsci_bwd_nacc_small(const ppsci_mlp_desc& d, const ppsci_derived& q) {
  return (q.d0 + d.n_hidden + d.d_out + (ppsci_act_has_param(d.activation) ? d.n_hidden : 0)) * q.HP + ((d.d_out + 3) / 4) * 4;
}
static inline int ppsci_bwd_small_floats(const ppsci_mlp_desc& d, const ppsci_derived& q, int S) {
  // WLs[m*HP] | per wave: small-tensor accumulators | per wave: transpose scratch [S*SCR] | per wave: tile inputs
  // x[d_raw*16] | row-pointer table
  return d.d_out * q.HP + PPSCI_BWD_WAVES * (ppsci_bwd_nacc_small(d, q) + S * PPSCI_SCR_FLOATS + d.d_raw * PPSCI_TILE) +
         PPSCI_BWD_TINP_FLOATS;
}
