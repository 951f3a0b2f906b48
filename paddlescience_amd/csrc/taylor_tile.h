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
//   T -> N goes through a 1.25 KiB per-wave LDS scratch.
#pragma once
#include "ppsci_common.h"

#ifndef PPSCI_EMU
#include <hip/hip_runtime.h>
#endif

// value and first three derivatives of the activation (SURVEY.md Appendix A;
// /root/reference/ppsci/arch/activation.py:77-88 Silu = x*sigmoid(x), :139-154 tanh / sin)
__device__ __forceinline__ void ppsci_act_eval(int act, float z, float& s, float& d1, float& d2, float& d3) {
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

// mlp.py:286-291: `skip = y; y = y + skip` on even hidden layers after the first one.
__device__ __forceinline__ float ppsci_zscale(const ppsci_mlp_desc& d, int layer) {
  return (d.skip_connection && (layer & 1) == 0 && layer >= 2) ? 2.f : 1.f;
}

// sum over the 16 lanes that share g (i.e. over the 16 points of the tile); every lane gets it.
__device__ __forceinline__ float ppsci_row_sum16(float v) {
  v += __shfl_xor(v, 1, 16);
  v += __shfl_xor(v, 2, 16);
  v += __shfl_xor(v, 4, 16);
  v += __shfl_xor(v, 8, 16);
  return v;
}

// sum over g (the four 16-lane groups); every lane gets it.
__device__ __forceinline__ float ppsci_group_sum4(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// T layout -> N layout of one 16x16 block through the wave-private scratch.
__device__ __forceinline__ f32x4 ppsci_t2n(f32x4 v, float* scr, int g, int c) {
  *(f32x4*)&scr[c * PPSCI_SCR_LD + 4 * g] = v;  // scr[point][feature]
  ppsci_wave_sync();
  f32x4 o;
#pragma unroll
  for (int step = 0; step < 4; ++step) o[step] = scr[(4 * g + step) * PPSCI_SCR_LD + c];
  ppsci_wave_sync();
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
