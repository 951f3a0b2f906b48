// ppsci_common.h -- shared between the gfx950 build (hipcc) and the test-only CPU SIMT emulator
// build (clang++ -DPPSCI_EMU, tests/emu/hip_emu.h).  The kernels are written once against the
// handful of primitives defined here.
#pragma once

#include "ppsci_hip.h"

#ifdef PPSCI_EMU
#include "hip_emu.h"
#define PPSCI_LAUNCH(KERNEL, ARGT, grid, block, lds, stream, args)                                   \
  do {                                                                                               \
    ARGT _a = (args);                                                                                \
    emu::launch_named(#KERNEL, emu_dim3{(unsigned)(grid)}, emu_dim3{(unsigned)(block)}, (size_t)(lds), \
                      [](void* p) { KERNEL(*(ARGT*)p); }, &_a);                                      \
  } while (0)
#define PPSCI_SET_MAX_LDS(KERNEL, bytes) (0)
#define PPSCI_LAST_LAUNCH_ERROR() (0)
#define PPSCI_OCCUPANCY(KERNEL, block, lds, out) (*(out) = 2, 0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define ppsci_block_sync_lds() __syncthreads()
#define ppsci_block_sync_mem() __syncthreads()
#define ppsci_acquire_agent() ((void)0)
#define ppsci_setprio(p) ((void)0)
#define PPSCI_OPAQUE(v) ((void)0)
static inline void ppsci_store_agent(float* p, float v) { *p = v; }
static inline void ppsci_store_agent4(f32x4* p, f32x4 v) { *p = v; }
// v_readlane_b32 with a wave-uniform lane index; EVERY lane of the wave must execute it (the emulator's collective)
static inline unsigned ppsci_readlane(unsigned v, int lane) {
  float f;
  std::memcpy(&f, &v, 4);
  const float* buf = emu::wave_publish(f);
  unsigned r;
  std::memcpy(&r, &buf[lane & 63], 4);
  return r;
}
#else
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define PPSCI_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) float name[]
// Makes the VGPR value `v` opaque to the optimiser at this point: what is derived from it afterwards can neither be
// hoisted out of the enclosing loops nor strength-reduced into per-operand address registers (taylor_bwd_wx.inc).
#define PPSCI_OPAQUE(v) asm volatile("" : "+v"(v))
// two fp32 -> packed bf16 (round to nearest even): one v_cvt_pk_bf16_f32; low half = a
__device__ __forceinline__ unsigned ppsci_cvt_pk_bf16(float a, float b) {
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){a, b}, bf16x2_));
}
// x minus the low / high bf16 of the packed pair h, as ONE instruction: v_dot2c_f32_bf16  x += h . (-1, 0)  resp.  h . (0, -1)
// (the products are exact, and so is the sum: h is x rounded to 8 significand bits) -- instead of unpacking the bf16 into a
// float (a shift or a mask) and subtracting: 7 instead of 9 VALU instructions per pair of values in ppsci_split.
// The constant operands are kept out of the optimiser's sight (SGPRs): hipcc folds the packed pair (-1, 0) into the inline
// operand "-1.0", which the hardware reads as something else -- 65 456 of 65 536 results wrong on MI355X with literal
// constants, 0 with register operands (tools/microbench/dot2_test.hip).
__device__ __forceinline__ float ppsci_bf16_sub_lo(unsigned h, float x) {
  typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
  unsigned k = 0x0000bf80u;  // (lo, hi) = (-1, 0)
  asm volatile("" : "+s"(k));
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_, h), __builtin_bit_cast(bf16x2_, k), x, false);
}
__device__ __forceinline__ float ppsci_bf16_sub_hi(unsigned h, float x) {
  typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
  unsigned k = 0xbf800000u;  // (lo, hi) = (0, -1)
  asm volatile("" : "+s"(k));
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_, h), __builtin_bit_cast(bf16x2_, k), x, false);
}
// v_mfma_f32_16x16x32_bf16: lane (g = l>>4, c = l&15) supplies A[i = c][k = 8g + j] and B[k = 8g + j][n = c], j = 0..7
// (here as two 4-bf16 halves: j = 0..3 from *_lo, 4..7 from *_hi); C/D as the fp32 16x16 MFMAs (row 4g + r, col c)
__device__ __forceinline__ f32x4 ppsci_xdl32(u32x2 a_lo, u32x2 a_hi, u32x2 b_lo, u32x2 b_hi, f32x4 c) {
  typedef __bf16 bf16x8_ __attribute__((ext_vector_type(8)));
  const u32x4 a = {a_lo[0], a_lo[1], a_hi[0], a_hi[1]}, b = {b_lo[0], b_lo[1], b_hi[0], b_hi[1]};
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_, a), __builtin_bit_cast(bf16x8_, b), c, 0, 0, 0);
}
// the same with the A operand already paired (one 16-byte fragment)
__device__ __forceinline__ f32x4 ppsci_xdl32a(u32x4 a, u32x2 b_lo, u32x2 b_hi, f32x4 c) {
  typedef __bf16 bf16x8_ __attribute__((ext_vector_type(8)));
  const u32x4 b = {b_lo[0], b_lo[1], b_hi[0], b_hi[1]};
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_, a), __builtin_bit_cast(bf16x8_, b), c, 0, 0, 0);
}
// both operands already paired
__device__ __forceinline__ f32x4 ppsci_xdl32aa(u32x4 a, u32x4 b, f32x4 c) {
  typedef __bf16 bf16x8_ __attribute__((ext_vector_type(8)));
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_, a), __builtin_bit_cast(bf16x8_, b), c, 0, 0, 0);
}
// ds_read_b64_tr_b16: 64 bits per lane with a 16-bit-element transpose inside each 16-lane group: element j of lane i
// (i = lane & 15) is element (i & 3) of the 8 bytes that lane 4j + (i >> 2) of the same group addresses (checked on
// MI355X, tools/microbench/tr_test.hip).  `p` must be 8-byte aligned LDS.
__device__ __forceinline__ u32x2 ppsci_lds_read_tr16(const void* p) {
  typedef short s16x4_ __attribute__((ext_vector_type(4)));
  return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_*)p));
}
// v_mfma_f32_16x16x16_bf16: A[i = c][k = 4g + j], B[k = 4g + j][n = c], j = 0..3
__device__ __forceinline__ f32x4 ppsci_xdl16(u32x2 a, u32x2 b, f32x4 c) {
  typedef short s16x4_ __attribute__((ext_vector_type(4)));
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4_, a), __builtin_bit_cast(s16x4_, b), c, 0, 0, 0);
}
#define PPSCI_LAUNCH(KERNEL, ARGT, grid, block, lds, stream, args)                                   \
  hipLaunchKernelGGL(KERNEL, dim3(grid), dim3(block), (lds), (hipStream_t)(stream), (args))
#define PPSCI_SET_MAX_LDS(KERNEL, bytes)                                                             \
  ((int)hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)))
#define PPSCI_LAST_LAUNCH_ERROR() ((int)hipGetLastError())
#define PPSCI_OCCUPANCY(KERNEL, block, lds, out)                                                     \
  ((int)hipOccupancyMaxActiveBlocksPerMultiprocessor((out), (const void*)KERNEL, (block), (size_t)(lds)))
// Orders this wave's LDS traffic for the per-wave scratch transposes.  The fences are restricted to the
// LDS address space ("local"): a plain wavefront-scope fence also emits `s_waitcnt vmcnt(0)`, which drained
// every in-flight stash prefetch and partial store at each of the ~16 transposes per layer (38 full drains
// in the reverse kernel's ISA; the largest single stall found on MI355X).
__device__ __forceinline__ void ppsci_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. every in-flight stash
// prefetch and partial-block store, at each of the ~10 exchange barriers per layer of the feature-split kernels.
__device__ __forceinline__ void ppsci_block_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// Workgroup barrier that also orders GLOBAL memory traffic among the waves of the workgroup (one CU, one L1: waiting for
// the outstanding stores is enough -- no L2 write-back as an agent-scope __threadfence() would do).  The wait is spelled
// out: a workgroup-scope release fence alone emits NO vmcnt wait on gfx950 outside threadgroup-split mode (checked in
// the ISA), and the reduction tree's tickets (taylor_step.inc, ppsci_epilogue_losses) rely on this wave's agent-scope
// stores having COMPLETED before the ticket atomic is issued (vmcnt counts stores as well as loads on gfx9).
__device__ __forceinline__ void ppsci_block_sync_mem() {
  __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) expcnt(0) lgkmcnt(0)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// Stores of the per-workgroup partial sums that OTHER workgroups read later in the same launch (the reduction tree of the
// one-launch step kernel, taylor_step.inc): agent-scope atomic stores, i.e. written through to the level all XCDs share --
// visible to any workgroup that (a) learns through an agent-scope atomic issued after these stores have completed that
// they exist and (b) invalidates its caches (ppsci_acquire_agent) before loading.  The alternative, plain stores +
// __threadfence(), writes the whole L2 of the XCD back (the MB of stash / U just written included): 10 us per
// workgroup measured on MI355X, against < 1 us for this.
__device__ __forceinline__ void ppsci_store_agent(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void ppsci_store_agent4(f32x4* p, f32x4 v) {
  typedef unsigned long long u64x2_ __attribute__((ext_vector_type(2)));
  const u64x2_ b = __builtin_bit_cast(u64x2_, v);
  __hip_atomic_store((unsigned long long*)p, b[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store((unsigned long long*)p + 1, b[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void ppsci_acquire_agent() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
// s_setprio: instruction-issue priority of this wave among the waves of its SIMD (0 .. 3)
#define ppsci_setprio(p) __builtin_amdgcn_s_setprio(p)
// v_readlane_b32 with a wave-uniform lane index (the result is a scalar); executed by every lane of the wave
__device__ __forceinline__ unsigned ppsci_readlane(unsigned v, int lane) {
  return (unsigned)__builtin_amdgcn_readlane((int)v, lane);
}
#endif

extern "C" int ppsci_get_max_grid(void);
extern "C" int ppsci_get_wide_min_nb(void);
extern "C" int ppsci_get_bwd_accum(void);
extern "C" int ppsci_get_bwd_layerwise(void);

#ifndef PPSCI_FWD_WAVES
#define PPSCI_FWD_WAVES 8   // waves (16-point tiles in flight) per forward workgroup (8 measured 7% faster than 4)
#endif
#ifndef PPSCI_BWD_WAVES
#define PPSCI_BWD_WAVES 4   // waves per reverse-sweep workgroup
#endif
#ifndef PPSCI_NUM_CU
#define PPSCI_NUM_CU 256         // MI355X; the CPU emulator build uses 4 so that multi-round paths stay cheap to test
#endif
#define PPSCI_TILE 16            // collocation points per wave tile (= MFMA N)
#define PPSCI_SCR_LD 20          // row stride (floats) of the per-wave 16x16 transpose scratch
#define PPSCI_SCR_FLOATS (16 * PPSCI_SCR_LD)
#define PPSCI_LDS_LIMIT_BYTES (160 * 1024)

// Everything the kernels need that is derived from ppsci_mlp_desc (filled on the host).
struct ppsci_derived {
  int32_t d0;   // embedded input width
  int32_t HP;   // hidden width padded to a multiple of 16
  int32_t NB;   // HP / 16
  int32_t P;    // parameter count
  int32_t offW[PPSCI_MAX_HIDDEN + 1];
  int32_t offB[PPSCI_MAX_HIDDEN + 1];
  int32_t offA;  // activation parameters p_l[H], l = 0 .. L-1 (behind the last bias); == P when there are none
};

__host__ __device__ static inline bool ppsci_act_has_param(int act) { return act == PPSCI_ACT_SWISH || act == PPSCI_ACT_STAN; }

static inline int ppsci_derive(const ppsci_mlp_desc* d, ppsci_derived* q) {
  if (d->d_raw < 1 || d->d_raw > PPSCI_MAX_IN) return PPSCI_E_INVALID;
  if (d->n_hidden < 1 || d->n_hidden > PPSCI_MAX_HIDDEN) return PPSCI_E_INVALID;
  if (d->width < 1 || d->d_out < 1 || d->d_out > PPSCI_MAX_OUT) return PPSCI_E_INVALID;
  if (d->n1 < 0 || d->n1 > PPSCI_MAX_DIRS || d->n2 < 0 || d->n2 > d->n1) return PPSCI_E_INVALID;
  if (d->n3 < 0 || d->n3 > d->n2 || d->n4 < 0 || d->n4 > d->n3) return PPSCI_E_INVALID;
  int d0 = 0;
  for (int j = 0; j < d->d_raw; ++j) d0 += (d->embed[j] == PPSCI_EMBED_PERIOD) ? 2 : 1;
  q->d0 = d0;
  // the kernels are instantiated for NB in {2, 4, 8, 16}: round the padded width up
  int nb = (d->width + 15) / 16;
  q->NB = nb <= 2 ? 2 : (nb <= 4 ? 4 : (nb <= 8 ? 8 : (nb <= 16 ? 16 : nb)));
  q->HP = q->NB * 16;
  int off = 0, fin = d0;
  for (int l = 0; l <= d->n_hidden; ++l) {
    int fout = (l == d->n_hidden) ? d->d_out : d->width;
    q->offW[l] = off;
    off += fin * fout;
    q->offB[l] = off;
    off += fout;
    fin = fout;
  }
  q->offA = off;
  if (ppsci_act_has_param(d->activation)) off += d->n_hidden * d->width;
  q->P = off;
  return PPSCI_OK;
}
