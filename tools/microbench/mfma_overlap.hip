// Does v_mfma_f32_16x16x4_f32 (fp32 inputs) overlap with independent VALU work of the SAME wave on gfx950?
// Each kernel runs ITER iterations of 8 MFMAs (4 independent accumulators) with K independent v_fma_f32
// behind every MFMA; one wave per SIMD (256 threads per block, 1 block per CU).  Prints cycles per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

#define REP8(x) x x x x x x x x

template <int K, int KIND>
__global__ void __launch_bounds__(256) kern(float* out, int iters) {
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  float x = threadIdx.x * 1e-3f, y = 1.0001f;
  float v0 = x, v1 = x + 1, v2 = x + 2, v3 = x + 3, v4 = x + 4, v5 = x + 5, v6 = x + 6, v7 = x + 7;
  s16x4 bx = {1, 2, 3, 4};
  for (int it = 0; it < iters; ++it) {
#define VAL(n) if (K > n) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v##n) : "v"(y));
#define VALS VAL(0) VAL(1) VAL(2) VAL(3) VAL(4) VAL(5) VAL(6) VAL(7)
#define MF(acc)                                                                                    \
  if (KIND == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y)); \
  if (KIND == 1) asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %1, %0" : "+v"(acc) : "v"(bx));      \
  VALS
    MF(a0) MF(a1) MF(a2) MF(a3) MF(a0) MF(a1) MF(a2) MF(a3)
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
}

template <int K, int KIND>
void run(const char* name, float* d, int blocks_per_cu) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  kern<K, KIND><<<256 * blocks_per_cu, 256>>>(d, 100);
  hipEventRecord(e0);
  kern<K, KIND><<<256 * blocks_per_cu, 256>>>(d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double per = ms * 1e-3 / ((double)iters * 8 * blocks_per_cu);  // seconds per MFMA per SIMD
  printf("%s K=%d waves/SIMD=%d: %.3f ms, %.2f ns per MFMA slot (%.1f cycles @2.4GHz)\n", name, K, blocks_per_cu, ms, per * 1e9,
         per * 2.4e9);
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 2 * 256 * 4);
  run<0, 0>("f32  16x16x4 ", d, 1);
  run<2, 0>("f32  16x16x4 ", d, 1);
  run<4, 0>("f32  16x16x4 ", d, 1);
  run<6, 0>("f32  16x16x4 ", d, 1);
  run<8, 0>("f32  16x16x4 ", d, 1);
  run<0, 0>("f32  16x16x4 ", d, 2);
  run<4, 0>("f32  16x16x4 ", d, 2);
  run<8, 0>("f32  16x16x4 ", d, 2);
  run<0, 1>("bf16 16x16x16", d, 1);
  run<2, 1>("bf16 16x16x16", d, 1);
  run<4, 1>("bf16 16x16x16", d, 1);
  run<8, 1>("bf16 16x16x16", d, 1);
  return 0;
}
