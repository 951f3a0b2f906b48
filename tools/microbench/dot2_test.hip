// dot2_test.hip -- does v_dot2c_f32_bf16 compute  x - bf16(x)  exactly, and what does an inline constant operand mean?
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/dot2_test tools/microbench/dot2_test.hip && tools/microbench/dot2_test
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ unsigned cvt_pk(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf16x2));
}
__global__ void k(const float* a, const float* b, float* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = a[i], y = b[i];
  const unsigned h = cvt_pk(x, y);
  // reference: unpack + subtract
  const float rx = x - __builtin_bit_cast(float, h << 16), ry = y - __builtin_bit_cast(float, h & 0xffff0000u);
  // (1) builtin with literal constants (the compiler may fold them into an inline operand)
  const float d1x = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, h), __builtin_bit_cast(bf16x2, 0x0000bf80u), x, false);
  const float d1y = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, h), __builtin_bit_cast(bf16x2, 0xbf800000u), y, false);
  // (2) the same with the constants hidden from the optimiser (VGPR operands)
  unsigned klo = 0x0000bf80u, khi = 0xbf800000u;
  asm volatile("" : "+v"(klo));
  asm volatile("" : "+v"(khi));
  const float d2x = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, h), __builtin_bit_cast(bf16x2, klo), x, false);
  const float d2y = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, h), __builtin_bit_cast(bf16x2, khi), y, false);
  out[6 * i + 0] = rx, out[6 * i + 1] = ry, out[6 * i + 2] = d1x, out[6 * i + 3] = d1y, out[6 * i + 4] = d2x, out[6 * i + 5] = d2y;
}
int main() {
  const int n = 1 << 16;
  float *ha = (float*)malloc(4 * n), *hb = (float*)malloc(4 * n), *ho = (float*)malloc(24 * n);
  srand(1);
  for (int i = 0; i < n; ++i) {
    const float s = (i % 7 == 0) ? 1e-6f : ((i % 11 == 0) ? 300.f : 1.f);
    ha[i] = s * (2.f * rand() / RAND_MAX - 1.f);
    hb[i] = s * (2.f * rand() / RAND_MAX - 1.f);
  }
  float *da, *db, *dout;
  hipMalloc(&da, 4 * n), hipMalloc(&db, 4 * n), hipMalloc(&dout, 24 * n);
  hipMemcpy(da, ha, 4 * n, hipMemcpyHostToDevice), hipMemcpy(db, hb, 4 * n, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, db, dout, n);
  hipMemcpy(ho, dout, 24 * n, hipMemcpyDeviceToHost);
  int bad1 = 0, bad2 = 0;
  for (int i = 0; i < n; ++i) {
    if (memcmp(&ho[6 * i], &ho[6 * i + 2], 8)) ++bad1;
    if (memcmp(&ho[6 * i], &ho[6 * i + 4], 8)) ++bad2;
  }
  printf("{\"n\": %d, \"mismatch_literal_constants\": %d, \"mismatch_register_constants\": %d, \"sample\": [%g, %g, %g, %g, %g, %g, %g, %g]}\n", n,
         bad1, bad2, ha[1], hb[1], ho[6], ho[7], ho[8], ho[9], ho[10], ho[11]);
  return 0;
}
