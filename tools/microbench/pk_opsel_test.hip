// pk_opsel_test.hip -- does  v_pk_mul_f32 d, a, b op_sel:[0,1]  (low result = a.lo * b.HI) always deliver?
//
// Found in the fused tile kernel (csrc/taylor_fused.inc) on MI355X: with TWO workgroups per CU the low result of exactly this
// instruction came out as (a.lo * 0) in lanes 48..63 of one wave in ~1 of 500 workgroups -- the only packed instruction of
// the kernel with a bare op_sel:[0,1].  This program runs the instruction sequence of that spot in a loop on every wave and
// counts wrong results, with other waves of the CU busy on MFMA / LDS / DPP work, for several variants of the sequence.
//   hipcc --offload-arch=gfx950 -O3 -o pk_opsel_test pk_opsel_test.hip && ./pk_opsel_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned long long u64;
__device__ __forceinline__ u64 pack2(float lo, float hi) { return (u64)__builtin_bit_cast(unsigned, lo) | ((u64)__builtin_bit_cast(unsigned, hi) << 32); }
__device__ __forceinline__ float lo32(u64 v) { return __builtin_bit_cast(float, (unsigned)v); }
__device__ __forceinline__ float hi32(u64 v) { return __builtin_bit_cast(float, (unsigned)(v >> 32)); }

template <int VARIANT>
__global__ void __launch_bounds__(256, 2) probe(const float* __restrict__ in, unsigned* __restrict__ bad, unsigned* __restrict__ badlane, int iters) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float t2 = in[(blockIdx.x * 256 + tid) * 4 + 0], t3 = in[(blockIdx.x * 256 + tid) * 4 + 1];
  const float d2 = in[(blockIdx.x * 256 + tid) * 4 + 2], d3 = in[(blockIdx.x * 256 + tid) * 4 + 3];
  unsigned nbad = 0;
  if ((blockIdx.x & 1) == 0 || VARIANT >= 100) {
    // checker workgroups: the sequence of the kernel
    for (int it = 0; it < iters; ++it) {
      f32x2 r23;
      float a98, a99, a216, a217, k6 = 6.0f * t3;
      if constexpr (VARIANT % 100 == 0) {
        asm volatile(
            "v_mul_f32 %2, 0x40c00000, %6\n\t"           // 6 t2
            "v_mul_f32 %3, -2.0, %6\n\t"                 // a98 = -2 t2
            "v_fma_f32 %2, %6, %2, -2.0\n\t"             // a99 = 6 t2^2 - 2
            "v_mul_f32 %4, -2.0, %7\n\t"                 // a216 = -2 t3
            "v_fma_f32 %5, %7, %8, -2.0\n\t"             // a217 = 6 t3^2 - 2
            "v_mov_b32 %0, %3\n\t"
            "v_mov_b32 %1, %2\n\t"
            : "=&v"(r23[0]), "=&v"(r23[1]), "=&v"(a99), "=&v"(a98), "=&v"(a216), "=&v"(a217)
            : "v"(t2), "v"(t3), "v"(k6));
        const u64 s0 = pack2(a98, a99), s1 = pack2(a216, a217), ddp = pack2(d2, d3);
        u64 o0p, o1p;
        asm volatile(
            "v_pk_mul_f32 %0, %2, %4 op_sel_hi:[1,0]\n\t"   // (a98 * d2, a99 * d2)
            "v_pk_mul_f32 %1, %3, %4 op_sel:[0,1]\n\t"      // (a216 * d3, a217 * d3)
            : "=&v"(o0p), "=&v"(o1p)
            : "v"(s0), "v"(s1), "v"(ddp));
        const float o0[2] = {lo32(o0p), hi32(o0p)}, o1[2] = {lo32(o1p), hi32(o1p)};
        const float e0 = a216 * d3, e1 = a217 * d3;
        if (__builtin_bit_cast(unsigned, e0) != __builtin_bit_cast(unsigned, o1[0]) ||
            __builtin_bit_cast(unsigned, e1) != __builtin_bit_cast(unsigned, o1[1]) ||
            __builtin_bit_cast(unsigned, a98 * d2) != __builtin_bit_cast(unsigned, o0[0]) ||
            __builtin_bit_cast(unsigned, a99 * d2) != __builtin_bit_cast(unsigned, o0[1]))
          ++nbad;
      } else {
        // the same through the compiler (whatever it selects)
        const float s1[2] = {-2.0f * t3, t3 * k6 - 2.0f};
        u64 o1p;
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(o1p) : "v"(pack2(s1[0], s1[1])), "v"(pack2(d2, d3)));
        const float o1[2] = {lo32(o1p), hi32(o1p)};
        if (__builtin_bit_cast(unsigned, s1[0] * d3) != __builtin_bit_cast(unsigned, o1[0]) ||
            __builtin_bit_cast(unsigned, s1[1] * d3) != __builtin_bit_cast(unsigned, o1[1]))
          ++nbad;
      }
    }
  } else {
    // noise workgroups: MFMA + LDS + DPP traffic on the other wave slot of every SIMD
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    u32x4 av = {__builtin_bit_cast(unsigned, t2), __builtin_bit_cast(unsigned, t3), 0x3f803f80u, 0x3f803f80u};
    for (int it = 0; it < iters; ++it) {
      lds[tid * 4 + (it & 3)] = acc[0];
      __syncthreads();
      const f32x4 v = *(const f32x4*)&lds[((tid + 17) & 255) * 4];
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, av), acc, 0, 0, 0);
      acc[1] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[0]), 0x111, 0xf, 0xf, true));
      acc[2] += v[1] * 1e-9f;
    }
    if (acc[0] == 12345.f) nbad = 1u << 31;
  }
  if (nbad) {
    atomicAdd(bad, nbad & 0x7fffffffu);
    atomicAdd(&badlane[lane], nbad & 0x7fffffffu);
  }
}

// FORM 0: v_pk_mul_f32 op_sel:[0,1] (b.HI to both results); 1: op_sel_hi:[1,0] (b.LO to both); 2: v_pk_fma_f32 op_sel:[0,1,0];
// 3: v_pk_mul_f32 op_sel:[1,0] (a.HI to both)
template <int FORM>
__global__ void __launch_bounds__(256, 2) probe_loads(const float* __restrict__ in, const float* __restrict__ far, unsigned* __restrict__ bad,
                                                       unsigned* __restrict__ badlane, int rounds, int inner, unsigned mask) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const float a0 = in[(blockIdx.x * 256 + tid) * 4 + 0], a1 = in[(blockIdx.x * 256 + tid) * 4 + 1];
  const float b0 = in[(blockIdx.x * 256 + tid) * 4 + 2], b1 = in[(blockIdx.x * 256 + tid) * 4 + 3];
  const u64 A = pack2(a0, a1), B = pack2(b0, b1);
  float e0, e1;
  if (FORM == 0) e0 = a0 * b1, e1 = a1 * b1;
  if (FORM == 1) e0 = a0 * b0, e1 = a1 * b0;
  if (FORM == 2) e0 = __builtin_fmaf(a0, b1, 1.0f), e1 = __builtin_fmaf(a1, b1, 1.0f);
  if (FORM == 3) e0 = a1 * b0, e1 = a1 * b1;
  if (FORM == 4 || FORM == 5) e0 = a0 * b1, e1 = a1 * b1;
  unsigned nbad = 0, idx = (blockIdx.x * 2654435761u + tid * 40503u) & mask;
  float sink = 0.f;
  for (int r = 0; r < rounds; ++r) {
    // three scattered loads in flight (as the kernel's input prefetch), partly masked
    float x0 = 0.f, x1 = 0.f, x2 = 0.f;
    if (lane & 1) x0 = far[idx];
    if (lane & 2) x1 = far[(idx * 7u + 12345u) & mask];
    x2 = far[(idx * 13u + 777u) & mask];
    for (int it = 0; it < inner; ++it) {
      u64 o;
      if (FORM == 0) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(o) : "v"(A), "v"(B));
      if (FORM == 1) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(o) : "v"(A), "v"(B));
      if (FORM == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, 1.0 op_sel:[0,1,0]" : "=v"(o) : "v"(A), "v"(B));
      if (FORM == 3) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(o) : "v"(A), "v"(B));
      if (FORM == 4 || FORM == 5) {
        // the kernel's pair: X is src0 of the first multiply and the DESTINATION of the second (write after read, back to back)
        u64 X = A, d1;
        float y0, y1;
        if (FORM == 4)
          asm volatile("v_mul_f32 %2, -2.0, %5\n\t"
                       "v_fma_f32 %3, %5, %6, -2.0\n\t"
                       "v_pk_mul_f32 %1, %0, %4 op_sel_hi:[1,0]\n\t"
                       "v_pk_mul_f32 %0, %[Y], %4 op_sel:[0,1]"
                       : "+v"(X), "=&v"(d1), "=&v"(y0), "=&v"(y1)
                       : "v"(B), "v"(a1), "v"(b0), [Y] "v"(A));
        else
          asm volatile("v_mul_f32 %2, -2.0, %5\n\t"
                       "v_fma_f32 %3, %5, %6, -2.0\n\t"
                       "v_pk_mul_f32 %1, %0, %4 op_sel_hi:[1,0]\n\t"
                       "s_nop 4\n\t"
                       "v_pk_mul_f32 %0, %[Y], %4 op_sel:[0,1]"
                       : "+v"(X), "=&v"(d1), "=&v"(y0), "=&v"(y1)
                       : "v"(B), "v"(a1), "v"(b0), [Y] "v"(A));
        o = X;
        if (__builtin_bit_cast(unsigned, lo32(d1)) != __builtin_bit_cast(unsigned, a0 * b0) ||
            __builtin_bit_cast(unsigned, hi32(d1)) != __builtin_bit_cast(unsigned, a1 * b0))
          ++nbad;
        sink += y0 + y1;
      }
      if (__builtin_bit_cast(unsigned, lo32(o)) != __builtin_bit_cast(unsigned, e0) ||
          __builtin_bit_cast(unsigned, hi32(o)) != __builtin_bit_cast(unsigned, e1))
        ++nbad;
    }
    sink += x0 + x1 + x2;
    idx = (idx * 1103515245u + 12345u + (unsigned)r) & mask;
  }
  if (sink == 12345.678f) nbad |= 1u << 30;
  if (nbad) {
    atomicAdd(bad, nbad);
    atomicAdd(&badlane[lane], nbad);
  }
}

template <int FORM>
static void run_loads(const char* name, const float* din, const float* dfar, unsigned mask, unsigned* dbad) {
  hipMemset(dbad, 0, 65 * 4);
  hipFuncSetAttribute((const void*)probe_loads<FORM>, hipFuncAttributeMaxDynamicSharedMemorySize, 60 * 1024);
  for (int rep = 0; rep < 10; ++rep)
    hipLaunchKernelGGL(probe_loads<FORM>, dim3(512), dim3(256), 60 * 1024, 0, din, dfar, dbad, dbad + 1, 400, 64, mask);
  hipDeviceSynchronize();
  unsigned h[65];
  hipMemcpy(h, dbad, sizeof(h), hipMemcpyDeviceToHost);
  unsigned q[4] = {0, 0, 0, 0};
  for (int l = 0; l < 64; ++l) q[l / 16] += h[1 + l];
  printf("%-44s wrong results: %u of %.3g  (per 16-lane group: %u %u %u %u)\n", name, h[0], 10.0 * 512 * 256 * 400 * 64, q[0], q[1], q[2], q[3]);
}

template <int VARIANT>
static void run(const char* name, const float* din, unsigned* dbad, int iters) {
  hipMemset(dbad, 0, 65 * 4);
  hipFuncSetAttribute((const void*)probe<VARIANT>, hipFuncAttributeMaxDynamicSharedMemorySize, 60 * 1024);
  for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL(probe<VARIANT>, dim3(512), dim3(256), 60 * 1024, 0, din, dbad, dbad + 1, iters);
  hipDeviceSynchronize();
  unsigned h[65];
  hipMemcpy(h, dbad, sizeof(h), hipMemcpyDeviceToHost);
  unsigned q[4] = {0, 0, 0, 0};
  for (int l = 0; l < 64; ++l) q[l / 16] += h[1 + l];
  printf("%-44s wrong results: %u  (per 16-lane group: %u %u %u %u)\n", name, h[0], q[0], q[1], q[2], q[3]);
}

int main() {
  const int n = 512 * 256 * 4;
  std::vector<float> h(n);
  unsigned s = 12345u;
  for (int i = 0; i < n; ++i) {
    s = s * 1664525u + 1013904223u;
    h[i] = ((s >> 8) & 0xffff) / 65536.0f * 1.8f - 0.9f;
  }
  float* din;
  unsigned* dbad;
  hipMalloc(&din, n * 4);
  hipMalloc(&dbad, 65 * 4);
  hipMemcpy(din, h.data(), n * 4, hipMemcpyHostToDevice);
  run<0>("kernel sequence, noise workgroups beside", din, dbad, 20000);
  run<100>("kernel sequence, every workgroup checks", din, dbad, 20000);
  run<1>("single instruction, noise beside", din, dbad, 20000);
  run<101>("single instruction, every workgroup checks", din, dbad, 20000);
  const unsigned mask = (64u << 20) - 1;  // 256 MiB of floats
  float* dfar;
  hipMalloc(&dfar, ((size_t)mask + 1) * 4);
  hipMemset(dfar, 0, ((size_t)mask + 1) * 4);
  run_loads<0>("loads in flight: v_pk_mul op_sel:[0,1]", din, dfar, mask, dbad);
  run_loads<1>("loads in flight: v_pk_mul op_sel_hi:[1,0]", din, dfar, mask, dbad);
  run_loads<2>("loads in flight: v_pk_fma op_sel:[0,1,0]", din, dfar, mask, dbad);
  run_loads<3>("loads in flight: v_pk_mul op_sel:[1,0]", din, dfar, mask, dbad);
  run_loads<4>("pair: 2nd pk_mul writes the 1st's src0", din, dfar, mask, dbad);
  run_loads<5>("the same with s_nop 4 between them", din, dfar, mask, dbad);
  return 0;
}
