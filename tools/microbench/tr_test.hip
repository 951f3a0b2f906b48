#include <hip/hip_runtime.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
// one wave: writes a T-layout bf16 block plane (lane (g,c): 4 values = features 4g+r at point c) and reads it back transposed
__global__ void k(unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[16 * 16];
  const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
  // element value encodes (feature, point): feature*16 + point
  unsigned short v[4];
  for (int r = 0; r < 4; ++r) v[r] = (unsigned short)((4 * g + r) * 16 + c);
  // chunk (fgroup g, point c) at byte (c*4 + g)*8
  u32x2 w = {(unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16)};
  *(u32x2*)&lds[(c * 4 + g) * 4] = w;
  __syncthreads();
  // reader lane (g', i): address = chunk(fgroup = i%4, point = 4g' + i/4)
  const int i = c;
  const unsigned short* p = &lds[((4 * g + i / 4) * 4 + (i % 4)) * 4];
  s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (unsigned short)t[j];
}
int main() {
  unsigned* d; hipMalloc(&d, 1024);
  k<<<1, 64>>>(d);
  unsigned h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
    int gg = l >> 4, cc = l & 15; // want feature cc, point 4gg + j
    unsigned want = cc * 16 + 4 * gg + j;
    if (h[l * 4 + j] != want) { if (bad < 8) printf("lane %d j %d got (f %u,p %u) want (f %u,p %u)\n", l, j, h[l*4+j] / 16, h[l*4+j] % 16, want / 16, want % 16); ++bad; }
  }
  printf("tr16 transpose: %d mismatches\n", bad);
  return 0;
}
