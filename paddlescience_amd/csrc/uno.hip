// uno.hip -- the two operations a U-shaped neural operator adds to the FNO block kernels: its blocks change resolution.
//
//   /root/reference/ppsci/arch/unonet.py:246-289        UNO.forward: blocks with output_scaling_factor, horizontal skips
//   /root/reference/ppsci/arch/fno_block.py:779-793     FactorizedSpectralConv.forward: irfftn(out_fft, s = the OUTPUT grid)
//   /root/reference/ppsci/arch/fno_block.py:466-498     resample: F.interpolate(bicubic, align_corners=True) for 2-D planes
//
// (1) irfftn(out_fft, s=(H2, W2)) of a spectrum laid out for an H x W grid crops / zero-pads it at the END of each axis -- rows
//     [0, H2), columns [0, W2/2 + 1) of the UNSHIFTED spectrum, whatever frequency those rows stood for on the input grid -- and
//     then transforms at the new size.  ppsci_spectrum_resize is that crop / pad as a copy between the contraction and the C2R
//     execution (ppsci_fft2d_c2r); the same call with the sizes swapped is its adjoint.  The Hermitian weights of the two real
//     transforms differ where a kept column is the Nyquist column of one grid but not of the other: c_num / c_den rescales
//     those columns on the way back (see uno_engine.UnoNative.backward).
// (2) Bicubic resampling with align_corners is separable and linear: Y = A_h X A_w^T per plane with 4-banded matrices built
//     once on the host (uno_engine.bicubic_matrix).  ppsci_resample2d evaluates the two products with the plane and the
//     intermediate in LDS (one read and one write of HBM per plane); with the transposed matrices it is the adjoint.
//     Planes are a few tens of points a side (16 x 16 ... 64 x 64 Darcy grids): dense small matrices from L2, no gather lists.
#include "ppsci_common.h"
#include "ppsci_hip.h"

extern "C" void ppsci_set_error(const char* fmt, ...);

struct SpecResizeArgs {
  const float* src;  // [n, H, Wf, 2]
  float* dst;        // [n, H2, Wf2, 2]
  long long total;   // n * H2 * Wf2
  int H, Wf, H2, Wf2;
  int w_num, w_den;  // real grid widths of the Hermitian weights c(j) = 1 on the DC / Nyquist column, 2 elsewhere; 0 = no rescaling
};

__device__ __forceinline__ float herm_c(int w_full, int j) { return (j == 0 || 2 * j == w_full) ? 1.f : 2.f; }

__global__ void __launch_bounds__(256) spectrum_resize_kernel(SpecResizeArgs a) {
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < a.total; t += (long long)gridDim.x * 256) {
    const int c = (int)(t % a.Wf2);
    const long long pr = t / a.Wf2;
    const int r = (int)(pr % a.H2);
    const long long p = pr / a.H2;
    float re = 0.f, im = 0.f;
    if (r < a.H && c < a.Wf) {
      const float* s = a.src + ((p * a.H + r) * a.Wf + c) * 2;
      re = s[0];
      im = s[1];
      if (a.w_num > 0) {
        const float f = herm_c(a.w_num, c) / herm_c(a.w_den, c);
        re *= f;
        im *= f;
      }
    }
    a.dst[t * 2] = re;
    a.dst[t * 2 + 1] = im;
  }
}

extern "C" int ppsci_spectrum_resize(int n, int H, int Wf, int H2, int Wf2, int w_num, int w_den, const float* src, float* dst,
                                     void* stream) {
  if (n < 1 || H < 1 || Wf < 1 || H2 < 1 || Wf2 < 1 || !src || !dst || (w_num > 0) != (w_den > 0) || w_num < 0) {
    ppsci_set_error("spectrum_resize: invalid argument");
    return PPSCI_E_INVALID;
  }
  SpecResizeArgs a{src, dst, (long long)n * H2 * Wf2, H, Wf, H2, Wf2, w_num, w_den};
  long long grid = (a.total + 255) / 256;
  if (grid > 8192) grid = 8192;
  PPSCI_LAUNCH(spectrum_resize_kernel, SpecResizeArgs, (int)grid, 256, 0, stream, a);
  if (PPSCI_LAST_LAUNCH_ERROR() != 0) {
    ppsci_set_error("spectrum_resize: launch failed");
    return PPSCI_E_LAUNCH;
  }
  return PPSCI_OK;
}

struct ResampleArgs {
  const float* x;   // [n, H, W]
  const float* Ah;  // [H2, H]
  const float* Aw;  // [W2, W]
  float* y;         // [n, H2, W2]
  int n, H, W, H2, W2, accumulate;
};

// One workgroup per plane (grid-stride over planes): X -> LDS, T = A_h X -> LDS, Y = T A_w^T -> HBM.  Every sum runs over its
// index in ascending order in one thread: results do not depend on the launch geometry.
__global__ void __launch_bounds__(256) resample2d_kernel(ResampleArgs a) {
  PPSCI_DYN_SMEM(smem);
  float* X = smem;
  float* T = smem + a.H * a.W;
  for (int p = blockIdx.x; p < a.n; p += gridDim.x) {
    const float* xp = a.x + (long long)p * a.H * a.W;
    for (int i = threadIdx.x; i < a.H * a.W; i += 256) X[i] = xp[i];
    __syncthreads();
    for (int i = threadIdx.x; i < a.H2 * a.W; i += 256) {
      const int o = i / a.W, w = i - o * a.W;
      const float* ar = a.Ah + (long long)o * a.H;
      float s = 0.f;
#pragma unroll 8
      for (int h = 0; h < a.H; ++h) s += ar[h] * X[h * a.W + w];
      T[i] = s;
    }
    __syncthreads();
    float* yp = a.y + (long long)p * a.H2 * a.W2;
    for (int i = threadIdx.x; i < a.H2 * a.W2; i += 256) {
      const int o = i / a.W2, q = i - o * a.W2;
      const float* ar = a.Aw + (long long)q * a.W;
      const float* tr = T + o * a.W;
      float s = 0.f;
#pragma unroll 8
      for (int w = 0; w < a.W; ++w) s += tr[w] * ar[w];
      yp[i] = a.accumulate ? yp[i] + s : s;
    }
    __syncthreads();  // X / T change hands
  }
}

extern "C" int ppsci_resample2d_supported(int H, int W, int H2, int W2) {
  if (H < 1 || W < 1 || H2 < 1 || W2 < 1) return 0;
  return ((long long)H * W + (long long)H2 * W) * 4 <= PPSCI_LDS_LIMIT_BYTES - 1024 ? 1 : 0;
}

extern "C" int ppsci_resample2d(int n, int H, int W, int H2, int W2, const float* x, const float* Ah, const float* Aw, float* y,
                                int accumulate, void* stream) {
  if (n < 1 || !x || !Ah || !Aw || !y || H < 1 || W < 1 || H2 < 1 || W2 < 1) {
    ppsci_set_error("resample2d: invalid argument");
    return PPSCI_E_INVALID;
  }
  if (!ppsci_resample2d_supported(H, W, H2, W2)) {
    ppsci_set_error("resample2d: a %d x %d plane and its %d x %d intermediate do not fit LDS", H, W, H2, W);
    return PPSCI_E_UNSUPPORTED;
  }
  ResampleArgs a{x, Ah, Aw, y, n, H, W, H2, W2, accumulate ? 1 : 0};
  const int lds = (int)(((long long)H * W + (long long)H2 * W) * 4);
  if (PPSCI_SET_MAX_LDS(resample2d_kernel, lds) != 0) {
    ppsci_set_error("resample2d: cannot raise dynamic LDS to %d B", lds);
    return PPSCI_E_LAUNCH;
  }
  const int grid = n < 4 * PPSCI_NUM_CU ? n : 4 * PPSCI_NUM_CU;
  PPSCI_LAUNCH(resample2d_kernel, ResampleArgs, grid, 256, lds, stream, a);
  if (PPSCI_LAST_LAUNCH_ERROR() != 0) {
    ppsci_set_error("resample2d: launch failed");
    return PPSCI_E_LAUNCH;
  }
  return PPSCI_OK;
}
