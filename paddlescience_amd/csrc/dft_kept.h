// dft_kept.h -- the forward transform on the kept modes (see spectral_conv.hip, "kept modes only") as device functions, so
// that a kernel which has just PRODUCED a plane (the block tail's apply kernels in fno.hip) can transform it from LDS
// instead of storing it for a separate transform launch to read back.
#pragma once
#include "ppsci_common.h"

#ifndef PPSCI_F32X2_DEFINED
#define PPSCI_F32X2_DEFINED
typedef float f32x2 __attribute__((ext_vector_type(2)));
#endif

struct DftArgs {
  const float* src;
  float* dst;
  const float* tab;  // twiddles tw [W][my][2] | th [H][mx][2], computed once per shape on the host in double (a "plan")
  int n, H, W, mx, my, c0, rows;
  // inverse only: also the row sums of the GroupNorm that follows (sum of y + sbias[c], sum of its square -> rows_out[p][4])
  const float* sbias;
  float* rows_out;
  int C;
};

// LDS (floats): tw [W][my][2] | th [H][mx][2] | plane [H*W] (fwd) or Z [mx*my*2] (inv) | T [H][my][2] | fwd: partial sums
__device__ __forceinline__ void dft_twiddles(const DftArgs& a, float* tw) {
  const int nt = 2 * (a.W * a.my + a.H * a.mx);
  for (int idx = threadIdx.x; idx < nt; idx += blockDim.x) tw[idx] = a.tab[idx];
}

// The twiddle table of a shape (device memory, built once on the host; spectral_conv.hip) or null
const float* ppsci_dft_table(int H, int W, int mx, int my, int rows);

// floats of LDS the forward stages need: tw | th | plane [H][W + 1] | T [H][my][2] | partial sums
__host__ __device__ inline long long dft_fwd_lds_floats(int H, int W, int mx, int my) {
  return 2LL * W * my + 2LL * H * mx + (long long)H * (W + 1) + 2LL * H * my + 2LL * (mx * my > 256 ? mx * my : 256);
}

// Plane `pl` ([H][W + 1] in LDS, written and synchronised by the caller; tw / th filled by dft_twiddles) -> its kept modes
// X [mx][my][2] at `dst`.  256 threads, contains three barriers, all threads must call it.
__device__ __forceinline__ void dft_fwd_stages(const DftArgs& a, const float* tw, const float* th, const float* pl, float* T,
                                               float* dst) {
  const int tid = threadIdx.x, ldp = a.W + 1;
  // rows: T[h][q] = sum_w x[h][w] e^{-2 pi i w q / W}
  const f32x2* th2 = (const f32x2*)th;
  f32x2* T2 = (f32x2*)T;
  {
    // [H x W] . [W x 2 my] on the fp32 MFMA (16 x 16 x 4): rows h, columns 2 q + {re, im}, k = w.  The column index of the
    // D tile is the float index inside row h of T ([h][my][2]), so the result rows are stored as they come.
    const int lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int nrb = (a.H + 15) / 16, ncol = 2 * a.my, nnb = (ncol + 15) / 16;
    for (int it = wave; it < nrb * nnb; it += 4) {
      const int rb = it / nnb, nb = it - rb * nnb;
      const int h = 16 * rb + c, col = 16 * nb + c;
      const bool hok = h < a.H, cok = col < ncol;
      const float* arow = pl + (hok ? h : 0) * ldp;
      const float* bcol = tw + (cok ? col : 0);  // tw[w][my][2] = row w of 2 my floats: (cos, sin) pairs
      const float bs = (col & 1) ? -1.f : 1.f;   // e^{-i phi} = cos - i sin
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int w0 = 0; w0 < a.W; w0 += 4) {
        const int w = w0 + g;
        const bool wok = w < a.W;
        const float av = (hok && wok) ? arow[w] : 0.f;
        const float bv = (cok && wok) ? bs * bcol[w * ncol] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
      }
      if (cok) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int hh = 16 * rb + 4 * g + rr;
          if (hh < a.H) T[hh * ncol + col] = acc[rr];
        }
      }
    }
  }
  __syncthreads();
  // columns: X[m][q] = sum_h T[h][q] e^{-2 pi i h k_m / H}; the h range in `parts` pieces (mx * my outputs are fewer than
  // the workgroup's threads), combined in a fixed order through LDS
  const int nm = a.mx * a.my;
  const int parts = nm < 256 ? (256 / nm < a.H ? 256 / nm : a.H) : 1;
  f32x2* part = (f32x2*)(T + 2 * a.H * a.my);  // [parts][nm], parts * nm <= 256 (or one part of nm)
  for (int it = tid; it < parts * nm || (parts == 1 && it < nm); it += 256) {
    const int pt = it / nm, idx = it - pt * nm;
    const int m = idx / a.my, q = idx - m * a.my;
    const int h0 = (int)((long long)a.H * pt / parts), h1 = (int)((long long)a.H * (pt + 1) / parts);
    float re = 0.f, im = 0.f;
#pragma unroll 4
    for (int h = h0; h < h1; ++h) {
      const f32x2 t = T2[h * a.my + q], e = th2[h * a.mx + m];
      re += t[0] * e[0] + t[1] * e[1];
      im += t[1] * e[0] - t[0] * e[1];
    }
    part[it] = (f32x2){re, im};
  }
  __syncthreads();
  for (int idx = tid; idx < nm; idx += 256) {
    f32x2 acc = part[idx];
    for (int pt = 1; pt < parts; ++pt) acc += part[pt * nm + idx];
    *(f32x2*)(dst + 2 * idx) = acc;
  }
}

// floats of LDS of the inverse stages: tw | th | Z [mx][my][2] | T [H][my][2]   (+ the caller's own output plane, if in LDS)
__host__ __device__ inline long long dft_inv_lds_floats(int H, int W, int mx, int my) {
  return 2LL * W * my + 2LL * H * mx + 2LL * mx * my + 2LL * H * my;
}

// Kept modes Z ([mx][my][2] in LDS, written and synchronised by the caller) -> the plane y[h * ldo + w] (global memory or
// LDS).  256 threads, one barrier inside; the caller synchronises before it reads y from LDS.
template <typename F>
__device__ __forceinline__ void dft_inv_stages(const DftArgs& a, const float* tw, const float* th, const float* Z, float* T,
                                               F&& emit) {
  const int tid = threadIdx.x;
  // columns: T[h][q] = sum_m Z[m][q] e^{+2 pi i h k_m / H}
  const f32x2* th2 = (const f32x2*)th;
  const f32x2* Z2 = (const f32x2*)Z;
  f32x2* T2 = (f32x2*)T;
  for (int idx = tid; idx < a.H * a.my; idx += 256) {
    const int h = idx / a.my, q = idx - h * a.my;
    float re = 0.f, im = 0.f;
#pragma unroll 4
    for (int m = 0; m < a.mx; ++m) {
      const f32x2 z = Z2[m * a.my + q], e = th2[h * a.mx + m];
      re += z[0] * e[0] - z[1] * e[1];
      im += z[0] * e[1] + z[1] * e[0];
    }
    const float c = (q == 0 || 2 * q == a.W) ? 1.f : 2.f;  // the Hermitian weight of column q, folded in here
    T2[idx] = (f32x2){c * re, c * im};
  }
  __syncthreads();
  // rows: y[h][w] = sum_q c(q) Re(T[h][q] e^{+2 pi i w q / W}),  c = 1 on the DC / Nyquist column, else 2
  //   = [H x 2 my] . [2 my x W] on the fp32 MFMA: k = 2 q + {re, im} (the float index inside a row of T), B = (cos, -sin)
  const int lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
  const int nrb = (a.H + 15) / 16, nnb = (a.W + 15) / 16, ncol = 2 * a.my;
  for (int it = wave; it < nrb * nnb; it += 4) {
    const int rb = it / nnb, nb = it - rb * nnb;
    const int h = 16 * rb + c, w = 16 * nb + c;
    const bool hok = h < a.H, wok = w < a.W;
    const float* arow = T + (hok ? h : 0) * ncol;
    const float* brow = tw + (wok ? w : 0) * ncol;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < ncol; k0 += 4) {
      const int k = k0 + g;
      const bool kok = k < ncol;
      const float av = (hok && kok) ? arow[k] : 0.f;
      const float bv = (wok && kok) ? ((k & 1) ? -brow[k] : brow[k]) : 0.f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
    }
    if (wok) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int hh = 16 * rb + 4 * g + rr;
        if (hh < a.H) emit(hh, w, acc[rr]);
      }
    }
  }
}
