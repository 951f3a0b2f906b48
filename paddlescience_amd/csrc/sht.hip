// sht.hip -- the spherical-harmonic transform pair and the per-degree contraction of the spherical FNO.
//
//   /root/reference/ppsci/arch/sfnonet.py:322-360                   SphericalConv.forward: sht -> contraction -> isht (+ bias)
//   /root/reference/ppsci/arch/paddle_harmonics/sht.py:118-150      RealSHT.forward         rfft(lon) * 2 pi, then the Legendre quadrature
//   /root/reference/ppsci/arch/paddle_harmonics/sht.py:216-232      InverseRealSHT.forward  Legendre synthesis, then irfft(n = nlon)
//   /root/reference/ppsci/arch/sfnonet.py:45-74                     _contract_dense_trick(dhconv=True): weights per DEGREE l
//
// On an nlat x nlon plane (colatitude k, longitude j), degrees l < L, orders m < M (M <= nlon/2 + 1):
//   analysis   X[l][m] = sum_k leg[m][l][k] T[k][m],          T[k][m] = sum_j x[k][j] e^{-2 pi i j m / nlon}
//   synthesis  y[k][j] = sum_m Re( T[k][m] e^{+2 pi i j m / nlon} ),   T[k][m] = sum_l leg[m][l][k] Z[l][m]
// `leg` is a REAL table of (m, l, k) built on the host (arch/sht_tables.py): quadrature weights and the rfft scaling for the
// forward transform, the Hermitian weights of the real inverse DFT for the inverse transform.  Both maps are linear with real
// tables, so with L = sum(...) real:
//   adjoint of synthesis:  dL/dZ[l][m] = sum_k leg[m][l][k] ( sum_j g[k][j] e^{-i ..} )   = the ANALYSIS kernel on the synthesis table
//   adjoint of analysis:   dL/dx[k][j] = sum_m Re( ( sum_l leg[m][l][k] G[l][m] ) e^{+i ..} ) = the SYNTHESIS kernel on the analysis table
// (G, dL/dZ: gradient w.r.t. real and imaginary part as one complex number).  Two kernels serve the four transforms of a training step.
// The table is stored so that the threads of a wave read consecutive addresses: [nlat][L][M] for the analysis kernel (threads run
// over (l, m), the loop over k), [L][nlat][M] for the synthesis kernel (threads over (k, m), the loop over l).  (Read as [M][L][nlat],
// one cache line per thread, the kernels took 18-20 us per launch at the reference's shape.)
//
// One workgroup per plane, the plane / the coefficients and the intermediate T in LDS (a 64 x 128 plane with 16 orders: 56 KB), the
// longitude twiddles in LDS, the Legendre table from L2 (shared by every plane of the launch).  The grids of the reference's SFNO
// example are 32 x 64 and 64 x 128 points with 32 x 16 coefficients: ~0.3 MFLOP per plane, launch-latency-sized; plain fp32 FMA loops,
// every sum in one thread in ascending index order (results do not depend on the launch geometry).
#include "ppsci_common.h"
#include "ppsci_hip.h"

extern "C" void ppsci_set_error(const char* fmt, ...);

#define SHT_BATCH 16

struct ShtArgs {
  const float* src;
  float* dst;
  const float* tw;   // [W][M][2] (cos, sin)(2 pi j m / W)
  const float* leg;  // analysis: [H][L][M]; synthesis: [L][H][M]
  int n, H, W, L, M;
  // synthesis only: the per-degree contraction in front of it (cx != null; src unused): plane p = (b, c),
  // Z[l][m] = sum_s cx[b][s][l][m] * w[s][c][l]  (conj_t: * conj(w[c][s][l])) -- ppsci_sht_contract's sum, in its order
  const float* cx;
  const float* wr;
  const float* wi;
  int Ci, Co, conj_t;
};

static long long sht_lds_floats(int H, int W, int L, int M) {
  const long long plane = (long long)H * W, coef = 2LL * L * M;
  return 2LL * W * M + 2LL * H * M + (plane > coef ? plane : coef);
}

__global__ void __launch_bounds__(256) sht_analysis_kernel(ShtArgs a) {
  PPSCI_DYN_SMEM(smem);
  float* tw = smem;                    // [W][M][2]
  float* T = tw + 2 * a.W * a.M;       // [H][M][2]
  float* pl = T + 2 * a.H * a.M;       // [H][W]
  const int tid = threadIdx.x;
  for (int i = tid; i < 2 * a.W * a.M; i += 256) tw[i] = a.tw[i];
  for (int p = blockIdx.x; p < a.n; p += gridDim.x) {
    const float* x = a.src + (long long)p * a.H * a.W;
    __syncthreads();  // (the previous plane's readers are done; the twiddles are in place)
    for (int i = tid; i < a.H * a.W; i += 256) pl[i] = x[i];
    __syncthreads();
    for (int i = tid; i < a.H * a.M; i += 256) {
      const int k = i / a.M, m = i - k * a.M;
      const float* row = pl + k * a.W;
      float re = 0.f, im = 0.f;
#pragma unroll 8
      for (int j = 0; j < a.W; ++j) {
        const float v = row[j];
        re += v * tw[2 * (j * a.M + m)];
        im -= v * tw[2 * (j * a.M + m) + 1];
      }
      T[2 * i] = re;
      T[2 * i + 1] = im;
    }
    __syncthreads();
    float* X = a.dst + (long long)p * a.L * a.M * 2;
    for (int i = tid; i < a.L * a.M; i += 256) {
      const int l = i / a.M, m = i - l * a.M;
      const float* lg = a.leg + i;  // [k][l][m]: element (k, l, m) at k * L * M + i
      const int lm = a.L * a.M;
      float re = 0.f, im = 0.f;
      for (int k0 = 0; k0 < a.H; k0 += SHT_BATCH) {  // SHT_BATCH table loads in flight (one per iteration: an L2 round trip each)
        float w[SHT_BATCH];
#pragma unroll
        for (int u = 0; u < SHT_BATCH; ++u) w[u] = k0 + u < a.H ? lg[(long long)(k0 + u) * lm] : 0.f;
#pragma unroll
        for (int u = 0; u < SHT_BATCH; ++u) {
          const int k = k0 + u < a.H ? k0 + u : 0;  // (w = 0 behind the end)
          re += w[u] * T[2 * (k * a.M + m)];
          im += w[u] * T[2 * (k * a.M + m) + 1];
        }
      }
      X[2 * i] = re;
      X[2 * i + 1] = im;
    }
  }
}

template <bool CONTRACT>  // (its own instantiation: as a run-time branch the unused contraction cost the plain kernel 1.1 us per launch)
__global__ void __launch_bounds__(256) sht_synthesis_kernel(ShtArgs a) {
  PPSCI_DYN_SMEM(smem);
  float* tw = smem;
  float* T = tw + 2 * a.W * a.M;
  float* Z = T + 2 * a.H * a.M;  // [L][M][2]
  const int tid = threadIdx.x;
  for (int i = tid; i < 2 * a.W * a.M; i += 256) tw[i] = a.tw[i];
  for (int p = blockIdx.x; p < a.n; p += gridDim.x) {
    __syncthreads();
    if constexpr (CONTRACT) {
      const int lm = a.L * a.M;
      const int Cp = a.conj_t ? a.Ci : a.Co, Cs = a.conj_t ? a.Co : a.Ci;
      const int b = p / Cp, c = p - b * Cp;
      for (int i = tid; i < lm; i += 256) {
        const int l = i / a.M;
        float sr = 0.f, si = 0.f;
#pragma unroll 8
        for (int s = 0; s < Cs; ++s) {
          const float* xp = a.cx + (((long long)b * Cs + s) * lm + i) * 2;
          const long long wi_ = (a.conj_t ? (long long)c * a.Co + s : (long long)s * a.Co + c) * a.L + l;
          const float xr = xp[0], xi = xp[1], wr = a.wr[wi_], wim = a.conj_t ? -a.wi[wi_] : a.wi[wi_];
          sr += xr * wr - xi * wim;
          si += xr * wim + xi * wr;
        }
        Z[2 * i] = sr;
        Z[2 * i + 1] = si;
      }
    } else {
      const float* z = a.src + (long long)p * a.L * a.M * 2;
      for (int i = tid; i < 2 * a.L * a.M; i += 256) Z[i] = z[i];
    }
    __syncthreads();
    for (int i = tid; i < a.H * a.M; i += 256) {
      const int k = i / a.M, m = i - k * a.M;
      const float* lg = a.leg + i;  // [l][k][m]: element (l, k, m) at l * H * M + i
      const int hm = a.H * a.M;
      float re = 0.f, im = 0.f;
      for (int l0 = 0; l0 < a.L; l0 += SHT_BATCH) {
        float w[SHT_BATCH];
#pragma unroll
        for (int u = 0; u < SHT_BATCH; ++u) w[u] = l0 + u < a.L ? lg[(long long)(l0 + u) * hm] : 0.f;
#pragma unroll
        for (int u = 0; u < SHT_BATCH; ++u) {
          const int l = l0 + u < a.L ? l0 + u : 0;
          re += w[u] * Z[2 * (l * a.M + m)];
          im += w[u] * Z[2 * (l * a.M + m) + 1];
        }
      }
      T[2 * i] = re;
      T[2 * i + 1] = im;
    }
    __syncthreads();
    float* y = a.dst + (long long)p * a.H * a.W;
    if (256 % a.W == 0 && a.M <= 16) {
      // every output of this thread has the same longitude j = tid % W: its twiddles stay in registers, the T row of a latitude is
      // read by all lanes of that latitude at one address (LDS broadcast)
      const int j = tid % a.W;
      float ec[16], es[16];
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        ec[m] = m < a.M ? tw[2 * (j * a.M + m)] : 0.f;
        es[m] = m < a.M ? tw[2 * (j * a.M + m) + 1] : 0.f;
      }
      for (int i = tid; i < a.H * a.W; i += 256) {
        const float* t = T + 2 * (i / a.W) * a.M;
        float s = 0.f;
#pragma unroll
        for (int m = 0; m < 16; ++m)
          if (m < a.M) s += t[2 * m] * ec[m] - t[2 * m + 1] * es[m];
        y[i] = s;
      }
    } else {
      for (int i = tid; i < a.H * a.W; i += 256) {
        const int k = i / a.W, j = i - k * a.W;
        const float* t = T + 2 * k * a.M;
        const float* e = tw + 2 * j * a.M;
        float s = 0.f;
        for (int m = 0; m < a.M; ++m) s += t[2 * m] * e[2 * m] - t[2 * m + 1] * e[2 * m + 1];
        y[i] = s;
      }
    }
  }
}

extern "C" int ppsci_sht_supported(int H, int W, int L, int M) {
  if (H < 2 || W < 2 || L < 1 || M < 1 || M > W / 2 + 1) return 0;
  return 4 * sht_lds_floats(H, W, L, M) <= PPSCI_LDS_LIMIT_BYTES - 1024 ? 1 : 0;
}

static int sht_run(int synthesis, int n, int H, int W, int L, int M, const float* tw, const float* leg, const float* src, float* dst,
                   void* stream, const float* cx = nullptr, const float* wr = nullptr, const float* wi = nullptr, int Ci = 0, int Co = 0,
                   int conj_t = 0) {
  if (n < 1 || !tw || !leg || (!src && !cx) || !dst || !ppsci_sht_supported(H, W, L, M)) {
    ppsci_set_error("sht: invalid argument or a %d x %d plane with %d x %d coefficients does not fit LDS", H, W, L, M);
    return PPSCI_E_INVALID;
  }
  ShtArgs a{src, dst, tw, leg, n, H, W, L, M, cx, wr, wi, Ci, Co, conj_t};
  const int lds = (int)(4 * sht_lds_floats(H, W, L, M));
  const int grid = n < 8 * PPSCI_NUM_CU ? n : 8 * PPSCI_NUM_CU;
  int se;
  if (synthesis && cx != nullptr) {
    se = PPSCI_SET_MAX_LDS(sht_synthesis_kernel<true>, lds);
    if (se == 0) PPSCI_LAUNCH(sht_synthesis_kernel<true>, ShtArgs, grid, 256, lds, stream, a);
  } else if (synthesis) {
    se = PPSCI_SET_MAX_LDS(sht_synthesis_kernel<false>, lds);
    if (se == 0) PPSCI_LAUNCH(sht_synthesis_kernel<false>, ShtArgs, grid, 256, lds, stream, a);
  } else {
    se = PPSCI_SET_MAX_LDS(sht_analysis_kernel, lds);
    if (se == 0) PPSCI_LAUNCH(sht_analysis_kernel, ShtArgs, grid, 256, lds, stream, a);
  }
  if (se != 0 || PPSCI_LAST_LAUNCH_ERROR() != 0) {
    ppsci_set_error("sht: launch failed");
    return PPSCI_E_LAUNCH;
  }
  return PPSCI_OK;
}

extern "C" int ppsci_sht_analysis(int n, int H, int W, int L, int M, const float* tw, const float* leg, const float* x, float* X,
                                  void* stream) {
  return sht_run(0, n, H, W, L, M, tw, leg, x, X, stream);
}
extern "C" int ppsci_sht_synthesis(int n, int H, int W, int L, int M, const float* tw, const float* leg, const float* Z, float* y,
                                   void* stream) {
  return sht_run(1, n, H, W, L, M, tw, leg, Z, y, stream);
}

// ppsci_sht_contract + ppsci_sht_synthesis in one launch: the workgroup that synthesises plane (b, c) contracts that plane's
// coefficients itself (the same sum in the same order: bit-identical to the two launches)
extern "C" int ppsci_sht_synthesis_contract(int B, int Ci, int Co, int conj_t, int H, int W, int L, int M, const float* tw,
                                            const float* leg, const float* x, const float* w_re, const float* w_im, float* y,
                                            void* stream) {
  if (B < 1 || Ci < 1 || Co < 1 || !x || !w_re || !w_im) {
    ppsci_set_error("sht_synthesis_contract: invalid argument");
    return PPSCI_E_INVALID;
  }
  return sht_run(1, B * (conj_t ? Ci : Co), H, W, L, M, tw, leg, nullptr, y, stream, x, w_re, w_im, Ci, Co, conj_t ? 1 : 0);
}

// ------------------------------------------------------------------------------------------ contraction with weights per degree
struct ShtConArgs {
  const float* x;   // [B, Cs, L, M, 2]   the summed side
  const float* wr;  // [Ci, Co, L]
  const float* wi;
  float* out;       // [B, Cp, L, M, 2]
  int B, Ci, Co, L, M, conj_t;
  long long total;
};

// conj_t = 0: out[b, o] = sum_i x[b, i] w[i, o, l];   1: out[b, i] = sum_o x[b, o] conj(w[i, o, l])   (the data gradient)
__global__ void __launch_bounds__(256) sht_contract_kernel(ShtConArgs a) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= a.total) return;
  const int lm = a.L * a.M;
  const int pos = (int)(t % lm), l = pos / a.M;
  const long long bc = t / lm;
  const int Cp = a.conj_t ? a.Ci : a.Co, Cs = a.conj_t ? a.Co : a.Ci;
  const int c = (int)(bc % Cp), b = (int)(bc / Cp);
  float sr = 0.f, si = 0.f;
#pragma unroll 8
  for (int s = 0; s < Cs; ++s) {
    const float* xp = a.x + (((long long)b * Cs + s) * lm + pos) * 2;
    const long long wi_ = (a.conj_t ? (long long)c * a.Co + s : (long long)s * a.Co + c) * a.L + l;
    const float xr = xp[0], xi = xp[1], wr = a.wr[wi_], wim = a.conj_t ? -a.wi[wi_] : a.wi[wi_];
    sr += xr * wr - xi * wim;
    si += xr * wim + xi * wr;
  }
  a.out[2 * t] = sr;
  a.out[2 * t + 1] = si;
}

struct ShtWArgs {
  const float* x;  // [B, Ci, L, M, 2]
  const float* g;  // [B, Co, L, M, 2]
  float* gwr;      // [Ci, Co, L]
  float* gwi;
  int B, Ci, Co, L, M;
  long long total;
};

// gw[i, o, l] = sum_b sum_m conj(x[b, i, l, m]) g[b, o, l, m]
__global__ void __launch_bounds__(256) sht_wgrad_kernel(ShtWArgs a) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= a.total) return;
  const int l = (int)(t % a.L);
  const long long io = t / a.L;
  const int o = (int)(io % a.Co), i = (int)(io / a.Co);
  const int lm = a.L * a.M;
  float sr = 0.f, si = 0.f;
  for (int b = 0; b < a.B; ++b) {
    const float* xp = a.x + (((long long)b * a.Ci + i) * lm + (long long)l * a.M) * 2;
    const float* gp = a.g + (((long long)b * a.Co + o) * lm + (long long)l * a.M) * 2;
#pragma unroll 8
    for (int m = 0; m < a.M; ++m) {
      const float xr = xp[2 * m], xi = xp[2 * m + 1], gr = gp[2 * m], gi = gp[2 * m + 1];
      sr += xr * gr + xi * gi;
      si += xr * gi - xi * gr;
    }
  }
  a.gwr[t] = sr;
  a.gwi[t] = si;
}

extern "C" int ppsci_sht_contract(int B, int Ci, int Co, int L, int M, const float* x, const float* w_re, const float* w_im, int conj_t,
                                  float* out, void* stream) {
  if (B < 1 || Ci < 1 || Co < 1 || L < 1 || M < 1 || !x || !w_re || !w_im || !out) {
    ppsci_set_error("sht_contract: invalid argument");
    return PPSCI_E_INVALID;
  }
  ShtConArgs a{x, w_re, w_im, out, B, Ci, Co, L, M, conj_t ? 1 : 0, (long long)B * (conj_t ? Ci : Co) * L * M};
  PPSCI_LAUNCH(sht_contract_kernel, ShtConArgs, (int)((a.total + 255) / 256), 256, 0, stream, a);
  if (PPSCI_LAST_LAUNCH_ERROR() != 0) {
    ppsci_set_error("sht_contract: launch failed");
    return PPSCI_E_LAUNCH;
  }
  return PPSCI_OK;
}

extern "C" int ppsci_sht_contract_wgrad(int B, int Ci, int Co, int L, int M, const float* x, const float* g, float* gw_re, float* gw_im,
                                        void* stream) {
  if (B < 1 || Ci < 1 || Co < 1 || L < 1 || M < 1 || !x || !g || !gw_re || !gw_im) {
    ppsci_set_error("sht_contract_wgrad: invalid argument");
    return PPSCI_E_INVALID;
  }
  ShtWArgs a{x, g, gw_re, gw_im, B, Ci, Co, L, M, (long long)Ci * Co * L};
  PPSCI_LAUNCH(sht_wgrad_kernel, ShtWArgs, (int)((a.total + 255) / 256), 256, 0, stream, a);
  if (PPSCI_LAST_LAUNCH_ERROR() != 0) {
    ppsci_set_error("sht_contract_wgrad: launch failed");
    return PPSCI_E_LAUNCH;
  }
  return PPSCI_OK;
}
