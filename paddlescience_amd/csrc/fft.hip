// fft.hip -- batched 2-D real FFTs of the FNO spectral convolution straight on hipFFT (rocFFT), on the caller's
// stream: rfftn / irfftn of /root/reference/ppsci/arch/fno_block.py:718-720, :791 without a tensor library in between
// (no input clones, no separate normalisation kernels: hipFFT does not scale, and the 1/(H*W) of any `fft_norm` pair is
// folded into the spectral contraction, see ppsci_spectral_conv2d_fwd_scaled).
//
//   ppsci_fft2d_r2c : [batch, H, W] real            -> [batch, H, W/2+1] complex (interleaved floats), unscaled
//   ppsci_fft2d_c2r : [batch, H, W/2+1] complex     -> [batch, H, W] real, unscaled; the INPUT is destroyed (hipFFT C2R)
//
// Plans are created on first use per (batch, H, W, direction) and cached for the life of the process; creating a plan
// allocates its work buffer, executing it does not (so executions can be captured into a HIP graph after a warm-up).
// The CPU SIMT emulator build (tests only) has no hipFFT: it evaluates the same transforms by a separable O(n^2) DFT in
// double precision.
#include "ppsci_common.h"

#include <math.h>
#include <string.h>

extern "C" void ppsci_set_error(const char* fmt, ...);

#ifdef PPSCI_EMU
#include <complex>
#include <vector>

static void dft_rows(std::vector<std::complex<double>>& a, int rows, int n, int sign) {
  std::vector<std::complex<double>> tmp(n);
  for (int r = 0; r < rows; ++r) {
    for (int k = 0; k < n; ++k) {
      std::complex<double> s = 0;
      for (int j = 0; j < n; ++j) s += a[(size_t)r * n + j] * std::polar(1.0, sign * 2.0 * M_PI * (double)((long long)k * j % n) / n);
      tmp[k] = s;
    }
    for (int k = 0; k < n; ++k) a[(size_t)r * n + k] = tmp[k];
  }
}

static void dft2(std::vector<std::complex<double>>& img, int H, int W, int sign) {  // full complex 2-D DFT, in place
  dft_rows(img, H, W, sign);
  std::vector<std::complex<double>> t((size_t)H * W);
  for (int h = 0; h < H; ++h)
    for (int w = 0; w < W; ++w) t[(size_t)w * H + h] = img[(size_t)h * W + w];
  dft_rows(t, W, H, sign);
  for (int h = 0; h < H; ++h)
    for (int w = 0; w < W; ++w) img[(size_t)h * W + w] = t[(size_t)w * H + h];
}

extern "C" int ppsci_fft2d_r2c(int batch, int H, int W, const float* in, float* out, void* stream) {
  const int Wf = W / 2 + 1;
  std::vector<std::complex<double>> img((size_t)H * W);
  for (int b = 0; b < batch; ++b) {
    for (size_t i = 0; i < (size_t)H * W; ++i) img[i] = in[(size_t)b * H * W + i];
    dft2(img, H, W, -1);
    for (int h = 0; h < H; ++h)
      for (int k = 0; k < Wf; ++k) {
        out[(((size_t)b * H + h) * Wf + k) * 2 + 0] = (float)img[(size_t)h * W + k].real();
        out[(((size_t)b * H + h) * Wf + k) * 2 + 1] = (float)img[(size_t)h * W + k].imag();
      }
  }
  return PPSCI_OK;
}

extern "C" int ppsci_fft2d_c2r(int batch, int H, int W, float* in, float* out, void* stream) {
  // the complex-to-real transform of hipFFT / pocketfft: inverse DFT along H for every kept column, then along W with
  // the Hermitian extension of the half spectrum; imaginary parts of the DC / Nyquist column results are dropped
  const int Wf = W / 2 + 1;
  std::vector<std::complex<double>> col((size_t)Wf * H);
  for (int b = 0; b < batch; ++b) {
    for (int k = 0; k < Wf; ++k)
      for (int h = 0; h < H; ++h)
        col[(size_t)k * H + h] = std::complex<double>(in[(((size_t)b * H + h) * Wf + k) * 2], in[(((size_t)b * H + h) * Wf + k) * 2 + 1]);
    dft_rows(col, Wf, H, +1);
    for (int h = 0; h < H; ++h)
      for (int w = 0; w < W; ++w) {
        double s = col[(size_t)0 * H + h].real();
        for (int k = 1; k < Wf; ++k) {
          const std::complex<double> z = col[(size_t)k * H + h] * std::polar(1.0, 2.0 * M_PI * (double)((long long)k * w % W) / W);
          s += ((W % 2 == 0 && k == W / 2) ? 1.0 : 2.0) * z.real();
        }
        out[((size_t)b * H + h) * W + w] = (float)s;
      }
  }
  return PPSCI_OK;
}

#else
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>

#include <map>
#include <mutex>
#include <tuple>

static std::mutex g_mu;
static std::map<std::tuple<int, int, int, int, int>, hipfftHandle> g_plans;  // (device, batch, H, W, type)

static int get_plan(int batch, int H, int W, hipfftType type, hipfftHandle* out) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(g_mu);
  auto key = std::make_tuple(dev, batch, H, W, (int)type);
  auto it = g_plans.find(key);
  if (it != g_plans.end()) {
    *out = it->second;
    return PPSCI_OK;
  }
  hipfftHandle plan;
  int n[2] = {H, W};
  const int Wf = W / 2 + 1;
  const int rdist = H * W, cdist = H * Wf;
  hipfftResult r = hipfftPlanMany(&plan, 2, n, nullptr, 1, type == HIPFFT_R2C ? rdist : cdist, nullptr, 1,
                                  type == HIPFFT_R2C ? cdist : rdist, type, batch);
  if (r != HIPFFT_SUCCESS) {
    ppsci_set_error("fft2d: hipfftPlanMany failed (%d) for batch %d, %d x %d", (int)r, batch, H, W);
    return PPSCI_E_LAUNCH;
  }
  g_plans[key] = plan;
  *out = plan;
  return PPSCI_OK;
}

extern "C" int ppsci_fft2d_r2c(int batch, int H, int W, const float* in, float* out, void* stream) {
  if (batch < 1 || H < 1 || W < 2 || !in || !out) {
    ppsci_set_error("fft2d_r2c: invalid argument");
    return PPSCI_E_INVALID;
  }
  hipfftHandle plan;
  int rc = get_plan(batch, H, W, HIPFFT_R2C, &plan);
  if (rc != PPSCI_OK) return rc;
  if (hipfftSetStream(plan, (hipStream_t)stream) != HIPFFT_SUCCESS ||
      hipfftExecR2C(plan, (hipfftReal*)in, (hipfftComplex*)out) != HIPFFT_SUCCESS) {
    ppsci_set_error("fft2d_r2c: hipfftExecR2C failed");
    return PPSCI_E_LAUNCH;
  }
  return PPSCI_OK;
}

extern "C" int ppsci_fft2d_c2r(int batch, int H, int W, float* in, float* out, void* stream) {
  if (batch < 1 || H < 1 || W < 2 || !in || !out) {
    ppsci_set_error("fft2d_c2r: invalid argument");
    return PPSCI_E_INVALID;
  }
  hipfftHandle plan;
  int rc = get_plan(batch, H, W, HIPFFT_C2R, &plan);
  if (rc != PPSCI_OK) return rc;
  if (hipfftSetStream(plan, (hipStream_t)stream) != HIPFFT_SUCCESS ||
      hipfftExecC2R(plan, (hipfftComplex*)in, (hipfftReal*)out) != HIPFFT_SUCCESS) {
    ppsci_set_error("fft2d_c2r: hipfftExecC2R failed");
    return PPSCI_E_LAUNCH;
  }
  return PPSCI_OK;
}
#endif
