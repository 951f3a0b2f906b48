// taylor_api.hip -- C ABI entry points of the Taylor-mode forward / reverse kernels: argument
// checks, derived sizes, and dispatch to the per-activation translation units.
#include "taylor_tile.h"
#include "taylor_step.h"
#include "epi_static.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

// ---- pre-split hidden-weight fragments for the feature-split XDL kernels (taylor_fwd_wx.inc / taylor_bwd_wx.inc) ----
// One thread per (layer, row block, k-pair, lane): the two 4-value halves of the K = 32 A operand, split into three bf16
// planes.  forward (bwd = 0): rb = output block, lane (g, c) holds W[in = 16kb + 4g + r][out = 16rb + c];
// backward (bwd = 1): rb = input block, lane (g, c) holds W[in = 16rb + c][out = 16kb + 4g + r]  (taylor_tile.h).
__global__ void __launch_bounds__(256) ppsci_presplit_kernel(const float* params, ppsci_derived q, int H, int L, int bwd,
                                                             u32x4* out) {
  const int NB = q.NB, NKP = NB / 2;
  const int per_layer = NB * NKP * 64;
  const int total = (L - 1) * per_layer;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int l = 1 + idx / per_layer, rem = idx % per_layer;
    const int lane = rem & 63, pair = rem >> 6, rb = pair / NKP, kp = pair - rb * NKP;
    const int g = lane >> 4, c = lane & 15;
    const float* W = params + q.offW[l];
    ppsci_split4 sp[2];
#pragma unroll
    for (int hlf = 0; hlf < 2; ++hlf) {
      const int kb = 2 * kp + hlf;
      f32x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int in = bwd ? 16 * rb + c : 16 * kb + 4 * g + r, o = bwd ? 16 * kb + 4 * g + r : 16 * rb + c;
        v[r] = (in < H && o < H) ? W[in * H + o] : 0.f;
      }
      sp[hlf] = ppsci_split(v);
    }
    u32x4* dst = out + (long long)(l - 1) * (NB * NKP * 3 * 64) + (long long)(pair * 3) * 64 + lane;
#pragma unroll
    for (int p = 0; p < 3; ++p) dst[p * 64] = (u32x4){sp[0].p[p][0], sp[0].p[p][1], sp[1].p[p][0], sp[1].p[p][1]};
  }
}

// Both directions in ONE launch into a caller-owned buffer (the fused tile kernel's step workspace): forward fragments
// first, backward fragments PPSCI_GFRAG_PER_LAYER * (L - 1) u32x4 behind them.
__global__ void __launch_bounds__(256) ppsci_presplit2_kernel(const float* params, ppsci_derived q, int H, int L, u32x4* out) {
  const int NB = q.NB, NKP = NB / 2;
  const int per_layer = NB * NKP * 64;
  const int total = (L - 1) * per_layer;
  for (int idx2 = blockIdx.x * blockDim.x + threadIdx.x; idx2 < 2 * total; idx2 += gridDim.x * blockDim.x) {
    const int bwd = idx2 >= total ? 1 : 0, idx = idx2 - bwd * total;
    const int l = 1 + idx / per_layer, rem = idx % per_layer;
    const int lane = rem & 63, pair = rem >> 6, rb = pair / NKP, kp = pair - rb * NKP;
    const int g = lane >> 4, c = lane & 15;
    const float* W = params + q.offW[l];
    ppsci_split4 sp[2];
#pragma unroll
    for (int hlf = 0; hlf < 2; ++hlf) {
      const int kb = 2 * kp + hlf;
      f32x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int in = bwd ? 16 * rb + c : 16 * kb + 4 * g + r, o = bwd ? 16 * kb + 4 * g + r : 16 * rb + c;
        v[r] = (in < H && o < H) ? W[in * H + o] : 0.f;
      }
      sp[hlf] = ppsci_split(v);
    }
    u32x4* dst = out + (long long)bwd * (L - 1) * (NB * NKP * 3 * 64) + (long long)(l - 1) * (NB * NKP * 3 * 64) +
                 (long long)(pair * 3) * 64 + lane;
#pragma unroll
    for (int p = 0; p < 3; ++p) dst[p * 64] = (u32x4){sp[0].p[p][0], sp[0].p[p][1], sp[1].p[p][0], sp[1].p[p][1]};
  }
}

// Fragment cache of the separate (non-fused) feature-split kernels: one device buffer per (parameter buffer, direction),
// allocated on first use (an eager call: the engine captures HIP graphs only from the second step on) and re-filled by
// every launch that needs it -- the parameters change every step.  (The fused tile kernel keeps its fragments in the
// caller-owned step workspace instead.)  A buffer lives until its parameter buffer is released
// (ppsci_release_fragments, called by the engine that owns the parameters) or until the table is full, in which case
// the oldest entry is freed -- hipFree waits for the device, so no launch that reads it is still in flight.
struct FragEntry {
  const void* key;
  int bwd;
  size_t bytes;
  void* dev;
};
#define PPSCI_FRAG_SLOTS 64
static FragEntry g_frag[PPSCI_FRAG_SLOTS];
static int g_nfrag = 0, g_frag_next = 0;
static std::mutex g_frag_mutex;
static void frag_free(void* dev) {
#ifdef PPSCI_EMU
  free(dev);
#else
  (void)hipFree(dev);
#endif
}
static void* frag_cache_get(const void* key, int bwd, size_t bytes) {
  std::lock_guard<std::mutex> lock(g_frag_mutex);
  for (int i = 0; i < g_nfrag; ++i)
    if (g_frag[i].key == key && g_frag[i].bwd == bwd && g_frag[i].bytes >= bytes) return g_frag[i].dev;
  void* dev = nullptr;
#ifdef PPSCI_EMU
  dev = malloc(bytes);
#else
  if (hipMalloc(&dev, bytes) != hipSuccess) dev = nullptr;
#endif
  if (!dev) return nullptr;
  int slot = g_nfrag;
  if (g_nfrag < PPSCI_FRAG_SLOTS) {
    ++g_nfrag;
  } else {  // full: the oldest entry goes
    slot = g_frag_next;
    g_frag_next = (g_frag_next + 1) % PPSCI_FRAG_SLOTS;
    frag_free(g_frag[slot].dev);
  }
  g_frag[slot] = FragEntry{key, bwd, bytes, dev};
  return dev;
}
extern "C" void ppsci_release_fragments(const float* params) {
  std::lock_guard<std::mutex> lock(g_frag_mutex);
  int n = 0;
  for (int i = 0; i < g_nfrag; ++i) {
    if (g_frag[i].key == (const void*)params) frag_free(g_frag[i].dev);
    else g_frag[n++] = g_frag[i];
  }
  g_nfrag = n;
  g_frag_next = 0;
}

const void* ppsci_presplit(const float* params, const ppsci_mlp_desc& d, const ppsci_derived& q, int bwd, void* stream) {
  const int L = d.n_hidden;
  const long long n4 = (long long)(L - 1) * q.NB * (q.NB / 2) * 3 * 64;
  void* dev = frag_cache_get(params, bwd, (size_t)n4 * 16);
  if (!dev) {
    ppsci_set_error("presplit: cannot allocate %lld B of fragment cache", n4 * 16);
    return nullptr;
  }
  const int total = (L - 1) * q.NB * (q.NB / 2) * 64;
  int grid = (total + 255) / 256;
  if (grid > 2048) grid = 2048;
  struct PArgs {
    const float* params;
    ppsci_derived q;
    int H, L, bwd;
    u32x4* out;
  };
#ifdef PPSCI_EMU
  PArgs pa{params, q, d.width, L, bwd, (u32x4*)dev};
  emu::launch(emu_dim3{(unsigned)grid}, emu_dim3{256u}, 0,
              [](void* p) { PArgs& a = *(PArgs*)p; ppsci_presplit_kernel(a.params, a.q, a.H, a.L, a.bwd, a.out); }, &pa);
#else
  hipLaunchKernelGGL(ppsci_presplit_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, params, q, d.width, L, bwd,
                     (u32x4*)dev);
  if (hipGetLastError() != hipSuccess) {
    ppsci_set_error("presplit: launch failed");
    return nullptr;
  }
#endif
  return dev;
}

static int fill_fwd(FwdArgs& a, const ppsci_mlp_desc* d, int64_t n_points) {
  memset(&a, 0, sizeof(a));
  if (!d || ppsci_derive(d, &a.q) != PPSCI_OK) return PPSCI_E_INVALID;
  a.d = *d;
  a.N = n_points;
  a.ntiles = (int)((n_points + PPSCI_TILE - 1) / PPSCI_TILE);
  return PPSCI_OK;
}

static int fill_bwd(BwdArgs& a, const ppsci_mlp_desc* d, int64_t n_points) {
  memset(&a, 0, sizeof(a));
  if (!d || ppsci_derive(d, &a.q) != PPSCI_OK) return PPSCI_E_INVALID;
  a.d = *d;
  a.N = n_points;
  a.ntiles = (int)((n_points + PPSCI_TILE - 1) / PPSCI_TILE);
  return PPSCI_OK;
}

static int run_fwd_act(FwdArgs& a, void* stream, int launch, int* grid) {
  if (a.d.fourier_half > 0) {
    if (a.d.activation != PPSCI_ACT_TANH || 2 * a.d.fourier_half != a.d.width || a.d.n_hidden < 2) {
      ppsci_set_error("fourier embedding: needs a tanh net whose width equals the embedding dimension");
      return PPSCI_E_UNSUPPORTED;
    }
    return ppsci_fwd_run_tanh_fourier(a, stream, launch, grid);
  }
  switch (a.d.activation) {
    case PPSCI_ACT_TANH: return ppsci_fwd_run_tanh(a, stream, launch, grid);
    case PPSCI_ACT_SILU: return ppsci_fwd_run_silu(a, stream, launch, grid);
    case PPSCI_ACT_SIN: return ppsci_fwd_run_sin(a, stream, launch, grid);
    case PPSCI_ACT_SIGMOID: return ppsci_fwd_run_sigmoid(a, stream, launch, grid);
    case PPSCI_ACT_COS: return ppsci_fwd_run_cos(a, stream, launch, grid);
    case PPSCI_ACT_GELU: return ppsci_fwd_run_gelu(a, stream, launch, grid);
    case PPSCI_ACT_SWISH: return ppsci_fwd_run_swish(a, stream, launch, grid);
    case PPSCI_ACT_STAN: return ppsci_fwd_run_stan(a, stream, launch, grid);
    case PPSCI_ACT_RELU: return ppsci_fwd_run_relu(a, stream, launch, grid);
    case PPSCI_ACT_LEAKY_RELU: return ppsci_fwd_run_leaky_relu(a, stream, launch, grid);
    case PPSCI_ACT_ELU: return ppsci_fwd_run_elu(a, stream, launch, grid);
    case PPSCI_ACT_SELU: return ppsci_fwd_run_selu(a, stream, launch, grid);
    case PPSCI_ACT_IDENTITY: return ppsci_fwd_run_identity(a, stream, launch, grid);
    default: ppsci_set_error("unknown activation %d", a.d.activation); return PPSCI_E_UNSUPPORTED;
  }
}

static int run_bwd_act(BwdArgs& a, void* stream, int launch, int* grid) {
  if (a.d.fourier_half > 0) {
    if (a.d.activation != PPSCI_ACT_TANH || 2 * a.d.fourier_half != a.d.width || a.d.n_hidden < 2) {
      ppsci_set_error("fourier embedding: needs a tanh net whose width equals the embedding dimension");
      return PPSCI_E_UNSUPPORTED;
    }
    return ppsci_bwd_run_tanh_fourier(a, stream, launch, grid);
  }
  switch (a.d.activation) {
    case PPSCI_ACT_TANH: return ppsci_bwd_run_tanh(a, stream, launch, grid);
    case PPSCI_ACT_SILU: return ppsci_bwd_run_silu(a, stream, launch, grid);
    case PPSCI_ACT_SIN: return ppsci_bwd_run_sin(a, stream, launch, grid);
    case PPSCI_ACT_SIGMOID: return ppsci_bwd_run_sigmoid(a, stream, launch, grid);
    case PPSCI_ACT_COS: return ppsci_bwd_run_cos(a, stream, launch, grid);
    case PPSCI_ACT_GELU: return ppsci_bwd_run_gelu(a, stream, launch, grid);
    case PPSCI_ACT_SWISH: return ppsci_bwd_run_swish(a, stream, launch, grid);
    case PPSCI_ACT_STAN: return ppsci_bwd_run_stan(a, stream, launch, grid);
    case PPSCI_ACT_RELU: return ppsci_bwd_run_relu(a, stream, launch, grid);
    case PPSCI_ACT_LEAKY_RELU: return ppsci_bwd_run_leaky_relu(a, stream, launch, grid);
    case PPSCI_ACT_ELU: return ppsci_bwd_run_elu(a, stream, launch, grid);
    case PPSCI_ACT_SELU: return ppsci_bwd_run_selu(a, stream, launch, grid);
    case PPSCI_ACT_IDENTITY: return ppsci_bwd_run_identity(a, stream, launch, grid);
    default: ppsci_set_error("unknown activation %d", a.d.activation); return PPSCI_E_UNSUPPORTED;
  }
}

extern "C" int64_t ppsci_bwd_partial_rows(const ppsci_mlp_desc* d, int64_t n_points) {
  BwdArgs a;
  if (n_points <= 0 || fill_bwd(a, d, n_points) != PPSCI_OK) return 0;
  int grid = 0;
  if (run_bwd_act(a, nullptr, 0, &grid) != PPSCI_OK) return 0;
  // the per-workgroup partial sums are reduced inside ppsci_taylor_bwd (workspace): one finished row
  (void)grid;
  return 1;
}

static long long bwd_per_tile_floats(const BwdArgs& a) { return (long long)(a.d.n_hidden - 1) * a.q.HP * a.q.HP; }

// layer-by-layer kernel of padded width 256 (taylor_bwd_lw.inc): the adjoint handed from launch to launch, [ntiles][S][NB][64] float4
static long long bwd_hbuf_floats(const BwdArgs& a) {
  const int S = 1 + a.d.n1 + a.d.n2 + a.d.n3 + a.d.n4;
  return a.lw ? (long long)a.ntiles * S * a.q.NB * 64 * 4 : 0;
}

// slots of per-tile (or, accumulating kernels, per-workgroup) hidden-weight gradient blocks in the workspace
static long long bwd_wpart_slots(const BwdArgs& a, int grid) { return a.accum ? grid : (long long)a.ntiles + 1; }

extern "C" int64_t ppsci_bwd_workspace_bytes(const ppsci_mlp_desc* d, int64_t n_points) {
  BwdArgs a;
  if (n_points <= 0 || fill_bwd(a, d, n_points) != PPSCI_OK) return 0;
  int grid = 0;
  if (run_bwd_act(a, nullptr, 0, &grid) != PPSCI_OK) return 0;
  // Sized for BOTH layouts of the single-wave kernels (one slot per workgroup when they accumulate, one per tile + a spare
  // when they stream): which one runs depends on the process-global ppsci_set_bwd_accum flag at LAUNCH time, and a buffer
  // sized under one setting must not be overrun under the other.  The feature-split XDL kernels always accumulate.
  long long slots = bwd_wpart_slots(a, grid);
  if (!a.xdl_split && (long long)a.ntiles + 1 > slots) slots = (long long)a.ntiles + 1;
  const long long chunks = slots < PPSCI_WRED_CHUNKS ? slots : PPSCI_WRED_CHUNKS;
  // hidden-weight blocks per tile (+ the spare slot) or per workgroup | chunk sums | per-workgroup small-parameter
  // rows | their chunk sums
  const long long fl = (slots + chunks) * bwd_per_tile_floats(a) +
                       ((long long)grid + PPSCI_WRED_CHUNKS) * ppsci_small_params(a.d, a.q) + bwd_hbuf_floats(a);
  return fl * 4 + 16 + (a.lw ? 16 : 0);
}

extern "C" int ppsci_taylor_fwd(const ppsci_mlp_desc* d, const float* params, int64_t n_points,
                                const float* const* inputs_host, float* U, void* stash, void* stream) {
  FwdArgs a;
  if (!params || !inputs_host || !U || n_points < 0 || fill_fwd(a, d, n_points) != PPSCI_OK) {
    ppsci_set_error("taylor_fwd: invalid argument");
    return PPSCI_E_INVALID;
  }
  if (n_points == 0) return PPSCI_OK;
  a.params = params;
  for (int j = 0; j < d->d_raw; ++j) a.x[j] = inputs_host[j];
  a.U = U;
  a.stash = (f32x4*)stash;
  return run_fwd_act(a, stream, 1, nullptr);
}

static int taylor_bwd_impl(const ppsci_mlp_desc* d, const float* params, int64_t n_points,
                           const float* const* inputs_host, const float* Ubar, const void* stash,
                           void* workspace, float* grad_partials, void* stream) {
  BwdArgs a;
  if (!params || !inputs_host || !Ubar || !stash || !workspace || !grad_partials || n_points <= 0 ||
      fill_bwd(a, d, n_points) != PPSCI_OK) {
    ppsci_set_error("taylor_bwd: invalid argument");
    return PPSCI_E_INVALID;
  }
  a.params = params;
  for (int j = 0; j < d->d_raw; ++j) a.x[j] = inputs_host[j];
  a.Ubar = Ubar;
  a.stash = (const f32x4*)stash;
  int grid = 0;
  if (run_bwd_act(a, nullptr, 0, &grid) != PPSCI_OK) return PPSCI_E_UNSUPPORTED;
  const long long slots = bwd_wpart_slots(a, grid);
  const long long chunks = slots < PPSCI_WRED_CHUNKS ? slots : PPSCI_WRED_CHUNKS;
  const int psmall = ppsci_small_params(a.d, a.q);
  float* wpart = (float*)workspace;
  float* tmp = wpart + slots * bwd_per_tile_floats(a);
  float* small_rows = tmp + chunks * bwd_per_tile_floats(a);
  float* small_tmp = small_rows + (long long)grid * psmall;  // [PPSCI_WRED_CHUNKS][psmall]
  if (a.lw) {  // (16-byte aligned: float4 accesses)
    float* hb = small_tmp + (long long)PPSCI_WRED_CHUNKS * psmall;
    a.hbuf = (f32x4*)(((uintptr_t)hb + 15) & ~(uintptr_t)15);
  }
  a.partials = small_rows;
  a.wpart = (f32x4*)workspace;
  int rc = run_bwd_act(a, stream, 1, &grid);
  if (rc != PPSCI_OK) return rc;
  // fixed-order two-stage sum over the tiles' (or the workgroups') hidden-weight blocks and over the workgroups' compact
  // rows of W0 / biases / W_last, written in the canonical parameter layout
  return ppsci_wgrad_reduce(a.d, a.q, a.accum ? (int)slots : a.ntiles, wpart, tmp, small_rows, grid, small_tmp, grad_partials,
                            stream);
}

extern "C" int ppsci_taylor_bwd(const ppsci_mlp_desc* d, const float* params, int64_t n_points,
                                const float* const* inputs_host, const float* Ubar, const void* stash,
                                void* workspace, float* grad_partials, void* stream) {
  return taylor_bwd_impl(d, params, n_points, inputs_host, Ubar, stash, workspace, grad_partials, stream);
}

// The checked form: the kernel choice of padded width 129..256 (ppsci_set_bwd_layerwise) is read when the workspace is SIZED
// and again at LAUNCH; a workspace sized under one setting and run under the other would be overrun (the layer-by-layer
// kernel keeps its hand-over buffer there, the fp32-MFMA kernel its per-tile gradient blocks).  With the size known the
// launch runs the kernel the buffer was sized for, or refuses.
extern "C" int ppsci_taylor_bwd_ws(const ppsci_mlp_desc* d, const float* params, int64_t n_points,
                                   const float* const* inputs_host, const float* Ubar, const void* stash,
                                   void* workspace, int64_t workspace_bytes, float* grad_partials, void* stream) {
  if (workspace_bytes < 0 || n_points <= 0 || !d)
    return taylor_bwd_impl(d, params, n_points, inputs_host, Ubar, stash, workspace, grad_partials, stream);
  const int knob = ppsci_get_bwd_layerwise();
  int64_t need = ppsci_bwd_workspace_bytes(d, n_points);
  if (need > 0 && need <= workspace_bytes)
    return taylor_bwd_impl(d, params, n_points, inputs_host, Ubar, stash, workspace, grad_partials, stream);
  ppsci_set_bwd_layerwise(!knob);
  const int64_t other = ppsci_bwd_workspace_bytes(d, n_points);
  int rc;
  if (other > 0 && other <= workspace_bytes) {
    rc = taylor_bwd_impl(d, params, n_points, inputs_host, Ubar, stash, workspace, grad_partials, stream);
  } else {
    ppsci_set_error("taylor_bwd: workspace of %lld bytes, %lld needed (ppsci_bwd_workspace_bytes)", (long long)workspace_bytes,
                    (long long)need);
    rc = PPSCI_E_INVALID;
  }
  ppsci_set_bwd_layerwise(knob);
  return rc;
}

// ------------------------------------------------------------------------------------ one-launch step
// process-global knobs of the one-launch step (tests / tools): the fused tile kernel on or off; how a fused launch ends
// (-1: by grid size, 0: the in-kernel reduction tree, 1: the host issues the reduction kernels behind the launch)
static int g_fused_step = 1, g_step_tail = -1, g_fast_vm = 1, g_static_prog = 1;
extern "C" void ppsci_set_fast_program(int on) { g_fast_vm = on ? 1 : 0; }
extern "C" void ppsci_set_static_program(int on) { g_static_prog = on ? 1 : 0; }
extern "C" int ppsci_epilogue_predecode(const ppsci_epilogue_desc* e, uint32_t* out256, int* n_loads) {
  int nl = 0;
  if (!e || !out256) return -1;
  const int n = epi_fast_encode(*e, out256, &nl);
  if (n_loads) *n_loads = nl;
  return n;
}
extern "C" void ppsci_set_fused_step(int on) { g_fused_step = on ? 1 : 0; }
extern "C" void ppsci_set_step_tail(int mode) { g_step_tail = (mode < 0 || mode > 3) ? -1 : mode; }
// a tree over more rows than this is slower than the reduction kernels: every level is a ~10-20 us pass of ONE workgroup
// over 16 rows of (L-1) * 16 KB, against ~12 us for the two kernels over all rows
#define PPSCI_FUSED_TREE_MAX_GRID 48

static int run_step_act(StepArgs& a, void* stream, int launch, int* grid) {
  if (a.f.d.fourier_half > 0) return PPSCI_E_UNSUPPORTED;
  if (launch == 2 ? a.t.fused != 0 : g_fused_step != 0) {
    int rc = PPSCI_E_UNSUPPORTED;
    const bool st = a.e.static_id > 0;  // the residual program is a compile-time table: the kernels without the VM
    switch (a.f.d.activation) {
      case PPSCI_ACT_TANH: rc = st ? ppsci_fused_static_run_tanh(a, stream, launch, grid) : ppsci_fused_run_tanh(a, stream, launch, grid); break;
      case PPSCI_ACT_SILU: rc = st ? ppsci_fused_static_run_silu(a, stream, launch, grid) : ppsci_fused_run_silu(a, stream, launch, grid); break;
      case PPSCI_ACT_SIN: rc = st ? ppsci_fused_static_run_sin(a, stream, launch, grid) : ppsci_fused_run_sin(a, stream, launch, grid); break;
      default: break;
    }
    if (rc != PPSCI_E_UNSUPPORTED || launch == 2) return rc;
    a.e.static_id = 0;  // (no fused tile kernel for this net: the one-launch kernel of padded width 32 runs the VM)
  }
  a.t.fused = 0;
  switch (a.f.d.activation) {
    case PPSCI_ACT_TANH: return ppsci_step_run_tanh(a, stream, launch, grid);
    case PPSCI_ACT_SILU: return ppsci_step_run_silu(a, stream, launch, grid);
    case PPSCI_ACT_SIN: return ppsci_step_run_sin(a, stream, launch, grid);
    default: ppsci_set_error("taylor_step: no one-launch kernel for activation %d", a.f.d.activation); return PPSCI_E_UNSUPPORTED;
  }
}

// workspace layout (floats): rows_w [grid][per_tile] | rows_s [grid][psmall] | rows_l [grid][n_res] | tree rows | counters
// fused tile kernel: ... | pre-split fragments (both directions) | chunk sums of the reduction kernels | a spare gradient row
struct StepLayout {
  long long rows_w, rows_s, rows_l, tree, counters, frag, red_tmp, red_small, red_row, fastprog, total;
  int per_tile, psmall, rowlen, tree_rows;
};
static StepLayout step_layout(const StepArgs& a, int grid, int n_res) {
  StepLayout y;
  memset(&y, 0, sizeof(y));
  y.per_tile = (int)bwd_per_tile_floats(a.b);
  y.psmall = ppsci_small_params(a.b.d, a.b.q);
  y.rowlen = y.per_tile + ((y.psmall + 3) & ~3) + ((n_res + 3) & ~3);
  y.tree_rows = 0;
  for (int n = grid; n > 1;) {
    n = (n + PPSCI_STEP_FAN - 1) / PPSCI_STEP_FAN;
    y.tree_rows += n;
  }
  const auto pad4 = [](long long v) { return (v + 3) & ~3LL; };
  y.rows_w = 0;
  y.rows_s = y.rows_w + pad4((long long)grid * y.per_tile);
  y.rows_l = y.rows_s + pad4((long long)grid * y.psmall);
  y.tree = y.rows_l + pad4((long long)grid * (n_res > 0 ? n_res : 1));
  y.counters = y.tree + (long long)y.tree_rows * y.rowlen;
  y.total = y.counters + pad4(y.tree_rows + 1);
  if (a.t.fused) {
    const long long chunks = grid < PPSCI_WRED_CHUNKS ? grid : PPSCI_WRED_CHUNKS;
    y.frag = y.total;
    y.red_tmp = y.frag + 2LL * (a.f.d.n_hidden - 1) * PPSCI_GFRAG_PER_LAYER(a.f.q.NB) * 4;
    y.red_small = y.red_tmp + chunks * y.per_tile;
    y.red_row = y.red_small + pad4(PPSCI_WRED_CHUNKS * (long long)y.psmall);
    y.fastprog = y.red_row + pad4(a.f.q.P);  // pre-decoded residual program (epi_fast_encode)
    y.total = y.fastprog + EPI_FAST_WORDS;
  } else {
    y.fastprog = y.total;
    y.total += EPI_FAST_WORDS;
  }
  return y;
}

static int fill_step(StepArgs& a, const ppsci_mlp_desc* d, const ppsci_epilogue_desc* e, int64_t n_points) {
  memset(&a, 0, sizeof(a));
  if (!d || !e || n_points <= 0 || fill_fwd(a.f, d, n_points) != PPSCI_OK || fill_bwd(a.b, d, n_points) != PPSCI_OK) return PPSCI_E_INVALID;
  if (e->n_instr < 1 || e->n_instr > PPSCI_MAX_PROG || e->n_res < 0 || e->n_res > PPSCI_MAX_RES) return PPSCI_E_INVALID;
  for (int i = 0; i < e->n_instr; ++i)
    if (e->prog[i].op == PPSCI_OP_LD_PARAM) {
      ppsci_set_error("taylor_step: learnable equation parameters take the separate launches");
      return PPSCI_E_UNSUPPORTED;
    }
  for (int k = 0; k < e->n_res; ++k)
    if (e->res[k].kind == PPSCI_LOSS_LINEAR || e->res[k].scale_param != 0) {
      ppsci_set_error("taylor_step: batch sums (PPSCI_LOSS_LINEAR / scale_param) take the separate launches");
      return PPSCI_E_UNSUPPORTED;
    }
  a.e.e = *e;
  epi_fill_loads(a.e);
  a.e.N = n_points;
  a.e.ntiles = a.f.ntiles;
  // a program that IS one of the compile-time tables (epi_static.h) selects the fused tile kernels built without the VM:
  // known before the launch is planned (grid, LDS and occupancy are those of the kernel that will run)
  // (those kernels are also the ones written for RAW inputs -- no period embedding, no caller-supplied input streams: the
  // layer-0 code of the general case, with its descriptor look-ups per tile, is not in them)
  bool raw = true;
  for (int j = 0; j < d->d_raw; ++j) raw = raw && d->embed[j] == PPSCI_EMBED_NONE;
  if (g_fast_vm && g_static_prog && raw) {
    unsigned fast[EPI_FAST_WORDS];
    int nl = 0;
    const int nfast = epi_fast_encode(a.e.e, fast, &nl);
    if (nfast >= 0) a.e.static_id = epi_static_match(a.e.e, fast, nfast, nl, d->n1, d->n2, d->d_out);
  }
  return PPSCI_OK;
}

extern "C" int64_t ppsci_taylor_step_workspace_bytes(const ppsci_mlp_desc* d, const ppsci_epilogue_desc* e, int64_t n_points) {
  StepArgs a;
  int grid = 0;
  if (fill_step(a, d, e, n_points) != PPSCI_OK || run_step_act(a, nullptr, 0, &grid) != PPSCI_OK) return 0;
  return step_layout(a, grid, e->n_res).total * 4;
}

struct ppsci_step_plan {
  StepArgs a;
  StepLayout y;
  float* ws;
  bool frag_fresh;  // fused tile kernel: the fragments in the workspace are those of the parameters as this plan's last run left them
};

extern "C" int ppsci_taylor_step_kind(const ppsci_mlp_desc* d, const ppsci_epilogue_desc* e, int64_t n_points) {
  StepArgs a;
  int grid = 0;
  if (fill_step(a, d, e, n_points) != PPSCI_OK || run_step_act(a, nullptr, 0, &grid) != PPSCI_OK) return 0;
  return a.t.fused ? 2 : 1;
}

extern "C" ppsci_step_plan* ppsci_taylor_step_plan(const ppsci_mlp_desc* d, const ppsci_epilogue_desc* e, float* params,
                                                   int64_t n_points, const float* const* inputs_host,
                                                   const float* const* aux_host, float* U, float* Ubar, float* residual_out,
                                                   void* stash, void* workspace, int64_t workspace_bytes, float* loss_terms,
                                                   float* grad) {
  if (!params || !inputs_host || !workspace || !loss_terms || !grad) {
    ppsci_set_error("taylor_step: invalid argument");
    return nullptr;
  }
  ppsci_step_plan* plan = new ppsci_step_plan;
  StepArgs& a = plan->a;
  const auto fail = [&]() -> ppsci_step_plan* {
    delete plan;
    return nullptr;
  };
  int rc = fill_step(a, d, e, n_points);
  if (rc != PPSCI_OK) {
    if (rc == PPSCI_E_INVALID) ppsci_set_error("taylor_step: invalid argument");
    return fail();
  }
  // the program is checked exactly as ppsci_epilogue does (indices into inputs / streams / aux / earlier values)
  for (int i = 0; i < e->n_instr; ++i) {
    const ppsci_instr& ins = e->prog[i];
    bool ok = ins.op >= 0 && ins.op < PPSCI_OP_COUNT;
    if (ok) {
      if (ins.op == PPSCI_OP_LD_IN) ok = ins.a >= 0 && ins.a < e->n_in && ins.a < d->d_raw;
      else if (ins.op == PPSCI_OP_LD_U) ok = ins.a >= 0 && ins.a < e->n_streams;
      else if (ins.op == PPSCI_OP_LD_AUX) ok = ins.a >= 0 && ins.a < e->n_aux && aux_host;
      else if (ins.op != PPSCI_OP_CONST) {
        ok = ins.a >= 0 && ins.a < i;
        const bool binary = ins.op == PPSCI_OP_ADD || ins.op == PPSCI_OP_SUB || ins.op == PPSCI_OP_MUL ||
                            ins.op == PPSCI_OP_DIV || ins.op == PPSCI_OP_POW || ins.op == PPSCI_OP_MAX ||
                            ins.op == PPSCI_OP_MIN || ins.op == PPSCI_OP_ATAN2;
        if (binary) ok = ok && ins.b >= 0 && ins.b < i;
      }
    }
    if (!ok) {
      ppsci_set_error("taylor_step: bad instruction %d (op %d a %d b %d)", i, ins.op, ins.a, ins.b);
      return fail();
    }
  }
  for (int k = 0; k < e->n_res; ++k) {
    const ppsci_residual& r = e->res[k];
    if (r.value < 0 || r.value >= e->n_instr || r.label >= e->n_aux || r.weight >= e->n_aux || r.area >= e->n_aux ||
        r.kind < PPSCI_LOSS_MSE || r.kind > PPSCI_LOSS_ABSREL || ((r.label >= 0 || r.weight >= 0 || r.area >= 0) && !aux_host)) {
      ppsci_set_error("taylor_step: bad residual %d", k);
      return fail();
    }
  }
  if (e->n_streams != d->d_out * (1 + d->n1 + d->n2 + d->n3 + d->n4) || e->n_in > PPSCI_MAX_IN || e->n_aux > PPSCI_MAX_AUX) {
    ppsci_set_error("taylor_step: the program's stream count does not match the network's");
    return fail();
  }
  int grid = 0;
  if (run_step_act(a, nullptr, 0, &grid) != PPSCI_OK) return fail();
  const StepLayout y = step_layout(a, grid, e->n_res);
  // the grid (hence the layout) depends on process-global knobs (ppsci_set_max_grid, the occupancy query): a workspace
  // sized under other settings must not be overrun
  if (y.total * 4 > workspace_bytes) {
    ppsci_set_error("taylor_step: the workspace holds %lld B, the planned launch (%d workgroups) needs %lld B", (long long)workspace_bytes,
                    grid, y.total * 4);
    return fail();
  }
  if (!a.t.fused && (!U || !Ubar || !stash)) {  // (the fused tile kernel keeps them on the chip; there they are optional outputs)
    ppsci_set_error("taylor_step: invalid argument");
    return fail();
  }
  float* ws = (float*)workspace;
  plan->y = y;
  plan->ws = ws;
  plan->frag_fresh = false;
  if (a.t.fused) {
    a.f.xfrag = ws + y.frag;
    a.b.xfrag = (const u32x4*)(ws + y.frag) + (long long)(d->n_hidden - 1) * PPSCI_GFRAG_PER_LAYER(a.f.q.NB);
    // 0: the whole reduction tree inside the launch (small grids); 1: nothing inside, two kernels behind it (large grids);
    // 2: the tree's first level inside the launch, one kernel behind it -- measured SLOWER than 1 at 512 workgroups
    // (100 k points: 302 us against 270 us per step: the rows then have to be agent-scope write-through stores and the
    // last workgroups of the groups add a serial pass over 16 x 49 KB each), kept as a tested knob
    // 3 (the default for large grids): as 1 with ONE kernel behind the launch, which also leaves the bf16 fragments of the
    // updated hidden matrices behind (no weight-split launch in front of the next step)
    a.t.external = g_step_tail >= 0 ? g_step_tail : (grid > PPSCI_FUSED_TREE_MAX_GRID ? 3 : 0);
    if (a.t.external == 2 && grid <= PPSCI_STEP_FAN) a.t.external = 0;  // (one group: its sum IS the total)
    a.t.one_tail = a.t.external == 3 ? 1 : 0;
    if (a.t.one_tail) a.t.external = 1;  // (what the kernel sees: the launch stops at the workgroups' rows)
  }
  {
    unsigned fast[EPI_FAST_WORDS];
    int nfast_loads = 0;
    const int nfast = g_fast_vm ? epi_fast_encode(a.e.e, fast, &nfast_loads) : -1;
    a.e.fast = nullptr;
    a.e.nfast = a.e.nfast_loads = 0;
    if (nfast >= 0) {  // (a synchronous copy of < 400 bytes; plans are made outside graph captures)
#ifdef PPSCI_EMU
      memcpy(ws + y.fastprog, fast, sizeof(fast));
#else
      if (hipMemcpy(ws + y.fastprog, fast, sizeof(fast), hipMemcpyHostToDevice) != hipSuccess) {
        ppsci_set_error("taylor_step: cannot upload the pre-decoded residual program");
        return fail();
      }
#endif
      a.e.fast = (const unsigned*)(ws + y.fastprog);
      a.e.nfast = nfast;
      a.e.nfast_loads = nfast_loads;
    }
  }
  a.f.params = a.b.params = params;
  for (int j = 0; j < d->d_raw; ++j) a.f.x[j] = a.b.x[j] = a.e.x[j] = inputs_host[j];
  for (int j = 0; j < e->n_aux; ++j) a.e.aux[j] = aux_host[j];
  a.f.U = U;
  a.f.stash = (f32x4*)stash;
  a.e.U = U;
  a.e.Ubar = Ubar;
  a.e.resid = residual_out;
  a.e.partials = ws + y.rows_l;
  a.b.Ubar = Ubar;
  a.b.stash = (const f32x4*)stash;
  a.b.partials = ws + y.rows_s;
  a.b.wpart = (f32x4*)(ws + y.rows_w);
  StepTail& t = a.t;
  t.rows_w = ws + y.rows_w;
  t.rows_s = ws + y.rows_s;
  t.rows_l = ws + y.rows_l;
  t.tree = ws + y.tree;
  t.counters = (unsigned*)(ws + y.counters);
  t.grad = grad;
  t.loss_terms = loss_terms;
  t.p = params;
  t.per_tile = y.per_tile;
  t.psmall = y.psmall;
  t.n_res = e->n_res;
  t.rowlen = y.rowlen;
  return plan;
}

extern "C" void ppsci_taylor_step_plan_free(ppsci_step_plan* plan) { delete plan; }

// 0: the plan's residual program runs on the epilogue VM; > 0: it is compile-time table `id` of csrc/epi_static_programs.h
// (`name`, when given, receives the table's name)
extern "C" int ppsci_taylor_step_plan_static(const ppsci_step_plan* plan, const char** name) {
  const int id = plan ? plan->a.e.static_id : 0;
  if (name) *name = epi_static_name(id);
  return id;
}

extern "C" int ppsci_taylor_step_plan_set_scales(ppsci_step_plan* plan, const ppsci_epilogue_desc* e) {
  if (!plan || !e || e->n_res != plan->a.e.e.n_res) {
    ppsci_set_error("taylor_step_plan_set_scales: invalid argument");
    return PPSCI_E_INVALID;
  }
  for (int k = 0; k < e->n_res; ++k) plan->a.e.e.res[k].scale = e->res[k].scale;
  return PPSCI_OK;
}

extern "C" int ppsci_taylor_step_run(ppsci_step_plan* plan, int accumulate, const ppsci_adam_args* adam, void* stream) {
  return ppsci_taylor_step_run_ex(plan, accumulate, adam, stream, 0);
}

extern "C" int ppsci_taylor_step_run_ex(ppsci_step_plan* plan, int accumulate, const ppsci_adam_args* adam, void* stream, int flags) {
  if (!plan) {
    ppsci_set_error("taylor_step_run: invalid argument");
    return PPSCI_E_INVALID;
  }
  StepTail& t = plan->a.t;
  t.accumulate = accumulate ? 1 : 0;
  t.do_adam = 0;
  if (adam) {
    if (!adam->m || !adam->v || adam->step_t < 1) {
      ppsci_set_error("taylor_step: invalid Adam arguments");
      return PPSCI_E_INVALID;
    }
    const double b1t = pow((double)adam->beta1, (double)adam->step_t), b2t = pow((double)adam->beta2, (double)adam->step_t);
    const double c2 = sqrt(1.0 - b2t);  // == ppsci_adam_step
    t.do_adam = 1;
    t.m = adam->m;
    t.v = adam->v;
    t.lr_t = (float)(adam->lr * c2 / (1.0 - b1t));
    t.beta1 = adam->beta1;
    t.beta2 = adam->beta2;
    t.eps_t = (float)(adam->eps * c2);
    t.grad_scale = adam->grad_scale;
  }
  int grid = 0;
  if (!plan->a.t.fused) return run_step_act(plan->a, stream, 2, &grid);
  // ---- fused tile kernel: split the (new) hidden-to-hidden matrices, run the tiles, and -- for large grids -- sum the
  // workgroups' rows with the reduction kernels
  StepArgs& a = plan->a;
  const StepLayout& y = plan->y;
  float* ws = plan->ws;
  // PPSCI_STEP_KEEP_FRAGMENTS: the caller vouches that nothing has written the parameters since this plan's last run; if
  // that run left current fragments behind (its tail kernel writes them next to the Adam update) the weight split is skipped
  const bool keep = (flags & PPSCI_STEP_KEEP_FRAGMENTS) && plan->frag_fresh;
  plan->frag_fresh = false;
  if (!keep) {
    const int L = a.f.d.n_hidden, NB = a.f.q.NB;
    const int total = 2 * (L - 1) * NB * (NB / 2) * 64;
    struct PArgs {
      const float* params;
      ppsci_derived q;
      int H, L;
      u32x4* out;
    };
#ifdef PPSCI_EMU
    PArgs pa{a.f.params, a.f.q, a.f.d.width, L, (u32x4*)(ws + y.frag)};
    emu::launch(emu_dim3{(unsigned)((total + 255) / 256)}, emu_dim3{256u}, 0,
                [](void* p) { PArgs& x = *(PArgs*)p; ppsci_presplit2_kernel(x.params, x.q, x.H, x.L, x.out); }, &pa);
#else
    hipLaunchKernelGGL(ppsci_presplit2_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, a.f.params, a.f.q,
                       a.f.d.width, L, (u32x4*)(ws + y.frag));
    if (hipGetLastError() != hipSuccess) {
      ppsci_set_error("taylor_step: presplit launch failed");
      return PPSCI_E_LAUNCH;
    }
#endif
  }
  const int do_adam = t.do_adam;
  if (t.external) t.do_adam = 0;
  int rc = run_step_act(a, stream, 2, &grid);
  t.do_adam = do_adam;
  if (rc != PPSCI_OK || !t.external) return rc;  // (Adam inside the launch: the fragments are stale from here on)
  // two launches: chunk sums of the workgroups' rows, then -- one thread per parameter -- the total, grad (+)= it, the Adam
  // update; the loss terms by one more workgroup of the second launch
  ppsci_wred_extras x;
  memset(&x, 0, sizeof(x));
  x.accumulate = accumulate ? 1 : 0;
  x.loss_rows = t.rows_l;
  x.loss_out = t.loss_terms;
  x.loss_nrows = t.grid;
  x.n_res = t.n_res;
  if (adam) {
    x.p = t.p;
    x.m = t.m;
    x.v = t.v;
    x.lr_t = t.lr_t;
    x.beta1 = t.beta1;
    x.beta2 = t.beta2;
    x.eps_t = t.eps_t;
    x.grad_scale = t.grad_scale;
  }
  if (t.one_tail) {
    rc = ppsci_wgrad_tail(a.b.d, a.b.q, t.grid, t.rows_w, t.rows_s, t.grad, x, ws + y.frag, stream);
    plan->frag_fresh = rc == PPSCI_OK;  // (with or without Adam: the fragments are those of the parameters as they are now)
    return rc;
  }
  if (t.external == 2) {
    // the launch has left the sums of groups of PPSCI_STEP_FAN rows in the first rows of `tree`
    const int nrows = (t.grid + PPSCI_STEP_FAN - 1) / PPSCI_STEP_FAN;
    const long long off_s = t.per_tile, off_l = t.per_tile + ((t.psmall + 3) & ~3);
    x.loss_rows = nullptr;  // (taken from the rows)
    return ppsci_wgrad_reduce_chunks(a.b.d, a.b.q, nrows, t.tree, t.rowlen, off_s, off_l, t.grad, x, stream);
  }
  rc = ppsci_wgrad_reduce_ex(a.b.d, a.b.q, t.grid, t.rows_w, ws + y.red_tmp, t.rows_s, t.grid, ws + y.red_small, t.grad, x, stream);
  plan->frag_fresh = rc == PPSCI_OK && !adam;  // (no Adam here: the parameters are what the fragments were split from)
  return rc;
}

// Data parallelism (one SUM all-reduce of the flat gradient between the step's sums and the optimizer,
// /root/reference/ppsci/solver/train.py:168-175): the Adam update from the FINISHED gradient `grad` of the plan -- and, for
// the fused tile kernel, the bf16 fragments of the updated hidden matrices -- in one launch, so that a data-parallel step is
// tile kernel + tail kernel (sums) + all-reduce + this, with no weight-split launch in front of the next step.
extern "C" int ppsci_taylor_step_plan_apply(ppsci_step_plan* plan, const ppsci_adam_args* adam, void* stream) {
  if (!plan || !adam || !adam->m || !adam->v || adam->step_t < 1) {
    ppsci_set_error("taylor_step_plan_apply: invalid argument");
    return PPSCI_E_INVALID;
  }
  StepArgs& a = plan->a;
  StepTail& t = a.t;
  const double b1t = pow((double)adam->beta1, (double)adam->step_t), b2t = pow((double)adam->beta2, (double)adam->step_t);
  const double c2 = sqrt(1.0 - b2t);  // == ppsci_adam_step
  ppsci_wred_extras x;
  memset(&x, 0, sizeof(x));
  x.accumulate = 1;  // no rows: the "sum" is the gradient as it stands
  x.p = t.p;
  x.m = adam->m;
  x.v = adam->v;
  x.lr_t = (float)(adam->lr * c2 / (1.0 - b1t));
  x.beta1 = adam->beta1;
  x.beta2 = adam->beta2;
  x.eps_t = (float)(adam->eps * c2);
  x.grad_scale = adam->grad_scale;
  plan->frag_fresh = false;
  const int rc = ppsci_wgrad_tail(a.b.d, a.b.q, 0, t.rows_w, t.rows_s, t.grad, x, t.fused ? plan->ws + plan->y.frag : nullptr, stream);
  plan->frag_fresh = rc == PPSCI_OK && t.fused;
  return rc;
}

// measurement: the main kernel of the planned step alone (kind 2: the fused tile kernel without the weight split in front
// of it and without any reduction -- the workgroups' rows stay in the workspace; kind 1: the whole launch)
extern "C" int ppsci_taylor_step_run_main(ppsci_step_plan* plan, void* stream) {
  if (!plan) {
    ppsci_set_error("taylor_step_run_main: invalid argument");
    return PPSCI_E_INVALID;
  }
  StepTail& t = plan->a.t;
  const int ext = t.external, adam = t.do_adam;
  if (t.fused) t.external = 1;
  t.do_adam = 0;
  int grid = 0;
  const int rc = run_step_act(plan->a, stream, 2, &grid);
  t.external = ext;
  t.do_adam = adam;
  return rc;
}

extern "C" int ppsci_taylor_step(const ppsci_mlp_desc* d, const ppsci_epilogue_desc* e, float* params, int64_t n_points,
                                 const float* const* inputs_host, const float* const* aux_host, float* U, float* Ubar,
                                 float* residual_out, void* stash, void* workspace, int64_t workspace_bytes,
                                 float* loss_terms, float* grad, int accumulate, const ppsci_adam_args* adam, void* stream) {
  ppsci_step_plan* plan = ppsci_taylor_step_plan(d, e, params, n_points, inputs_host, aux_host, U, Ubar, residual_out, stash,
                                                 workspace, workspace_bytes, loss_terms, grad);
  if (!plan) {
    StepArgs a;
    const int rc = fill_step(a, d, e, n_points);
    int grid = 0;
    return rc != PPSCI_OK ? rc : (run_step_act(a, nullptr, 0, &grid) != PPSCI_OK ? PPSCI_E_UNSUPPORTED : PPSCI_E_INVALID);
  }
  const int rc = ppsci_taylor_step_run(plan, accumulate, adam, stream);
  ppsci_taylor_step_plan_free(plan);
  return rc;
}
