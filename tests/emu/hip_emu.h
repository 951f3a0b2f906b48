// hip_emu.h -- TEST INFRASTRUCTURE ONLY.  A tiny CPU SIMT emulator so that the *same* kernel
// source that hipcc compiles for gfx950 (paddlescience_amd/csrc/*.hip) can be compiled for the
// host (clang++ -DPPSCI_EMU) and executed lane-by-lane in this GPU-less container.  Every lane of
// a workgroup is a fiber (a private stack + a hand-written x86-64 context switch; glibc's swapcontext pays a signal-mask
// system call per switch, which was a fifth of the suite's time); __syncthreads / wave collectives (MFMA, shuffles) are fiber
// barriers.  The product never loads the emulator build: paddlescience_amd/_lib.py only accepts
// a library whose ppsci_is_device_build() returns 1 unless a test injects one explicitly.
//
// Emulated gfx950 semantics (from /opt/skills/guides/cdna_hip_programming.md section 3):
//   v_mfma_f32_16x16x4_f32 : lane l supplies A[i=l&15][k=l>>4] and B[k=l>>4][j=l&15]; D/C
//   register r of lane l is element (row = 4*(l>>4) + r, col = l&15); the result is a k-ordered
//   fmaf chain (bitwise the same as the hardware, per the guide).
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <map>
#include <string>
#include <unistd.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __restrict__
#define __shared__ static  // blocks run one after the other: one instance is the block's LDS variable

struct emu_dim3 {
  unsigned x = 1, y = 1, z = 1;
};

namespace emu {

constexpr int kWave = 64;
constexpr size_t kStack = 1024 * 1024;  // (the one-launch kernels inline forward + program + reverse: ~300 KB of -O0 frame)

struct Barrier {
  int n = 0, count = 0, gen = 0;
};

struct Wave {
  Barrier bar;
  float xa[2][kWave];
  float xb[2][kWave];
  const void* xp[2][kWave];  // addresses (LDS transpose reads)
  unsigned xw[2][kWave][8];  // wide operands (bf16 MFMAs): 4 dwords of A, 4 of B per lane
  unsigned op = 0;  // collective counter (same in every lane of the wave)
};

// ---- context switch: callee-saved registers on the fiber's own stack, the stack pointer in Ctx
struct Ctx {
  void* sp = nullptr;
};
#if !defined(__x86_64__)
#error "tests/emu/hip_emu.h: the fiber switch is written for x86-64 (System V ABI)"
#endif
__attribute__((naked, noinline)) static void emu_switch(Ctx* /*from: rdi*/, Ctx* /*to: rsi*/) {
  __asm__ volatile(
      "pushq %rbp\n\tpushq %rbx\n\tpushq %r12\n\tpushq %r13\n\tpushq %r14\n\tpushq %r15\n\t"
      "movq %rsp, (%rdi)\n\t"
      "movq (%rsi), %rsp\n\t"
      "popq %r15\n\tpopq %r14\n\tpopq %r13\n\tpopq %r12\n\tpopq %rbx\n\tpopq %rbp\n\t"
      "ret\n\t");
}

struct Fiber {
  Ctx ctx;
  char* stack = nullptr;
  bool done = false;
  unsigned tid = 0;
  unsigned wave_op = 0;
  const Barrier* wait_bar = nullptr;  // blocked on this barrier until its generation moves on (the scheduler skips the fiber)
  int wait_gen = 0;
};

struct State {
  Ctx sched;
  std::vector<Fiber> fibers;
  std::vector<Wave> waves;
  Barrier block_bar;
  int cur = -1;
  emu_dim3 blockIdx, blockDim, gridDim;
  std::vector<float> smem;
  void (*entry)(void*) = nullptr;
  void* args = nullptr;
};

inline State& st() {
  static State s;
  return s;
}

inline void yield() {
  State& s = st();
  emu_switch(&s.fibers[s.cur].ctx, &s.sched);
}

inline void barrier_wait(Barrier& b) {
  int g = b.gen;
  if (++b.count == b.n) {
    b.count = 0;
    b.gen++;
  } else {
    Fiber& f = st().fibers[st().cur];
    f.wait_bar = &b;
    f.wait_gen = g;
    while (b.gen == g) yield();
    f.wait_bar = nullptr;
  }
}

static void trampoline() {
  State& s = st();
  s.entry(s.args);
  s.fibers[s.cur].done = true;
  emu_switch(&s.fibers[s.cur].ctx, &s.sched);
  __builtin_trap();  // a finished fiber is never resumed
}

// A fresh fiber: its stack holds what emu_switch pops -- six callee-saved registers, then `trampoline` as the return address, with
// the stack pointer 8 below a 16-byte boundary at trampoline's entry, as the ABI has it behind a call.
inline void fiber_reset(Fiber& f) {
  uintptr_t top = ((uintptr_t)f.stack + kStack) & ~(uintptr_t)15;
  void** sp = (void**)top;
  *--sp = nullptr;                 // the "return address" of trampoline (it never returns)
  *--sp = (void*)&trampoline;      // popped by emu_switch's ret
  for (int k = 0; k < 6; ++k) *--sp = nullptr;
  f.ctx.sp = sp;
}

// Runs `entry(args)` for every thread of every block of the grid (blocks sequentially).
inline void launch(emu_dim3 grid, emu_dim3 block, size_t dyn_lds_bytes, void (*entry)(void*), void* args) {
  State& s = st();
  unsigned nthreads = block.x;
  if (nthreads % kWave != 0) {
    fprintf(stderr, "emu: block size must be a multiple of 64\n");
    abort();
  }
  s.entry = entry;
  s.args = args;
  s.blockDim = block;
  s.gridDim = grid;
  s.smem.assign(dyn_lds_bytes / sizeof(float) + 16, 0.f);
  if (s.fibers.size() < nthreads) {
    size_t old = s.fibers.size();
    s.fibers.resize(nthreads);
    for (size_t i = old; i < nthreads; ++i) s.fibers[i].stack = (char*)malloc(kStack);
  }
  s.waves.assign(nthreads / kWave, Wave());
  for (unsigned b = 0; b < grid.x; ++b) {
    s.blockIdx.x = b;
    // poison LDS between blocks: real LDS is not zero-initialised
    for (auto& v : s.smem) v = std::nanf("");
    s.block_bar = Barrier();
    s.block_bar.n = (int)nthreads;
    for (auto& w : s.waves) {
      w = Wave();
      w.bar.n = kWave;
    }
    for (unsigned t = 0; t < nthreads; ++t) {
      Fiber& f = s.fibers[t];
      f.done = false;
      f.tid = t;
      f.wave_op = 0;
      f.wait_bar = nullptr;
      fiber_reset(f);
    }
    unsigned remaining = nthreads;
    while (remaining) {
      for (unsigned t = 0; t < nthreads; ++t) {
        Fiber& f = s.fibers[t];
        if (f.done || (f.wait_bar != nullptr && f.wait_bar->gen == f.wait_gen)) continue;  // finished, or still blocked
        s.cur = (int)t;
        emu_switch(&s.sched, &f.ctx);
        if (f.done) --remaining;
      }
    }
  }
  s.cur = -1;
}

// PPSCI_EMU_PROFILE=<path prefix>: wall time per kernel name, written at exit to <prefix>.<pid> (which sources are worth an
// optimised emulator build, tests/emu/build_emu.py HOT).
struct Profile {
  std::map<std::string, std::pair<double, long>> t;
  const char* path = getenv("PPSCI_EMU_PROFILE");
  ~Profile() {
    if (!path || t.empty()) return;
    std::string fn = std::string(path) + "." + std::to_string((long)getpid());
    if (FILE* f = fopen(fn.c_str(), "w")) {
      for (auto& kv : t) fprintf(f, "%.6f %ld %s\n", kv.second.first, kv.second.second, kv.first.c_str());
      fclose(f);
    }
  }
};
inline Profile& profile() {
  static Profile p;
  return p;
}
inline void launch_named(const char* name, emu_dim3 grid, emu_dim3 block, size_t dyn_lds_bytes, void (*entry)(void*), void* args) {
  Profile& p = profile();
  if (!p.path) return launch(grid, block, dyn_lds_bytes, entry, args);
  auto t0 = std::chrono::steady_clock::now();
  launch(grid, block, dyn_lds_bytes, entry, args);
  auto& e = p.t[name];
  e.first += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  e.second += 1;
}

struct TidProxy {
  struct X {
    operator unsigned() const { return st().fibers[st().cur].tid; }
  } x;
};
struct BidProxy {
  struct X {
    operator unsigned() const { return st().blockIdx.x; }
  } x;
};
struct BdimProxy {
  struct X {
    operator unsigned() const { return st().blockDim.x; }
  } x;
};
struct GdimProxy {
  struct X {
    operator unsigned() const { return st().gridDim.x; }
  } x;
};

inline Wave& my_wave() {
  State& s = st();
  return s.waves[s.fibers[s.cur].tid / kWave];
}
inline unsigned my_lane() { return st().fibers[st().cur].tid % kWave; }

// Exchange one float per lane through the wave; returns pointer to the published buffer.
inline const float* wave_publish(float v, float** other = nullptr, float v2 = 0.f) {
  Wave& w = my_wave();
  Fiber& f = st().fibers[st().cur];
  unsigned buf = f.wave_op & 1u;
  f.wave_op++;
  w.xa[buf][my_lane()] = v;
  w.xb[buf][my_lane()] = v2;
  barrier_wait(w.bar);
  if (other) *other = w.xb[buf];
  return w.xa[buf];
}

}  // namespace emu

static emu::TidProxy threadIdx;
static emu::BidProxy blockIdx;
static emu::BdimProxy blockDim;
static emu::GdimProxy gridDim;

#define PPSCI_DYN_SMEM(name) float* name = emu::st().smem.data()

inline void __syncthreads() { emu::barrier_wait(emu::st().block_bar); }
inline void ppsci_wave_sync() { emu::barrier_wait(emu::my_wave().bar); }

// device intrinsics the product sources use unconditionally (the emulator executes one lane at a time):
inline float __expf(float x) { return expf(x); }                    // v_exp_f32 path
inline float __builtin_amdgcn_rcpf(float x) { return 1.f / x; }     // v_rcp_f32
inline int __builtin_amdgcn_readfirstlane(int x) { return x; }      // the value is wave-uniform by construction

inline float __shfl_xor(float v, int mask, int width = 64) {
  const float* buf = emu::wave_publish(v);
  unsigned l = emu::my_lane();
  unsigned src = l ^ (unsigned)mask;
  // width semantics: stay inside the aligned `width` segment (masks used here never leave it)
  if ((src / width) != (l / width)) src = l;
  return buf[src];
}

inline float __shfl(float v, int src_lane, int width = 64) {
  const float* buf = emu::wave_publish(v);
  unsigned l = emu::my_lane();
  unsigned src = (l / width) * width + ((unsigned)src_lane % width);
  return buf[src];
}

inline f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, f32x4 c, int, int, int) {
  float* bb = nullptr;
  const float* aa = emu::wave_publish(a, &bb, b);
  unsigned l = emu::my_lane();
  unsigned g = l >> 4, col = l & 15;
  f32x4 d = c;
  for (int r = 0; r < 4; ++r) {
    unsigned row = 4 * g + r;
    float acc = c[r];
    for (int k = 0; k < 4; ++k) acc = std::fmaf(aa[16 * k + row], bb[16 * k + col], acc);
    d[r] = acc;
  }
  return d;
}

// ---- bf16 (XDL) MFMAs: operands are packed bf16 pairs; products are exact in fp32, accumulated in k order ----------
inline unsigned ppsci_cvt_pk_bf16(float a, float b) {  // v_cvt_pk_bf16_f32: round to nearest even, low half = a
  auto rne = [](float x) -> unsigned {
    unsigned u;
    std::memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;  // NaN stays NaN
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
  };
  return (rne(a) & 0xffffu) | (rne(b) << 16);
}
// v_dot2c_f32_bf16 with the constant operands (-1, 0) / (0, -1): x minus the low / high bf16 of h (exact in fp32)
inline float ppsci_bf16_sub_lo(unsigned h, float x) {
  const unsigned v = h << 16;
  float f;
  std::memcpy(&f, &v, 4);
  return x - f;
}
inline float ppsci_bf16_sub_hi(unsigned h, float x) {
  const unsigned v = h & 0xffff0000u;
  float f;
  std::memcpy(&f, &v, 4);
  return x - f;
}
inline float emu_bf16_at(const unsigned* w, int j) {  // j-th bf16 of a packed array
  const unsigned v = (j & 1) ? (w[j >> 1] & 0xffff0000u) : (w[j >> 1] << 16);
  float f;
  std::memcpy(&f, &v, 4);
  return f;
}
inline f32x4 emu_xdl(const unsigned* a, const unsigned* b, int ndw, f32x4 c) {  // ndw dwords per operand and lane
  emu::Wave& w = emu::my_wave();
  emu::Fiber& f = emu::st().fibers[emu::st().cur];
  const unsigned buf = f.wave_op & 1u;
  f.wave_op++;
  const unsigned l = emu::my_lane();
  for (int i = 0; i < ndw; ++i) w.xw[buf][l][i] = a[i], w.xw[buf][l][4 + i] = b[i];
  emu::barrier_wait(w.bar);
  const unsigned g = l >> 4, col = l & 15;
  const int per = 2 * ndw;  // bf16 per lane: k = per * group + j
  f32x4 d = c;
  for (int r = 0; r < 4; ++r) {
    const unsigned row = 4 * g + r;
    float acc = c[r];
    for (int kg = 0; kg < 4; ++kg)
      for (int j = 0; j < per; ++j)
        acc = std::fmaf(emu_bf16_at(w.xw[buf][16 * kg + row], j), emu_bf16_at(&w.xw[buf][16 * kg + col][4], j), acc);
    d[r] = acc;
  }
  return d;
}
inline f32x4 ppsci_xdl32(u32x2 a_lo, u32x2 a_hi, u32x2 b_lo, u32x2 b_hi, f32x4 c) {
  const unsigned a[4] = {a_lo[0], a_lo[1], a_hi[0], a_hi[1]}, b[4] = {b_lo[0], b_lo[1], b_hi[0], b_hi[1]};
  return emu_xdl(a, b, 4, c);
}
inline f32x4 ppsci_xdl32a(u32x4 a, u32x2 b_lo, u32x2 b_hi, f32x4 c) {
  const unsigned aa[4] = {a[0], a[1], a[2], a[3]}, b[4] = {b_lo[0], b_lo[1], b_hi[0], b_hi[1]};
  return emu_xdl(aa, b, 4, c);
}
inline f32x4 ppsci_xdl32aa(u32x4 a, u32x4 b, f32x4 c) {
  const unsigned aa[4] = {a[0], a[1], a[2], a[3]}, bb[4] = {b[0], b[1], b[2], b[3]};
  return emu_xdl(aa, bb, 4, c);
}
// ds_read_b64_tr_b16: element j of lane i (of a 16-lane group) = 16-bit element (i & 3) at the address of lane 4j + (i >> 2)
inline u32x2 ppsci_lds_read_tr16(const void* p) {
  emu::Wave& w = emu::my_wave();
  emu::Fiber& f = emu::st().fibers[emu::st().cur];
  const unsigned buf = f.wave_op & 1u;
  f.wave_op++;
  const unsigned l = emu::my_lane();
  w.xp[buf][l] = p;
  emu::barrier_wait(w.bar);
  const unsigned grp = l & ~15u, i = l & 15u;
  unsigned short e[4];
  for (unsigned j = 0; j < 4; ++j) e[j] = ((const unsigned short*)w.xp[buf][grp + 4 * j + (i >> 2)])[i & 3];
  emu::barrier_wait(w.bar);  // nobody may overwrite the source before every lane has gathered
  return (u32x2){(unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16)};
}
inline f32x4 ppsci_xdl16(u32x2 a, u32x2 b, f32x4 c) {
  const unsigned aa[2] = {a[0], a[1]}, bb[2] = {b[0], b[1]};
  return emu_xdl(aa, bb, 2, c);
}

// DPP row_shr:n (ctrl 0x110+n) with bound_ctrl: lane i of each 16-lane row reads lane i-n, 0 if outside.
inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  float f;
  std::memcpy(&f, &src, 4);
  const float* buf = emu::wave_publish(f);
  unsigned l = emu::my_lane();
  if (ctrl > 0x110 && ctrl <= 0x11f && row_mask == 0xf && bank_mask == 0xf) {
    int n = ctrl - 0x110;
    int i = (int)(l & 15) - n;
    if (i < 0) return bound_ctrl ? 0 : old;
    float r = buf[(l & ~15u) + (unsigned)i];
    int ri;
    std::memcpy(&ri, &r, 4);
    return ri;
  }
  fprintf(stderr, "emu: unsupported DPP ctrl 0x%x\n", ctrl);
  abort();
}

inline float atomicAdd(float* p, float v) {
  float old = *p;
  *p = old + v;
  return old;
}
inline unsigned atomicAdd(unsigned* p, unsigned v) {
  unsigned old = *p;
  *p = old + v;
  return old;
}
inline void __threadfence() {}  // blocks run one after the other: every earlier block's writes are visible

typedef void* hipStream_t;
