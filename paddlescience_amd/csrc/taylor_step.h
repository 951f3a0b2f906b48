// taylor_step.h -- argument block of the one-launch step kernel (taylor_step.inc) and its per-activation entry points.
#pragma once
#include "epilogue_vm.h"

#ifdef PPSCI_EMU
#define PPSCI_STEP_FAN 3  // the emulator has 4 "CUs": a small, odd fan-in makes its 8 workgroups a three-level tree with a ragged group
#else
#define PPSCI_STEP_FAN 16
#endif

struct StepTail {
  float* rows_w;       // [grid][per_tile]  == BwdArgs::wpart
  float* rows_s;       // [grid][psmall]    == BwdArgs::partials
  float* rows_l;       // [grid][n_res]     == EpiArgs::partials
  float* tree;         // rows of levels >= 1: [sum n_k][rowlen]
  unsigned* counters;  // one per group of every level; zero between launches
  float* grad;         // [P]
  float* loss_terms;   // [n_res]
  float* p;            // Adam: parameters (== FwdArgs::params), moments
  float* m;
  float* v;
  int grid, per_tile, psmall, n_res, rowlen;
  int lds;             // host side only: dynamic LDS bytes of the planned launch
  int accumulate, do_adam;
  float lr_t, beta1, beta2, eps_t, grad_scale;
  int fused;           // host side only: planned onto the fused tile kernel (taylor_fused.inc)
  int one_tail;        // host side only (with external == 1): ONE kernel behind the launch -- sums, grad (+)=, loss terms, Adam and the
                       // fragments of the updated hidden matrices (wgrad_reduce.hip wgrad_tail_kernel) -- instead of two
  int external;        // fused tile kernel: 1 = the launch stops at the workgroups' rows; the host issues the two reduction kernels
                       // (+ loss sum, Adam) behind it -- a 512-row tree of 48 KB rows is slower than those (taylor_api.hip);
                       // 2 = the launch runs the FIRST level of the tree (the last workgroup of every group of
                       // PPSCI_STEP_FAN sums the group's rows into `tree`, hidden behind the workgroups still computing) and
                       // the host issues ONE kernel behind it (sum of the level-1 rows, loss terms, Adam)
};

struct StepArgs {
  FwdArgs f;
  BwdArgs b;
  EpiArgs e;
  StepTail t;
};

// launch == 0: only plan (fills iters / grid / lds and *grid_out); launch == 1: plan + launch; launch == 2: launch an
// argument block planned earlier as it is (no occupancy / attribute queries: the per-step host cost is one launch).  PPSCI_E_UNSUPPORTED: this net /
// stream set has no one-launch kernel (the caller uses the separate launches).
int ppsci_step_run_tanh(StepArgs& a, void* stream, int launch, int* grid_out);
int ppsci_step_run_silu(StepArgs& a, void* stream, int launch, int* grid_out);
int ppsci_step_run_sin(StepArgs& a, void* stream, int launch, int* grid_out);
// the fused tile kernels (taylor_fused.inc, padded width 64): same contract; PPSCI_E_UNSUPPORTED without an error
// message when the net / stream set has none
int ppsci_fused_run_tanh(StepArgs& a, void* stream, int launch, int* grid_out);
int ppsci_fused_run_silu(StepArgs& a, void* stream, int launch, int* grid_out);
int ppsci_fused_run_sin(StepArgs& a, void* stream, int launch, int* grid_out);
// the same kernels for plans whose residual program is a compile-time table (a.e.static_id > 0; taylor_fused_static_<act>.hip)
int ppsci_fused_static_run_tanh(StepArgs& a, void* stream, int launch, int* grid_out);
int ppsci_fused_static_run_silu(StepArgs& a, void* stream, int launch, int* grid_out);
int ppsci_fused_static_run_sin(StepArgs& a, void* stream, int launch, int* grid_out);
