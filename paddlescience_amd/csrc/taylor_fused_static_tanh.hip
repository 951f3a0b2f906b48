// taylor_fused_static_tanh.hip -- the fused tile kernels of plans whose residual program is a compile-time table
// (csrc/epi_static.h, epi_static_programs.h), activation "tanh".
#define PPSCI_ACT_ID PPSCI_ACT_TANH
#define PPSCI_FUSED_STATIC 1
#define PPSCI_FUSED_RUN_NAME ppsci_fused_static_run_tanh
#include "taylor_fused.inc"
