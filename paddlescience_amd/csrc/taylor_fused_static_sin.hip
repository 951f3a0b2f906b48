// taylor_fused_static_sin.hip -- the fused tile kernels of plans whose residual program is a compile-time table
// (csrc/epi_static.h, epi_static_programs.h), activation "sin".
#define PPSCI_ACT_ID PPSCI_ACT_SIN
#define PPSCI_FUSED_STATIC 1
#define PPSCI_FUSED_RUN_NAME ppsci_fused_static_run_sin
#include "taylor_fused.inc"
