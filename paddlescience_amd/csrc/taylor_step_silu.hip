// taylor_step_silu.hip -- instantiates the one-launch step kernels for activation "silu".
#define PPSCI_ACT_ID PPSCI_ACT_SILU
#define PPSCI_STEP_RUN_NAME ppsci_step_run_silu
#include "taylor_step.inc"
