// taylor_step_sin.hip -- instantiates the one-launch step kernels for activation "sin".
#define PPSCI_ACT_ID PPSCI_ACT_SIN
#define PPSCI_STEP_RUN_NAME ppsci_step_run_sin
#include "taylor_step.inc"
