// taylor_step_tanh.hip -- instantiates the one-launch step kernels for activation "tanh".
#define PPSCI_ACT_ID PPSCI_ACT_TANH
#define PPSCI_STEP_RUN_NAME ppsci_step_run_tanh
#include "taylor_step.inc"
