// taylor_fwd_tanh.hip -- instantiates the Taylor-mode forward kernels for activation "tanh".
#define PPSCI_ACT_ID PPSCI_ACT_TANH
#define PPSCI_FWD_RUN_NAME ppsci_fwd_run_tanh
#include "taylor_fwd.inc"
