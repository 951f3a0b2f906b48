// taylor_fwd_sigmoid.hip -- instantiates the Taylor-mode forward kernels for activation "sigmoid".
#define PPSCI_ACT_ID PPSCI_ACT_SIGMOID
#define PPSCI_FWD_RUN_NAME ppsci_fwd_run_sigmoid
#include "taylor_fwd.inc"
