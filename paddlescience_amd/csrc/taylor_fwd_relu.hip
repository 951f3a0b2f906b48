// taylor_fwd_relu.hip -- instantiates the Taylor-mode forward kernels for activation "relu".
#define PPSCI_ACT_ID PPSCI_ACT_RELU
#define PPSCI_FWD_RUN_NAME ppsci_fwd_run_relu
#include "taylor_fwd.inc"
