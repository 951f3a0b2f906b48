// taylor_fwd_leaky_relu.hip -- instantiates the Taylor-mode forward kernels for activation "leaky_relu".
#define PPSCI_ACT_ID PPSCI_ACT_LEAKY_RELU
#define PPSCI_FWD_RUN_NAME ppsci_fwd_run_leaky_relu
#include "taylor_fwd.inc"
