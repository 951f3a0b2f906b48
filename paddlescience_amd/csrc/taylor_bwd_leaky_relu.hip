// taylor_bwd_leaky_relu.hip -- instantiates the reverse-sweep kernels for activation "leaky_relu".
#define PPSCI_ACT_ID PPSCI_ACT_LEAKY_RELU
#define PPSCI_BWD_RUN_NAME ppsci_bwd_run_leaky_relu
#define PPSCI_BWD_RUN_NAME_B ppsci_bwd_run_leaky_relu_b
#define PPSCI_BWD_WX_RUN_NAME ppsci_bwd_wx_run_leaky_relu
#include "taylor_bwd.inc"
