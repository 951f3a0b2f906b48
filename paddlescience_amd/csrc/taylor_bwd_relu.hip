// taylor_bwd_relu.hip -- instantiates the reverse-sweep kernels for activation "relu".
#define PPSCI_ACT_ID PPSCI_ACT_RELU
#define PPSCI_BWD_RUN_NAME ppsci_bwd_run_relu
#define PPSCI_BWD_RUN_NAME_B ppsci_bwd_run_relu_b
#define PPSCI_BWD_WX_RUN_NAME ppsci_bwd_wx_run_relu
#include "taylor_bwd.inc"
