// taylor_bwd_sigmoid.hip -- instantiates the reverse-sweep kernels for activation "sigmoid".
#define PPSCI_ACT_ID PPSCI_ACT_SIGMOID
#define PPSCI_BWD_RUN_NAME ppsci_bwd_run_sigmoid
#define PPSCI_BWD_RUN_NAME_B ppsci_bwd_run_sigmoid_b
#define PPSCI_BWD_WX_RUN_NAME ppsci_bwd_wx_run_sigmoid
#include "taylor_bwd.inc"
