// taylor_bwd_tanh.hip -- instantiates the reverse-sweep kernels for activation "tanh".
#define PPSCI_ACT_ID PPSCI_ACT_TANH
#define PPSCI_BWD_RUN_NAME ppsci_bwd_run_tanh
#define PPSCI_BWD_RUN_NAME_B ppsci_bwd_run_tanh_b
#define PPSCI_BWD_WX_RUN_NAME ppsci_bwd_wx_run_tanh
#include "taylor_bwd.inc"
