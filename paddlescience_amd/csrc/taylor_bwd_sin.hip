// taylor_bwd_sin.hip -- instantiates the reverse-sweep kernels for activation "sin".
#define PPSCI_ACT_ID PPSCI_ACT_SIN
#define PPSCI_BWD_RUN_NAME ppsci_bwd_run_sin
#define PPSCI_BWD_RUN_NAME_B ppsci_bwd_run_sin_b
#define PPSCI_BWD_WX_RUN_NAME ppsci_bwd_wx_run_sin
#include "taylor_bwd.inc"
