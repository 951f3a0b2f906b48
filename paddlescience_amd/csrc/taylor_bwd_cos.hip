// taylor_bwd_cos.hip -- instantiates the reverse-sweep kernels for activation "cos".
#define PPSCI_ACT_ID PPSCI_ACT_COS
#define PPSCI_BWD_RUN_NAME ppsci_bwd_run_cos
#define PPSCI_BWD_RUN_NAME_B ppsci_bwd_run_cos_b
#define PPSCI_BWD_WX_RUN_NAME ppsci_bwd_wx_run_cos
#include "taylor_bwd.inc"
