// taylor_bwd_gelu.hip -- instantiates the reverse-sweep kernels for activation "gelu".
#define PPSCI_ACT_ID PPSCI_ACT_GELU
#define PPSCI_BWD_RUN_NAME ppsci_bwd_run_gelu
#define PPSCI_BWD_RUN_NAME_B ppsci_bwd_run_gelu_b
#define PPSCI_BWD_WX_RUN_NAME ppsci_bwd_wx_run_gelu
#include "taylor_bwd.inc"
