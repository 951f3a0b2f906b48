// taylor_bwd_elu.hip -- instantiates the reverse-sweep kernels for activation "elu".
#define PPSCI_ACT_ID PPSCI_ACT_ELU
#define PPSCI_BWD_RUN_NAME ppsci_bwd_run_elu
#define PPSCI_BWD_RUN_NAME_B ppsci_bwd_run_elu_b
#define PPSCI_BWD_WX_RUN_NAME ppsci_bwd_wx_run_elu
#include "taylor_bwd.inc"
