// taylor_bwd_selu.hip -- instantiates the reverse-sweep kernels for activation "selu".
#define PPSCI_ACT_ID PPSCI_ACT_SELU
#define PPSCI_BWD_RUN_NAME ppsci_bwd_run_selu
#define PPSCI_BWD_RUN_NAME_B ppsci_bwd_run_selu_b
#define PPSCI_BWD_WX_RUN_NAME ppsci_bwd_wx_run_selu
#include "taylor_bwd.inc"
