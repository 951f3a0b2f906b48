// taylor_bwd_silu.hip -- instantiates the reverse-sweep kernels for activation "silu".
#define PPSCI_ACT_ID PPSCI_ACT_SILU
#define PPSCI_BWD_RUN_NAME ppsci_bwd_run_silu
#define PPSCI_BWD_RUN_NAME_B ppsci_bwd_run_silu_b
#define PPSCI_BWD_WX_RUN_NAME ppsci_bwd_wx_run_silu
#include "taylor_bwd.inc"
