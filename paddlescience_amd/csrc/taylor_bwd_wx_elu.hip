// taylor_bwd_wx_elu.hip -- register-accumulating feature-split reverse kernels (XDL pipe) for activation "elu".
#define PPSCI_ACT_ID PPSCI_ACT_ELU
#define PPSCI_BWD_WX_RUN_NAME ppsci_bwd_wx_run_elu
#include "taylor_bwd_wx_tu.inc"
