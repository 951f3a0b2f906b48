// taylor_bwd_wx_selu.hip -- register-accumulating feature-split reverse kernels (XDL pipe) for activation "selu".
#define PPSCI_ACT_ID PPSCI_ACT_SELU
#define PPSCI_BWD_WX_RUN_NAME ppsci_bwd_wx_run_selu
#include "taylor_bwd_wx_tu.inc"
