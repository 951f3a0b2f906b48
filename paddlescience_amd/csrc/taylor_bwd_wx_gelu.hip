// taylor_bwd_wx_gelu.hip -- register-accumulating feature-split reverse kernels (XDL pipe) for activation "gelu".
#define PPSCI_ACT_ID PPSCI_ACT_GELU
#define PPSCI_BWD_WX_RUN_NAME ppsci_bwd_wx_run_gelu
#include "taylor_bwd_wx_tu.inc"
