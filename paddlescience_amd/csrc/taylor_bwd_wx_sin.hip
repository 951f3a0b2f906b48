// taylor_bwd_wx_sin.hip -- register-accumulating feature-split reverse kernels (XDL pipe) for activation "sin".
#define PPSCI_ACT_ID PPSCI_ACT_SIN
#define PPSCI_BWD_WX_RUN_NAME ppsci_bwd_wx_run_sin
#include "taylor_bwd_wx_tu.inc"
