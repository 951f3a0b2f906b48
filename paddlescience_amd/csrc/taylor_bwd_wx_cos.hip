// taylor_bwd_wx_cos.hip -- register-accumulating feature-split reverse kernels (XDL pipe) for activation "cos".
#define PPSCI_ACT_ID PPSCI_ACT_COS
#define PPSCI_BWD_WX_RUN_NAME ppsci_bwd_wx_run_cos
#include "taylor_bwd_wx_tu.inc"
