// taylor_bwd_wx_sigmoid.hip -- register-accumulating feature-split reverse kernels (XDL pipe) for activation "sigmoid".
#define PPSCI_ACT_ID PPSCI_ACT_SIGMOID
#define PPSCI_BWD_WX_RUN_NAME ppsci_bwd_wx_run_sigmoid
#include "taylor_bwd_wx_tu.inc"
