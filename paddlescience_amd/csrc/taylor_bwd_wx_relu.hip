// taylor_bwd_wx_relu.hip -- register-accumulating feature-split reverse kernels (XDL pipe) for activation "relu".
#define PPSCI_ACT_ID PPSCI_ACT_RELU
#define PPSCI_BWD_WX_RUN_NAME ppsci_bwd_wx_run_relu
#include "taylor_bwd_wx_tu.inc"
