// taylor_bwd_wx_leaky_relu.hip -- register-accumulating feature-split reverse kernels (XDL pipe) for activation "leaky_relu".
#define PPSCI_ACT_ID PPSCI_ACT_LEAKY_RELU
#define PPSCI_BWD_WX_RUN_NAME ppsci_bwd_wx_run_leaky_relu
#include "taylor_bwd_wx_tu.inc"
