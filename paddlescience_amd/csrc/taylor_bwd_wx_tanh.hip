// taylor_bwd_wx_tanh.hip -- register-accumulating feature-split reverse kernels (XDL pipe) for activation "tanh".
#define PPSCI_ACT_ID PPSCI_ACT_TANH
#define PPSCI_BWD_WX_RUN_NAME ppsci_bwd_wx_run_tanh
#include "taylor_bwd_wx_tu.inc"
