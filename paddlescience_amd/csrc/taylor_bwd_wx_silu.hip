// taylor_bwd_wx_silu.hip -- register-accumulating feature-split reverse kernels (XDL pipe) for activation "silu".
#define PPSCI_ACT_ID PPSCI_ACT_SILU
#define PPSCI_BWD_WX_RUN_NAME ppsci_bwd_wx_run_silu
#include "taylor_bwd_wx_tu.inc"
