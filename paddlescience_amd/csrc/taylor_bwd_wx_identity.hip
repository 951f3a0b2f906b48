// taylor_bwd_wx_identity.hip -- register-accumulating feature-split reverse kernels (XDL pipe) for activation "identity".
#define PPSCI_ACT_ID PPSCI_ACT_IDENTITY
#define PPSCI_BWD_WX_RUN_NAME ppsci_bwd_wx_run_identity
#include "taylor_bwd_wx_tu.inc"
