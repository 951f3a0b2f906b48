// taylor_bwd_identity.hip -- instantiates the reverse-sweep kernels for activation "identity".
#define PPSCI_ACT_ID PPSCI_ACT_IDENTITY
#define PPSCI_BWD_RUN_NAME ppsci_bwd_run_identity
#define PPSCI_BWD_RUN_NAME_B ppsci_bwd_run_identity_b
#define PPSCI_BWD_WX_RUN_NAME ppsci_bwd_wx_run_identity
#include "taylor_bwd.inc"
