// taylor_fwd_identity.hip -- instantiates the Taylor-mode forward kernels for activation "identity".
#define PPSCI_ACT_ID PPSCI_ACT_IDENTITY
#define PPSCI_FWD_RUN_NAME ppsci_fwd_run_identity
#include "taylor_fwd.inc"
