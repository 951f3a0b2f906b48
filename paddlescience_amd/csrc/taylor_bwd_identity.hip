// taylor_bwd_identity.hip -- instantiates the reverse-sweep kernels for activation "identity".
#define PPSCI_ACT_ID PPSCI_ACT_IDENTITY
#define PPSCI_BWD_RUN_NAME ppsci_bwd_run_identity
#include "taylor_bwd.inc"
