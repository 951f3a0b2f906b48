// taylor_fwd_silu.hip -- instantiates the Taylor-mode forward kernels for activation "silu".
#define PPSCI_ACT_ID PPSCI_ACT_SILU
#define PPSCI_FWD_RUN_NAME ppsci_fwd_run_silu
#include "taylor_fwd.inc"
