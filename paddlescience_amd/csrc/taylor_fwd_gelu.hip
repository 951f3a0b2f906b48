// taylor_fwd_gelu.hip -- instantiates the Taylor-mode forward kernels for activation "gelu".
#define PPSCI_ACT_ID PPSCI_ACT_GELU
#define PPSCI_FWD_RUN_NAME ppsci_fwd_run_gelu
#include "taylor_fwd.inc"
