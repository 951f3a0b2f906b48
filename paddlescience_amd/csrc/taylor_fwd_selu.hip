// taylor_fwd_selu.hip -- instantiates the Taylor-mode forward kernels for activation "selu".
#define PPSCI_ACT_ID PPSCI_ACT_SELU
#define PPSCI_FWD_RUN_NAME ppsci_fwd_run_selu
#include "taylor_fwd.inc"
