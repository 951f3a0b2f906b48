// taylor_fwd_elu.hip -- instantiates the Taylor-mode forward kernels for activation "elu".
#define PPSCI_ACT_ID PPSCI_ACT_ELU
#define PPSCI_FWD_RUN_NAME ppsci_fwd_run_elu
#include "taylor_fwd.inc"
