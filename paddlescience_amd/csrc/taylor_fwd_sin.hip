// taylor_fwd_sin.hip -- instantiates the Taylor-mode forward kernels for activation "sin".
#define PPSCI_ACT_ID PPSCI_ACT_SIN
#define PPSCI_FWD_RUN_NAME ppsci_fwd_run_sin
#include "taylor_fwd.inc"
