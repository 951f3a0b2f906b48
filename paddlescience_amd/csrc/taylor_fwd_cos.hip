// taylor_fwd_cos.hip -- instantiates the Taylor-mode forward kernels for activation "cos".
#define PPSCI_ACT_ID PPSCI_ACT_COS
#define PPSCI_FWD_RUN_NAME ppsci_fwd_run_cos
#include "taylor_fwd.inc"
