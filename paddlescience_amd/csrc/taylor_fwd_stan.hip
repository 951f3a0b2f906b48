// taylor_fwd_stan.hip -- Taylor-mode forward kernels for activation "stan" (trainable per-feature parameter).
#define PPSCI_ACT_HAS_PARAM 1
#define PPSCI_ACT_ID PPSCI_ACT_STAN
#define PPSCI_FWD_RUN_NAME ppsci_fwd_run_stan
#include "taylor_fwd.inc"
