// taylor_bwd_stan.hip -- reverse kernels for activation "stan" (trainable per-feature parameter).
#define PPSCI_ACT_HAS_PARAM 1
#define PPSCI_ACT_ID PPSCI_ACT_STAN
#define PPSCI_BWD_RUN_NAME ppsci_bwd_run_stan
#define PPSCI_BWD_RUN_NAME_B ppsci_bwd_run_stan_b
#include "taylor_bwd.inc"
