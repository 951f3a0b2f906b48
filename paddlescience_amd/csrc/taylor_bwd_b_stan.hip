// taylor_bwd_b_stan.hip -- part 1 of the reverse-sweep kernels for activation "stan": single-wave kernels of padded width 64 / 128.
#define PPSCI_ACT_HAS_PARAM 1
#define PPSCI_ACT_ID PPSCI_ACT_STAN
#define PPSCI_BWD_PART 1
#define PPSCI_BWD_RUN_NAME ppsci_bwd_run_stan_b
#include "taylor_bwd.inc"
