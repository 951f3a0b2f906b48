// taylor_bwd_elu.hip -- instantiates the reverse-sweep kernels for activation "elu".
#define PPSCI_ACT_ID PPSCI_ACT_ELU
#define PPSCI_BWD_RUN_NAME ppsci_bwd_run_elu
#include "taylor_bwd.inc"
