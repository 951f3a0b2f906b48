// taylor_bwd_gelu.hip -- instantiates the reverse-sweep kernels for activation "gelu".
#define PPSCI_ACT_ID PPSCI_ACT_GELU
#define PPSCI_BWD_RUN_NAME ppsci_bwd_run_gelu
#include "taylor_bwd.inc"
