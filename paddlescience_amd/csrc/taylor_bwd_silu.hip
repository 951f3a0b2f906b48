// taylor_bwd_silu.hip -- instantiates the reverse-sweep kernels for activation "silu".
#define PPSCI_ACT_ID PPSCI_ACT_SILU
#define PPSCI_BWD_RUN_NAME ppsci_bwd_run_silu
#include "taylor_bwd.inc"
