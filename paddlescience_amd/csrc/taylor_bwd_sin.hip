// taylor_bwd_sin.hip -- instantiates the reverse-sweep kernels for activation "sin".
#define PPSCI_ACT_ID PPSCI_ACT_SIN
#define PPSCI_BWD_RUN_NAME ppsci_bwd_run_sin
#include "taylor_bwd.inc"
