// taylor_bwd_sigmoid.hip -- instantiates the reverse-sweep kernels for activation "sigmoid".
#define PPSCI_ACT_ID PPSCI_ACT_SIGMOID
#define PPSCI_BWD_RUN_NAME ppsci_bwd_run_sigmoid
#include "taylor_bwd.inc"
