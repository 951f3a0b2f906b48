// taylor_bwd_cos.hip -- instantiates the reverse-sweep kernels for activation "cos".
#define PPSCI_ACT_ID PPSCI_ACT_COS
#define PPSCI_BWD_RUN_NAME ppsci_bwd_run_cos
#include "taylor_bwd.inc"
