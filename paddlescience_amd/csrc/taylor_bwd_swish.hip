// taylor_bwd_swish.hip -- reverse kernels for activation "swish" (trainable per-feature parameter).
#define PPSCI_ACT_HAS_PARAM 1
#define PPSCI_ACT_ID PPSCI_ACT_SWISH
#define PPSCI_BWD_RUN_NAME ppsci_bwd_run_swish
#define PPSCI_BWD_RUN_NAME_B ppsci_bwd_run_swish_b
#include "taylor_bwd.inc"
