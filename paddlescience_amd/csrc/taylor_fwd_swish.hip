// taylor_fwd_swish.hip -- Taylor-mode forward kernels for activation "swish" (trainable per-feature parameter).
#define PPSCI_ACT_HAS_PARAM 1
#define PPSCI_ACT_ID PPSCI_ACT_SWISH
#define PPSCI_FWD_RUN_NAME ppsci_fwd_run_swish
#include "taylor_fwd.inc"
