// taylor_bwd_b_sin.hip -- part 1 of the reverse-sweep kernels for activation "sin": single-wave kernels of padded width 64 / 128.
#define PPSCI_ACT_ID PPSCI_ACT_SIN
#define PPSCI_BWD_PART 1
#define PPSCI_BWD_RUN_NAME ppsci_bwd_run_sin_b
#include "taylor_bwd.inc"
