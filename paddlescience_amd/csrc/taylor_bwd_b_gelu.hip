// taylor_bwd_b_gelu.hip -- part 1 of the reverse-sweep kernels for activation "gelu": single-wave kernels of padded width 64 / 128.
#define PPSCI_ACT_ID PPSCI_ACT_GELU
#define PPSCI_BWD_PART 1
#define PPSCI_BWD_RUN_NAME ppsci_bwd_run_gelu_b
#include "taylor_bwd.inc"
