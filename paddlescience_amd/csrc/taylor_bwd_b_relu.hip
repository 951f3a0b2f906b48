// taylor_bwd_b_relu.hip -- part 1 of the reverse-sweep kernels for activation "relu": single-wave kernels of padded width 64 / 128.
#define PPSCI_ACT_ID PPSCI_ACT_RELU
#define PPSCI_BWD_PART 1
#define PPSCI_BWD_RUN_NAME ppsci_bwd_run_relu_b
#include "taylor_bwd.inc"
