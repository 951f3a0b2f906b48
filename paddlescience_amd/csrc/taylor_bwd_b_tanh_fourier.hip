// taylor_bwd_b_tanh_fourier.hip -- part 1 of the reverse-sweep kernels for activation "tanh_fourier": single-wave kernels of padded width 64 / 128.
#define PPSCI_ACT_ID PPSCI_ACT_TANH_FOURIER
#define PPSCI_BWD_PART 1
#define PPSCI_BWD_RUN_NAME ppsci_bwd_run_tanh_fourier_b
#include "taylor_bwd.inc"
