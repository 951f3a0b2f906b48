// taylor_fwd_tanh_fourier.hip -- Taylor-mode forward kernels for tanh nets behind a FourierEmbedding layer.
#define PPSCI_ACT_ID PPSCI_ACT_TANH_FOURIER
#define PPSCI_FWD_RUN_NAME ppsci_fwd_run_tanh_fourier
#include "taylor_fwd.inc"
