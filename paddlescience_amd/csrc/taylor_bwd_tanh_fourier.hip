// taylor_bwd_tanh_fourier.hip -- reverse kernels for tanh nets behind a FourierEmbedding layer.
#define PPSCI_ACT_ID PPSCI_ACT_TANH_FOURIER
#define PPSCI_BWD_RUN_NAME ppsci_bwd_run_tanh_fourier
#define PPSCI_BWD_RUN_NAME_B ppsci_bwd_run_tanh_fourier_b
#define PPSCI_BWD_WX_RUN_NAME ppsci_bwd_wx_run_tanh_fourier
#include "taylor_bwd.inc"
