// taylor_bwd_wx_tanh_fourier.hip -- register-accumulating feature-split reverse kernels (XDL pipe) for activation "tanh_fourier".
#define PPSCI_ACT_ID PPSCI_ACT_TANH_FOURIER
#define PPSCI_BWD_WX_RUN_NAME ppsci_bwd_wx_run_tanh_fourier
#include "taylor_bwd_wx_tu.inc"
