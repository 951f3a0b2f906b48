// taylor_fused_tanh.hip -- instantiates the fused tile kernels (forward -> residual program -> reverse per 16-point
// tile, nothing of a tile leaving the CU) for activation "tanh".
#define PPSCI_ACT_ID PPSCI_ACT_TANH
#define PPSCI_FUSED_RUN_NAME ppsci_fused_run_tanh
#include "taylor_fused.inc"
