// taylor_fused_sin.hip -- instantiates the fused tile kernels (forward -> residual program -> reverse per 16-point
// tile, nothing of a tile leaving the CU) for activation "sin".
#define PPSCI_ACT_ID PPSCI_ACT_SIN
#define PPSCI_FUSED_RUN_NAME ppsci_fused_run_sin
#include "taylor_fused.inc"
