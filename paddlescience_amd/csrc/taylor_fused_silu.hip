// taylor_fused_silu.hip -- instantiates the fused tile kernels (forward -> residual program -> reverse per 16-point
// tile, nothing of a tile leaving the CU) for activation "silu".
#define PPSCI_ACT_ID PPSCI_ACT_SILU
#define PPSCI_FUSED_RUN_NAME ppsci_fused_run_silu
#include "taylor_fused.inc"
