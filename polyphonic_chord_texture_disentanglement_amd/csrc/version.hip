// version.hip -- build identification of libptvae_hip.so
#include "../../include/ptvae_hip.h"
extern "C" const char* ptv_arch(void) { return "gfx950"; }
// 2 (round 3): ptv_step_params / ptv_ordered_reductions / ptv_wgrad_mode added; ptv_gemm dtypes bit 3 (column-blocked C); the
// row-partitioned GRU pair takes gc / ext column-blocked and keeps its gate planes unit-blocked
extern "C" int ptv_abi_version(void) { return 2; }
