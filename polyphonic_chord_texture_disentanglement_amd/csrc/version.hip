// version.hip -- build identification of libptvae_hip.so
#include "../../include/ptvae_hip.h"
extern "C" const char* ptv_arch(void) { return "gfx950"; }
extern "C" int ptv_abi_version(void) { return 1; }
