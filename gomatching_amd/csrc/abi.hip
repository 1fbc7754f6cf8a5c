#include "common.h"

extern "C" int gom_abi_version(void) { return GOM_ABI_VERSION; }
extern "C" const char* gom_built_for_arch(void) { return "gfx950"; }
