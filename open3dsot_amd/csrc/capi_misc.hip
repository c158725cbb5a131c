// capi_misc.hip -- library identification for libo3dsot_hip.so
#include "o3d_common.hpp"

extern "C" const char* o3d_version(void) { return "o3dsot-hip 0.1 gfx950"; }
