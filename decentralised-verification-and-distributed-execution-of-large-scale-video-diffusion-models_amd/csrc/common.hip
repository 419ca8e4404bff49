#include "vdx_common.h"
thread_local char g_vdx_err[512] = {0};
extern "C" const char* vdx_last_error(void) { return g_vdx_err; }
extern "C" int vdx_version(void) { return 1; }
