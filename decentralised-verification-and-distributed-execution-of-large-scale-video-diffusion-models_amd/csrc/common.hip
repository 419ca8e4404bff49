#include "vdx_common.h"
thread_local char g_vdx_err[512] = {0};
extern "C" const char* vdx_last_error(void) { return g_vdx_err; }
extern "C" int vdx_version(void) { return 1; }
int g_vdx_reserved_cus = 0;
extern "C" int vdx_set_reserved_cus(int n) {
    if (n < 0 || n > vdx_num_cus() - 8) return vdx_fail("set_reserved_cus: %d outside [0, %d]", n, vdx_num_cus() - 8);
    g_vdx_reserved_cus = n;
    return 0;
}
extern "C" int vdx_reserved_cus(void) { return g_vdx_reserved_cus; }
extern "C" int vdx_persistent_grid_cus(void) { return vdx_grid_cus(); }
// 0 = the product build.  Bit per translation unit that was compiled with a lab macro (stamps / ablations).
extern "C" int vdx_lab_gemm(void); extern "C" int vdx_lab_gemm_ws(void); extern "C" int vdx_lab_tattn_fused(void);
extern "C" int vdx_lab_tattn2(void); extern "C" int vdx_lab_flash(void); extern "C" int vdx_lab_ff_fused(void);
extern "C" int vdx_lab_conv_fused(void); extern "C" int vdx_lab_xattn(void);
extern "C" int vdx_build_flags(void) {
    return vdx_lab_gemm() | vdx_lab_gemm_ws() | vdx_lab_tattn_fused() | vdx_lab_tattn2() | vdx_lab_flash() | vdx_lab_ff_fused() |
           vdx_lab_conv_fused() | vdx_lab_xattn();
}
