// Shared device/host helpers for libvdx_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/vdx.h"

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ---- error plumbing ---------------------------------------------------------------------
extern thread_local char g_vdx_err[512];
static inline int vdx_fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_vdx_err, sizeof(g_vdx_err), fmt, ap);
    va_end(ap);
    return -1;
}
#define VDX_CHECK(cond, ...)                      \
    do {                                          \
        if (!(cond)) return vdx_fail(__VA_ARGS__); \
    } while (0)
static inline int vdx_launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return vdx_fail("%s: launch failed: %s", what, hipGetErrorString(e));
    return 0;
}

// compute units of the current device (persistent grids, tile-round pricing); 256 on MI355X
static inline int vdx_num_cus() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
            v = 256;
        return v;
    }();
    return n;
}

// Compute units a PERSISTENT grid may fill: all of them, minus the reserve the shard store asks for when parameter gathers
// run beside the step (RCCL's channel kernels each hold a CU: an exact-fit grid of 256 blocks would leave the displaced
// blocks to a second round).  A multiple of 8 (the XCD-aware block maps).  Results never depend on it: every persistent
// kernel computes a tile's bits from the tile alone.  `vdx_set_reserved_cus` (common.hip), include/vdx.h.
extern int g_vdx_reserved_cus;
static inline int vdx_grid_cus() {
    int n = (vdx_num_cus() - g_vdx_reserved_cus) & ~7;
    return n < 8 ? 8 : n;
}

// ---- device helpers ---------------------------------------------------------------------
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// exact (erf) GELU.  erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, far below fp16
// output rounding): one v_rcp, one v_exp and 6 FMAs instead of libm's branchy erff — the GEGLU
// epilogue evaluates it 80 times per lane per tile, which otherwise outweighs the tile's MFMAs.
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
    float p = 1.061405429f;
    p = p * t - 1.453152027f;
    p = p * t + 1.421413741f;
    p = p * t - 0.284496736f;
    p = p * t + 0.254829592f;
    const float e = __builtin_amdgcn_exp2f(-1.44269504088896341f * ax * ax);
    const float r = 1.0f - p * t * e;
    return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf_f(float x) {
    return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752f));
}
// GELU through a table of the normal CDF in LDS (the GEGLU epilogues).  Evaluating erf costs ~17 VALU
// instructions, two of them quarter-rate, per output: at K = 320 that is as many issue cycles as the
// output's MFMAs, and the feed-forward GEMMs were VALU-bound.  The table holds (Phi(x_i), Phi(x_i+1) -
// Phi(x_i)) on [-5, 5) in steps of 10/1024; linear interpolation is exact to 3e-6 (h^2/8 * max|Phi''|),
// far below the fp16 rounding of the result, and costs 8 VALU instructions and one ds_read_b64.
#define GELU_TAB_N 1024
#define GELU_TAB_BYTES (GELU_TAB_N * 8)
__device__ __forceinline__ void gelu_tab_init(float2* tab, int tid, int nthreads) {
    const float h = 10.0f / GELU_TAB_N;
    for (int i = tid; i < GELU_TAB_N; i += nthreads) {
        const float x0 = -5.0f + i * h;
        const float p0 = 0.5f * (1.0f + erf_fast(x0 * 0.70710678118654752f));
        const float p1 = 0.5f * (1.0f + erf_fast((x0 + h) * 0.70710678118654752f));
        tab[i] = make_float2(p0, p1 - p0);
    }
}
__device__ __forceinline__ float gelu_tab(float x, const float2* tab) {
    float u = fmaf(x, GELU_TAB_N / 10.0f, GELU_TAB_N / 2.0f);
    u = fminf(fmaxf(u, 0.0f), GELU_TAB_N - 0.001f);
    const float fi = floorf(u);
    const float2 e = tab[(int)fi];
    return x * fmaf(u - fi, e.y, e.x);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// XCD-aware bijective remap of a linear block id (8 XCDs; blocks b and b+8 share an XCD):
// gives every XCD a contiguous range of logical tiles so neighbouring tiles share its L2.
__device__ __forceinline__ int xcd_remap(int id, int n) {
    const int q = n >> 3, r = n & 7, x = id & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
}

// 128 zero bytes per translation unit: padding taps / rows past the end read from here, so
// gather loads stay unconditional (no exec-masked branches around loads).
static __device__ __attribute__((aligned(128))) u32x4 g_zero_page[8];
