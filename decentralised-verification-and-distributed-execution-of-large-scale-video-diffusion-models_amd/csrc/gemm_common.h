// gemm_common.h — parameter block and epilogue shared by the GEMM kernels (gemm.hip, gemm_ring.hip).
#pragma once
#include "vdx_common.h"

struct GemmP {
    const f16 *a, *a2, *w, *bias, *bias2, *res;
    f16* out;
    int M, N, K, c1, c2;
    int lda, lda2, ldo, ldr;
    int h_in, w_in, h_out, w_out, stride, ups;
    int frames, hw, rpb2, ldb2;
    int ntn;
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// ---- epilogue shared by both kernels: lane holds rows mb + 16i + frow, and per accumulator pair
// (2a, 2a+1) the 8 consecutive columns nb + 32a + 8*fq .. +7 --------------------------------------
template <int TM, int TN, bool GEGLU>
__device__ __forceinline__ void gemm_epilogue(const GemmP& p, f32x4 (&acc)[TM][TN], int mb, int nb, int frow, int fq) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = mb + i * 16 + frow;
        if (m >= p.M) continue;
        const f16* b2row = p.bias2 ? p.bias2 + (size_t)(m / p.rpb2) * p.ldb2 : nullptr;
#pragma unroll
        for (int a = 0; a < TN / 2; ++a) {
            const int n = nb + a * 32 + fq * 8;
            if (n >= p.N) continue;
            float v[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = acc[i][2 * a][j];
                v[4 + j] = acc[i][2 * a + 1][j];
            }
            if (p.bias) {
                const f16x8 b = *(const f16x8*)(p.bias + n);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += (float)b[j];
            }
            if (GEGLU) {
                f16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (f16)(v[j] * gelu_erf_f(v[4 + j]));
                *(f16x4*)(p.out + (size_t)m * p.ldo + (n >> 1)) = o;
            } else {
                if (b2row) {
                    const f16x8 b = *(const f16x8*)(b2row + n);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += (float)b[j];
                }
                if (p.res) {
                    const f16x8 r = *(const f16x8*)(p.res + (size_t)m * p.ldr + n);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += (float)r[j];
                }
                f16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (f16)v[j];
                *(f16x8*)(p.out + (size_t)m * p.ldo + n) = o;
            }
        }
    }
}

// K-step-32 LDS-ring kernels (gemm_ring.hip); mode = VDX_GEMM_*; variant 0 = 256x320 tile with a
// four-stage ring (one block per CU), variant 1 = 128x320 tile, two stages, two blocks per CU.
int vdx_gemm_ring_launch(const GemmP& p, int mode, bool geglu, int variant, hipStream_t st);
