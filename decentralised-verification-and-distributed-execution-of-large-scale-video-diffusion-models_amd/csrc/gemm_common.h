// gemm_common.h — parameter block and epilogue shared by the GEMM kernels (gemm.hip, gemm_ring.hip).
#pragma once
#include "vdx_common.h"

struct GemmP {
    const f16 *a, *a2, *w, *bias, *bias2, *res;
    f16* out;
    int M, N, K, c1, c2;
    int lda, lda2, ldo, ldr;
    int h_in, w_in, h_out, w_out, stride;
    int ups;          // conv3x3 source: 0 as is, 1 nearest x2, 2 nearest to (h_up, w_up) (torch's rule: floor(dst * in / out) in fp32)
    int h_up, w_up;   // extent of the (virtually) upsampled source
    float usy, usx;   // ups == 2: in / out as fp32
    int frames, hw, rpb2, ldb2;
    int ntn, ntm;
    int m_begin;      // first row computed (tiles start here)
    int ksplit;       // > 1: every tile is computed by `ksplit` blocks, each over a slice of the K tiles, which leave their
    float* partial;   //      fp32 accumulators in a slab of `partial`; vdx_gemm_reduce_kernel sums the slabs and runs the epilogue
    int wset_rows;    // > 0: one weight set (w + s*N*K, wset_bias + s*N) per `wset_rows` rows (weights-stationary kernels only)
    const float* wset_bias;
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// ---- epilogue shared by the kernels: lane holds rows mb + 16i + frow, and per accumulator pair
// (2a, 2a+1) the 8 consecutive columns nb + 32a + 8*fq .. +7.
// All side inputs of one row group (bias, time-embedding row, residual) are requested TOGETHER and
// waited for once: issued one by one next to their use, each load was a full serialized memory
// round trip (20 per lane per tile), which is what the short-K layers' time went to.
// GEGLU kernels pass the block's GELU table (GELU_TAB_BYTES of LDS, filled by gelu_tab_init before a barrier).
template <int TM, int TN, bool GEGLU>
__device__ __forceinline__ void gemm_epilogue(const GemmP& p, f32x4 (&acc)[TM][TN], int mb, int nb, int frow, int fq,
                                              const float2* gelu = nullptr) {
    constexpr int NA = TN / 2;
    const f16* zp = (const f16*)g_zero_page;
    // per-column bias: the same for every row group -> loaded once
    f16x8 bv[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int j = 0; j < 8; ++j) bv[a][j] = (f16)0.f;
    if (p.bias) {                                   // absent inputs cost nothing (uniform branches)
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const int n = nb + a * 32 + fq * 8;
            bv[a] = *(const f16x8*)(n < p.N ? p.bias + n : zp);
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = mb + i * 16 + frow;
        const bool row_ok = m < p.M;
        f16x8 rv[NA], b2v[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int j = 0; j < 8; ++j) rv[a][j] = b2v[a][j] = (f16)0.f;
        if (!GEGLU && p.res) {
            const f16* rrow = p.res + (size_t)(row_ok ? m : 0) * p.ldr;
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const int n = nb + a * 32 + fq * 8;
                rv[a] = *(const f16x8*)((row_ok && n < p.N) ? rrow + n : zp);
            }
        }
        if (!GEGLU && p.bias2) {
            const f16* b2row = p.bias2 + (size_t)((row_ok ? m : 0) / p.rpb2) * p.ldb2;
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const int n = nb + a * 32 + fq * 8;
                b2v[a] = *(const f16x8*)((row_ok && n < p.N) ? b2row + n : zp);
            }
        }
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const int n = nb + a * 32 + fq * 8;
            float v[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = acc[i][2 * a][j] + (float)bv[a][j];
                v[4 + j] = acc[i][2 * a + 1][j] + (float)bv[a][4 + j];
            }
            if (GEGLU) {
                f16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (f16)(v[j] * gelu_tab(v[4 + j], gelu));
                if (row_ok && n < p.N) *(f16x4*)(p.out + (size_t)m * p.ldo + (n >> 1)) = o;
            } else {
                f16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (f16)(v[j] + (float)b2v[a][j] + (float)rv[a][j]);
                if (row_ok && n < p.N) *(f16x8*)(p.out + (size_t)m * p.ldo + n) = o;
            }
        }
    }
}

// K-step-32 LDS-ring kernels (gemm_ring.hip); mode = VDX_GEMM_*; variant 0 = 256x320 tile with a
// four-stage ring (one block per CU), variant 1 = 128x320 tile, two stages, two blocks per CU.
int vdx_gemm_ring_launch(const GemmP& p, int mode, bool geglu, int variant, hipStream_t st);

// Logical tile id -> (m tile, n tile).  Up to 4 column tiles: n fastest (the tiles of a row block share its
// activations).  More (N >= 1600: the level-2/3 GEGLU and q|k|v layers): column PANELS of 4 n tiles, m-major inside
// a panel, so the 32 tiles an XCD works on at a time are 8 row blocks x 4 column tiles — 8.5 MB of first-touch
// bytes instead of 27 MB at K = 1280 — and a panel's weights stay in that XCD's L2 while the rows sweep past
// (PMC before: 2.6 GB of L2 fills per level-2 GEGLU launch for 0.24 GB of operands, fabric-bound at 4.5 TB/s).
__device__ __forceinline__ void gemm_tile_of(int bid, int ntm, int ntn, int& mt, int& nt) {
    if (ntn <= 4) {
        mt = bid / ntn;
        nt = bid - mt * ntn;
        return;
    }
    const int per_panel = ntm * 4;
    const int panel = bid / per_panel, r = bid - panel * per_panel;
    const int width = min(4, ntn - panel * 4);          // the last panel may be narrower
    mt = r / width;
    nt = panel * 4 + (r - mt * width);
}

// Weights-stationary streaming kernels for short-K Linear layers (gemm_ws.hip): family 0 = shape not covered.
int vdx_gemm_ws_family(const GemmP& p, int mode, bool geglu);
int vdx_gemm_ws_launch(const GemmP& p, int family, bool geglu, hipStream_t st);
