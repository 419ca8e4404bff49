// conv_fused.hip — K1 (SURVEY.md §2.3, row ResnetBlock2D; App. A.3): the 3x3 convolution of a ResNet block with the GroupNorm
// apply and the SiLU in front of it FUSED IN,
//
//     out[n, y, x, :] = bias + temb[n] + residual + sum_{ky, kx} W_{ky,kx} . silu( cat(a, a2)[n, y+ky-1, x+kx-1, :] * scale[n, :] + shift[n, :] )
//
// (zero for pixels outside the image: the padding applies to the NORMALISED tensor), reached twice per ResnetBlock2D from
// fsdp_chunked_coherent.py:140 (conv1: optional two-source channel concat of the up blocks + the time-embedding row; conv2:
// the residual / shortcut).  (scale, shift) per (image, channel) = what the statistics pass leaves behind (norm.hip, 4-D
// GroupNorm: a sample is one frame).  It replaces the apply pass (one read + one write of the [rows][C1 + C2] activation)
// followed by gemm_kernel<.., MODE 1, ..>, which gathers the normalised rows nine times (once per tap).
//
// The temporal sibling is K3 (tconv_fused.hip); the structure is the same.  Tile = a patch of 6 x 32 pixels of ONE image x
// 320 output channels (192 rows, rows ordered (y, x)).  Per 64-channel slice the block stages ONE image of the patch and its
// one-pixel halo — 8 x 34 pixels x 64 channels, 34 KB — by LDS-DMA (zero page outside the image), normalises it IN PLACE in
// LDS (once per element, not once per tap) while the previous slice's last tap runs on the matrix cores, and the nine taps
// read their activation fragments from that one image at row offsets (ky * 34 + kx).  A fragment's 16 lanes read 16
// consecutive image columns; the 16-byte chunk swizzle is the image COLUMN's low three bits, so a tap's swizzle depends on kx
// alone and a fragment address is one add on precomputed lane offsets.  Weights per (slice, tap) as [320][64] tiles,
// double-buffered, in gemm.hip's layout / swizzle / row permutation (the same packed weights: K = (c / 64) * 576 + tap * 64 +
// c % 64); accumulators, weight fragments and the 16-byte epilogue stores are gemm.hip's.
// LDS: 2 x 34 KB (image) + 2 x 40 KB (weights) = 148 KB, one 512-thread block per CU.
#include "gemm_common.h"

struct C1P {
    const f16 *a, *a2, *w, *bias, *bias2, *res;
    f16* out;
    const float* ab;          // [n_img][C][2]: scale, shift  (C = c1 + c2)
    int lda, lda2, ldo, ldr, ldb2, rpb2;
    int n_img, h, w_, c1, c2, N;
    int npy, npx, ntn;        // patches per image in y / x, column tiles (N / 320)
};

constexpr int C1_PH = 6, C1_PW = 32, C1_IW = C1_PW + 2, C1_IROWS = (C1_PH + 2) * C1_IW;     // 272 image rows

__global__ __launch_bounds__(512) void conv3x3_gn_kernel(const C1P p) {
    constexpr int BN = 320, WM = 4, WN = 2;
    constexpr int TM = (C1_PH * C1_PW / 16) / WM, TN = BN / WN / 16;     // a wave: 3 row tiles of 16 pixels x 160 columns
    constexpr int IMG = C1_IROWS * 128, WT = BN * 128;
    constexpr int IK = (C1_IROWS + 63) / 64;                             // DMA instructions per thread slot for one image
    static_assert(C1_IROWS % 8 == 0 && TM * WM * 16 == C1_PH * C1_PW, "tile shape");
    extern __shared__ __attribute__((aligned(128))) char smem[];
    char* img = smem;                   // [2][272][128 B]
    char* wt = smem + 2 * IMG;          // [2][320][128 B]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int frow = lane & 15, fq = lane >> 4;
    // tile id -> (image, patch row, patch column, column tile); column tiles of the same patch are neighbours
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int nt_ = bid % p.ntn;
    bid /= p.ntn;
    const int pxi = bid % p.npx;
    bid /= p.npx;
    const int pyi = bid % p.npy, n = bid / p.npy;
    const int y0 = pyi * C1_PH, x0 = pxi * C1_PW, n0 = nt_ * BN;
    const int C = p.c1 + p.c2, nslices = C >> 6, K = 9 * C;
    const size_t img_row0 = (size_t)n * p.h * p.w_;

    // ---- image staging: slot `tid` fills 16-byte slot tid & 7 of image rows (tid >> 3) + 64 k.  Image row ir = iy * 34 + ix
    // is pixel (y0 + iy - 1, x0 + ix - 1); slot s of it holds data chunk s ^ (ix & 7).
    const f16* zp = (const f16*)g_zero_page;
    const int prow = tid >> 3;
    long long poff[IK];                 // pixel row index (rows of a / a2) of my source, or -1: outside the image -> zero page
    int csw[IK];                        // element offset of the data chunk my slot fetches
#pragma unroll
    for (int k = 0; k < IK; ++k) {
        const int ir = prow + 64 * k;
        const int iy = ir / C1_IW, ix = ir - iy * C1_IW;
        const int y = y0 + iy - 1, x = x0 + ix - 1;
        const bool ok = ir < C1_IROWS && (unsigned)y < (unsigned)p.h && (unsigned)x < (unsigned)p.w_;
        poff[k] = ok ? (long long)(img_row0 + (size_t)y * p.w_ + x) : -1;
        csw[k] = ((tid & 7) ^ (ix & 7)) * 8;
    }
    auto issue_img = [&](int s, int buf) __attribute__((always_inline)) {
        char* dst = img + buf * IMG + wave * 1024;
        const bool first = s * 64 < p.c1;                    // (wave-uniform: the slice lies in source 0 or in source 1)
        const f16* base = first ? p.a + s * 64 : p.a2 + (s * 64 - p.c1);
        const int ld = first ? p.lda : p.lda2;
#pragma unroll
        for (int k = 0; k < IK; ++k) {
            if (k * 64 + wave * 8 < C1_IROWS) {              // (wave-uniform: the last instruction covers a quarter of the waves)
                const f16* src = poff[k] >= 0 ? base + poff[k] * ld + csw[k] : zp;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + k * 8192), 16, 0, 0);
            }
        }
    };
    // ---- weight staging: gemm.hip's Stager (physical LDS row pr holds weight row 8q+4b+j, pr = 16b+4q+j per 32 rows)
    const int wcsw = ((tid & 7) ^ (prow & 7)) * 8;
    const int wrow0 = n0 + ((((prow >> 2) & 3) << 3) | (((prow >> 4) & 1) << 2) | (prow & 3)) + (prow & ~31);
    auto issue_w = [&](int step, int buf) __attribute__((always_inline)) {
        char* dst = wt + buf * WT + wave * 1024;
#pragma unroll
        for (int i = 0; i < BN / 64; ++i) {
            const int r = min(wrow0 + i * 64, p.N - 1);
            __builtin_amdgcn_global_load_lds((gptr_t)(p.w + (size_t)r * K + step * 64 + wcsw), (lptr_t)(dst + i * 8192), 16, 0, 0);
        }
    };
    // ---- normalisation of an image in place: thread = data chunk tid & 7 (8 channels) of image rows (tid >> 3) + 64 k;
    // pixels outside the image stay zero
    const int nch = tid & 7;
    auto normalise = [&](int s, int buf) __attribute__((always_inline)) {
        float a[8], sh[8];
        const float* ab = p.ab + ((size_t)n * C + s * 64 + nch * 8) * 2;
        const f32x4 v0 = *(const f32x4*)ab, v1 = *(const f32x4*)(ab + 4), v2 = *(const f32x4*)(ab + 8), v3 = *(const f32x4*)(ab + 12);
        a[0] = v0[0]; sh[0] = v0[1]; a[1] = v0[2]; sh[1] = v0[3]; a[2] = v1[0]; sh[2] = v1[1]; a[3] = v1[2]; sh[3] = v1[3];
        a[4] = v2[0]; sh[4] = v2[1]; a[5] = v2[2]; sh[5] = v2[3]; a[6] = v3[0]; sh[6] = v3[1]; a[7] = v3[2]; sh[7] = v3[3];
#pragma unroll
        for (int k = 0; k < IK; ++k) {
            if (k * 64 + wave * 8 < C1_IROWS && poff[k] >= 0) {
                const int ir = prow + 64 * k;
                const int ix = ir % C1_IW;
                f16x8* q = (f16x8*)(img + buf * IMG + ir * 128 + ((nch ^ (ix & 7)) << 4));
                const f16x8 v = *q;
                f16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float y = (float)v[j] * a[j] + sh[j];
                    o[j] = (f16)(y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * y)));
                }
                *q = o;
            }
        }
    };
    // ---- activation fragment addressing: row tile rt = 3 wm + i of the patch = patch row rt >> 1, columns 16 (rt & 1) ..;
    // lane frow reads image row (ty + ky) * 34 + 16 (rt & 1) + frow + kx, chunk (4 ks + fq) ^ ((frow + kx) & 7)
    int rbase[TM];                      // byte offset of image row (ty * 34 + 16 (rt & 1) + frow)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int rt = wm * TM + i;
        rbase[i] = ((rt >> 1) * C1_IW + 16 * (rt & 1) + frow) * 128;
    }
    int coff[3][2];                     // [kx][ks]
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) coff[kx][ks] = (((ks * 4 + fq) ^ ((frow + kx) & 7)) << 4) + kx * 128;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    issue_img(0, 0);
    issue_w(0, 0);
    __syncthreads();                      // landed (the barrier's fence waits vmcnt(0))
    normalise(0, 0);
    __syncthreads();
    const int nsteps = nslices * 9;
    for (int step = 0, s = 0, tap = 0; step < nsteps; ++step) {
        const int wcur = step & 1, icur = s & 1;
        if (step + 1 < nsteps) issue_w(step + 1, wcur ^ 1);                  // (buffer last read before the previous barrier)
        if (tap == 0 && s + 1 < nslices) issue_img(s + 1, icur ^ 1);        // normalised during tap 8
        const int ky = tap / 3, kx = tap - ky * 3;
        const char* As = img + icur * IMG + ky * (C1_IW * 128);
        const char* Bs = wt + wcur * WT;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8 af[TM];
            const int co = kx == 0 ? coff[0][ks] : kx == 1 ? coff[1][ks] : coff[2][ks];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *(const f16x8*)(As + rbase[i] + co);
            const int c = ks * 4 + fq;
            constexpr int NG = 2, GS = TN / NG;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                f16x8 bf[GS];
#pragma unroll
                for (int j = 0; j < GS; ++j) {
                    const int row = wn * (BN / WN) + (g * GS + j) * 16 + frow;
                    bf[j] = *(const f16x8*)(Bs + row * 128 + ((c ^ (row & 7)) << 4));
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < GS; ++j)
                        acc[i][g * GS + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][g * GS + j], 0, 0, 0);
            }
        }
        if (tap == 8 && s + 1 < nslices) normalise(s + 1, icur ^ 1);        // its DMA was waited for by the barriers since tap 0
        if (++tap == 9) {
            tap = 0;
            ++s;
        }
        __syncthreads();                  // next weights (and image) landed, this step's buffers fully read
    }

    // ---- epilogue: bias + time-embedding row + residual, 16-byte stores; row (i, frow) of this wave = patch pixel
    // (rt >> 1, 16 (rt & 1) + frow), rt = 3 wm + i
    constexpr int NA = TN / 2;
    const int nb = n0 + wn * (BN / WN);
    const f16* b2row = p.bias2 ? p.bias2 + (img_row0 / p.rpb2) * p.ldb2 : nullptr;      // (an image never straddles two rows of bias2)
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        const int nn = nb + a * 32 + fq * 8;
        const bool col_ok = nn < p.N;
        f16x8 bv = *(const f16x8*)((p.bias && col_ok) ? p.bias + nn : zp);
        const f16x8 b2v = *(const f16x8*)((b2row && col_ok) ? b2row + nn : zp);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int rt = wm * TM + i;
            const int y = y0 + (rt >> 1), x = x0 + 16 * (rt & 1) + frow;
            const bool row_ok = y < p.h && x < p.w_;
            const size_t m = img_row0 + (size_t)(row_ok ? y : 0) * p.w_ + (row_ok ? x : 0);
            const f16x8 rv = *(const f16x8*)((p.res && row_ok && col_ok) ? p.res + m * p.ldr + nn : zp);
            f16x8 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = (f16)(acc[i][2 * a][j] + (float)bv[j] + (float)b2v[j] + (float)rv[j]);
                o[4 + j] = (f16)(acc[i][2 * a + 1][j] + (float)bv[4 + j] + (float)b2v[4 + j] + (float)rv[4 + j]);
            }
            if (row_ok && col_ok) *(f16x8*)(p.out + m * p.ldo + nn) = o;
        }
    }
}

extern "C" int vdx_conv3x3_gn_supported(int c1, int c2, int N) {
    return c1 > 0 && c1 % 64 == 0 && c2 >= 0 && c2 % 64 == 0 && N > 0 && N % 320 == 0 ? 1 : 0;
}

// Is K1 expected to beat the apply pass + conv GEMM?  One column tile (N = 320: the image is staged and normalised once), a
// width the 32-pixel patches tile, and at least two tiles per CU (profiles/r04_k1.md).
extern "C" int vdx_conv3x3_gn_preferred(int c1, int c2, int N, int n_img, int h, int w) {
    if (!vdx_conv3x3_gn_supported(c1, c2, N) || N != 320 || w % C1_PW != 0) return 0;
    const long long tiles = (long long)n_img * ((h + C1_PH - 1) / C1_PH) * (w / C1_PW);
    return tiles >= 2 * vdx_num_cus() ? 1 : 0;
}

extern "C" int vdx_conv3x3_gn_f16(const void* a, int lda, const void* a2, int lda2, int c1, int c2, const float* scale_shift,
                                  const void* w, const void* bias, const void* bias2, int rows_per_bias2, int ldb2,
                                  const void* residual, int ldr, void* out, int ldo, int n_img, int h, int w_px, int N,
                                  vdx_stream_t stream) {
    VDX_CHECK(a && scale_shift && w && out, "conv3x3_gn: null pointer");
    VDX_CHECK(n_img > 0 && h > 0 && w_px > 0, "conv3x3_gn: empty problem");
    VDX_CHECK(vdx_conv3x3_gn_supported(c1, c2, N), "conv3x3_gn: c1=%d c2=%d (%% 64), N=%d (%% 320) not supported", c1, c2, N);
    VDX_CHECK((c2 == 0) == (a2 == nullptr), "conv3x3_gn: a2/c2 mismatch");
    VDX_CHECK(lda % 8 == 0 && lda >= c1 && (!a2 || (lda2 % 8 == 0 && lda2 >= c2)) && ldo % 8 == 0 && ldo >= N &&
                  (!residual || ldr % 8 == 0), "conv3x3_gn: leading dimensions");
    VDX_CHECK(!bias2 || (rows_per_bias2 > 0 && rows_per_bias2 % (h * w_px) == 0 && ldb2 % 8 == 0),
              "conv3x3_gn: bias2 rows must cover whole images (rows_per_bias2=%d, image %dx%d)", rows_per_bias2, h, w_px);
    VDX_CHECK((long long)n_img * h * w_px < (1ll << 31), "conv3x3_gn: too many rows");
    C1P p;
    p.a = (const f16*)a; p.a2 = (const f16*)a2; p.w = (const f16*)w; p.bias = (const f16*)bias; p.bias2 = (const f16*)bias2;
    p.res = (const f16*)residual; p.out = (f16*)out; p.ab = scale_shift;
    p.lda = lda; p.lda2 = lda2; p.ldo = ldo; p.ldr = ldr; p.ldb2 = ldb2; p.rpb2 = rows_per_bias2 > 0 ? rows_per_bias2 : 1;
    p.n_img = n_img; p.h = h; p.w_ = w_px; p.c1 = c1; p.c2 = c2; p.N = N;
    p.npy = (h + C1_PH - 1) / C1_PH; p.npx = (w_px + C1_PW - 1) / C1_PW; p.ntn = N / 320;
    constexpr int lds = 2 * C1_IROWS * 128 + 2 * 320 * 128;
    static const hipError_t attr_rc = hipFuncSetAttribute((const void*)conv3x3_gn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (attr_rc != hipSuccess) return vdx_fail("conv3x3_gn: cannot reserve %d bytes of LDS", lds);
    const long long blocks = (long long)n_img * p.npy * p.npx * p.ntn;
    VDX_CHECK(blocks < (1ll << 31), "conv3x3_gn: grid too large");
    hipLaunchKernelGGL(conv3x3_gn_kernel, dim3((unsigned)blocks), dim3(512), lds, (hipStream_t)stream, p);
    return vdx_launch_status("vdx_conv3x3_gn_f16");
}
