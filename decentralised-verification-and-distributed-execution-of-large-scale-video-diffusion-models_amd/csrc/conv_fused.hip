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
// LDS (once per element, not once per tap) in five 64-row pieces during taps 4..8 of the slice before, and the nine taps
// read their activation fragments from that one image at row offsets (ky * 34 + kx).  A fragment's 16 lanes read 16
// consecutive image columns; the 16-byte chunk swizzle is the image COLUMN's low three bits, so a tap's swizzle depends on kx
// alone and a fragment address is one add on precomputed lane offsets.  Weights per (slice, tap) as [320][64] tiles,
// double-buffered, in gemm.hip's layout / swizzle / row permutation (the same packed weights: K = (c / 64) * 576 + tap * 64 +
// c % 64); accumulators, weight fragments and the 16-byte epilogue stores are gemm.hip's.
// LDS: 2 x 34 KB (image) + 2 x 40 KB (weights) + 8 bytes per input channel (scale, shift) <= 160 KB, one 512-thread
// block per CU.  Measurements, the three versions of this kernel and what the in-LDS normalisation costs: profiles/r04_k1.md.
// Diagnostic switches (tools/k1_abl.sh builds them into csrc/build/abl/libk1_<tag>.so; the product defines none):
//   K1_ABL_NONORM  timing-only, WRONG RESULTS: later slices are not normalised;   K1_ABL_NOPIN  hipcc's own instruction order.
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
    constexpr int BN = 320, WM = 2, WN = 4;
    // a wave: 6 row tiles of 16 pixels (three patch rows) x 80 columns — (96 + 80) x 128 bytes of fragment reads per K = 64 for
    // 30 MFMAs; 48 x 160 reads 18 % more, and the LDS port is what this kernel runs against (profiles/r04_k1.md)
    constexpr int TM = (C1_PH * C1_PW / 16) / WM, TN = BN / WN / 16;
    constexpr int IMG = C1_IROWS * 128, WT = BN * 128;
    constexpr int IK = (C1_IROWS + 63) / 64;                             // DMA instructions per thread slot for one image
    static_assert(C1_IROWS % 8 == 0 && TM * WM * 16 == C1_PH * C1_PW, "tile shape");
    extern __shared__ __attribute__((aligned(128))) char smem[];
    char* img = smem;                   // [2][272][128 B]
    char* wt = smem + 2 * IMG;          // [2][320][128 B]
    float* abl = (float*)(wt + 2 * WT); // [C][2]: (scale, shift) of this block's image

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
    // (source row and chunk are recomputed per slice — a few dozen integer instructions once in nine K steps — rather than
    // kept: fifteen registers the K loop has better uses for)
    auto src_of = [&](int k, int& row, int& csw) __attribute__((always_inline)) {
        const int ir = prow + 64 * k;
        const int iy = ir / C1_IW, ix = ir - iy * C1_IW;
        const int y = y0 + iy - 1, x = x0 + ix - 1;
        const bool ok = ir < C1_IROWS && (unsigned)y < (unsigned)p.h && (unsigned)x < (unsigned)p.w_;
        row = ok ? (int)img_row0 + y * p.w_ + x : -1;       // (rows < 2^31: checked on the host)
        csw = ((tid & 7) ^ (ix & 7)) * 8;
        return ix;
    };
    auto issue_img = [&](int s, int buf) __attribute__((always_inline)) {
        char* dst = img + buf * IMG + wave * 1024;
        const bool first = s * 64 < p.c1;                    // (wave-uniform: the slice lies in source 0 or in source 1)
        const f16* base = first ? p.a + s * 64 : p.a2 + (s * 64 - p.c1);
        const int ld = first ? p.lda : p.lda2;
#pragma unroll
        for (int k = 0; k < IK; ++k) {
            if (k * 64 + wave * 8 < C1_IROWS) {              // (wave-uniform: the last instruction covers a quarter of the waves)
                int row, csw;
                src_of(k, row, csw);
                const f16* src = row >= 0 ? base + (size_t)row * ld + csw : zp;
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
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    // silu(x * a + sh) of my eight channels (four pairs; v_pk_fma / v_pk_mul / v_pk_add_f32 + 2 x (v_exp_f32, v_rcp_f32) per pair)
    auto silu8 = [&](const f16x8 v, const f32x4 (&t)[4]) __attribute__((always_inline)) {
        f16x8 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x2 x = {(float)v[2 * q], (float)v[2 * q + 1]};
            const f32x2 y = x * (f32x2){t[q][0], t[q][1]} + (f32x2){t[q][2], t[q][3]};
            const f32x2 u = y * -1.44269504088896341f;
            const f32x2 d = (f32x2){__builtin_amdgcn_exp2f(u[0]), __builtin_amdgcn_exp2f(u[1])} + 1.0f;
            const f32x2 r = y * (f32x2){__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
            o[2 * q] = (f16)r[0];
            o[2 * q + 1] = (f16)r[1];
        }
        return o;
    };
    auto normalise = [&](int s, int buf) __attribute__((always_inline)) {
        const float* ab = abl + (s * 64 + nch * 8) * 2;
        const f32x4 t[4] = {*(const f32x4*)ab, *(const f32x4*)(ab + 4), *(const f32x4*)(ab + 8), *(const f32x4*)(ab + 12)};
#pragma unroll
        for (int k = 0; k < IK; ++k) {
            int row, csw;
            const int ix = src_of(k, row, csw);
            if (k * 64 + wave * 8 < C1_IROWS && row >= 0) {
                const int ir = prow + 64 * k;
                f16x8* q = (f16x8*)(img + buf * IMG + ir * 128 + ((nch ^ (ix & 7)) << 4));
                *q = silu8(*q, t);
            }
        }
    };
    // ---- activation fragment addressing: row tile rt = 6 wm + i of the patch = patch row 3 wm + (i >> 1), columns
    // 16 (i & 1) ..; lane frow reads image row (3 wm + (i >> 1) + ky) * 34 + 16 (i & 1) + frow + kx, chunk
    // (4 ks + fq) ^ ((frow + kx) & 7): one base per lane, the tile's offset is a constant
    const int rbase = (wm * 3 * C1_IW + frow) * 128;
    auto roff = [](int i) constexpr { return ((i >> 1) * C1_IW + 16 * (i & 1)) * 128; };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // (scale, shift) of the image's channels, per channel PAIR as (a0, a1, sh0, sh1): operands of the packed fp32 instructions
    for (int i = tid * 4; i < 2 * C; i += 2048) {
        const f32x4 t = *(const f32x4*)(p.ab + (size_t)n * C * 2 + i);
        *(f32x4*)(abl + i) = (f32x4){t[0], t[2], t[1], t[3]};
    }
    issue_img(0, 0);
    issue_w(0, 0);
    __syncthreads();                      // landed (the barrier's fence waits vmcnt(0))
    normalise(0, 0);
    __syncthreads();

    // ---- one K step (tap, 64 channels) = four groups of 15 MFMAs: (ks, g) = (K half, row tiles 3 g .. 3 g + 2) against the
    // wave's five weight fragments of that K half.  The fragment reads of a group are issued among the MFMAs of the group
    // before it (two register sets, the order pinned by sched_group_barrier: hipcc left to itself drains the LDS queue in
    // front of every group), only the first group of a step waits for reads issued after the barrier.
    const int browb = (wn * (BN / WN) + frow) * 128, bsw = frow & 7;
    auto rdB = [&](f16x8 (&bf)[TN], const char* Bs, int ks) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *(const f16x8*)(Bs + browb + j * 2048 + (((ks * 4 + fq) ^ bsw) << 4));
    };
    auto rdA = [&](f16x8 (&af)[3], const char* As, int co, int g) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 3; ++i) af[i] = *(const f16x8*)(As + rbase + roff(g * 3 + i) + co);
    };
    auto mm = [&](int g, const f16x8 (&af)[3], const f16x8 (&bf)[TN]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[g * 3 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[g * 3 + i][j], 0, 0, 0);
    };
#ifdef K1_ABL_NOPIN                        // timing-only build: hipcc's own order of the reads and MFMAs
#define K1_PIN(mask, n)
#else
#define K1_PIN(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#endif
    const int nsteps = nslices * 9;
    for (int step = 0, s = 0, tap = 0; step < nsteps; ++step) {
        const int wcur = step & 1, icur = s & 1;
        if (step + 1 < nsteps) issue_w(step + 1, wcur ^ 1);                  // (buffer last read before the previous barrier)
        if (tap == 0 && s + 1 < nslices) issue_img(s + 1, icur ^ 1);        // landed at the next barrier, normalised in taps 4..8
        const int ky = tap / 3, kx = tap - ky * 3;
        const char* As = img + icur * IMG + ky * (C1_IW * 128);
        const char* Bs = wt + wcur * WT;
        const int co0 = ((fq ^ ((frow + kx) & 7)) << 4) + kx * 128, co1 = (((4 + fq) ^ ((frow + kx) & 7)) << 4) + kx * 128;
        // the next slice's image is normalised in five pieces of 64 image rows, one in the last group of each of taps 4..8
        // (vector work in the shadow of that group's MFMAs); (wave-uniform) the last piece covers rows 256..271: waves 0, 1
        const int nir = prow + 64 * (tap - 4);
#ifdef K1_ABL_NONORM                       // timing-only build (tools/k1_abl.sh): later slices are not normalised — wrong results
        const bool nwork = false;
#else
        const bool nwork = tap >= 4 && s + 1 < nslices && wave * 8 + 64 * (tap - 4) < C1_IROWS;
#endif
        f16x8 bf0[TN], bf1[TN], afA[3], afB[3];
        rdB(bf0, Bs, 0);
        rdA(afA, As, co0, 0);
        mm(0, afA, bf0);
        rdA(afB, As, co0, 1);
        mm(1, afB, bf0);
        rdB(bf1, Bs, 1);
        rdA(afA, As, co1, 0);
        mm(0, afA, bf1);
        rdA(afB, As, co1, 1);
        K1_PIN(0x100, 8);
#pragma unroll
        for (int q = 0; q < 3; ++q) {         // group 0: 15 MFMAs, 3 reads
            K1_PIN(0x008, 5);
            K1_PIN(0x100, 1);
        }
#pragma unroll
        for (int q = 0; q < 7; ++q) {         // group 1: 15 MFMAs, 8 reads
            K1_PIN(0x008, 2);
            K1_PIN(0x100, 1);
        }
        K1_PIN(0x008, 1);
        K1_PIN(0x100, 1);
#pragma unroll
        for (int q = 0; q < 3; ++q) {         // group 2: 15 MFMAs, 3 reads
            K1_PIN(0x008, 5);
            K1_PIN(0x100, 1);
        }
        if (nwork) {
            const int niy = nir / C1_IW, nix = nir - niy * C1_IW;
            const int ny = y0 + niy - 1, nx = x0 + nix - 1;
            const bool ninside = (unsigned)ny < (unsigned)p.h && (unsigned)nx < (unsigned)p.w_;
            f16x8* nq = (f16x8*)(img + (icur ^ 1) * IMG + nir * 128 + ((nch ^ (nix & 7)) << 4));
            const float* ab = abl + ((s + 1) * 64 + nch * 8) * 2;
            const f16x8 nv = *nq;
            const f32x4 t[4] = {*(const f32x4*)ab, *(const f32x4*)(ab + 4), *(const f32x4*)(ab + 8), *(const f32x4*)(ab + 12)};
            mm(1, afB, bf1);
            const f16x8 o = silu8(nv, t);
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const unsigned keep = ninside ? 0xffffffffu : 0u;          // pixels outside the image stay zero
            u32x4 ob = __builtin_bit_cast(u32x4, o);
            ob &= (u32x4){keep, keep, keep, keep};
            *nq = __builtin_bit_cast(f16x8, ob);
            K1_PIN(0x100, 5);     // the piece and its scale / shift
            K1_PIN(0x008, 3);
#pragma unroll
            for (int q = 0; q < 8; ++q) {     // group 3: 15 MFMAs among ~36 vector instructions + 16 transcendentals
                K1_PIN(0x002, 3);
                K1_PIN(0x400, 2);
                K1_PIN(0x008, 1);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                K1_PIN(0x002, 3);
                K1_PIN(0x008, 1);
            }
        } else {
            mm(1, afB, bf1);
        }
        if (++tap == 9) {
            tap = 0;
            ++s;
        }
        __syncthreads();                  // next weights (and image) landed, this step's buffers fully read
    }

    // ---- epilogue: bias + time-embedding row + residual.  Row (i, frow) of this wave = patch pixel (3 wm + (i >> 1),
    // 16 (i & 1) + frow).  Column tile t = 5 wn + j of the block holds, per lane, columns 32 (t >> 1) + 8 fq + 4 (t & 1) + 0..3
    // (gemm.hip's weight row permutation): an even tile and its successor make 8 consecutive columns = one 16-byte store; a
    // wave's five tiles are two such pairs and one single (8-byte stores), which one depends on the parity of wn.
    const f16* b2row = p.bias2 ? p.bias2 + (img_row0 / p.rpb2) * p.ldb2 : nullptr;      // (an image never straddles two rows of bias2)
    auto row_of = [&](int i, bool& ok) __attribute__((always_inline)) {
        const int y = y0 + wm * 3 + (i >> 1), x = x0 + 16 * (i & 1) + frow;
        ok = y < p.h && x < p.w_;
        return img_row0 + (size_t)(ok ? y : 0) * p.w_ + (ok ? x : 0);
    };
    auto pair = [&](int j0) __attribute__((always_inline)) {            // tiles j0 (even t), j0 + 1
        const int nn = n0 + ((wn * TN + j0) >> 1) * 32 + fq * 8;
        const bool col_ok = nn < p.N;
        const f16x8 bv = *(const f16x8*)((p.bias && col_ok) ? p.bias + nn : zp);
        const f16x8 b2v = *(const f16x8*)((b2row && col_ok) ? b2row + nn : zp);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            bool row_ok;
            const size_t m = row_of(i, row_ok);
            const f16x8 rv = *(const f16x8*)((p.res && row_ok && col_ok) ? p.res + m * p.ldr + nn : zp);
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = (f16)(acc[i][j0][e] + (float)bv[e] + (float)b2v[e] + (float)rv[e]);
                o[4 + e] = (f16)(acc[i][j0 + 1][e] + (float)bv[4 + e] + (float)b2v[4 + e] + (float)rv[4 + e]);
            }
            if (row_ok && col_ok) *(f16x8*)(p.out + m * p.ldo + nn) = o;
        }
    };
    auto single = [&](int j) __attribute__((always_inline)) {
        const int t = wn * TN + j;
        const int nn = n0 + (t >> 1) * 32 + fq * 8 + (t & 1) * 4;
        const bool col_ok = nn < p.N;
        const f16x4 bv = *(const f16x4*)((p.bias && col_ok) ? p.bias + nn : zp);
        const f16x4 b2v = *(const f16x4*)((b2row && col_ok) ? b2row + nn : zp);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            bool row_ok;
            const size_t m = row_of(i, row_ok);
            const f16x4 rv = *(const f16x4*)((p.res && row_ok && col_ok) ? p.res + m * p.ldr + nn : zp);
            f16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (f16)(acc[i][j][e] + (float)bv[e] + (float)b2v[e] + (float)rv[e]);
            if (row_ok && col_ok) *(f16x4*)(p.out + m * p.ldo + nn) = o;
        }
    };
    static_assert(TN == 5, "the pairing below is written for five column tiles per wave");
    if (wn & 1) {          // tiles t = 5, 6..9 or 15, 16..19: the first is the odd half of a pair owned by the wave before
        single(0);
        pair(1);
        pair(3);
    } else {
        pair(0);
        pair(2);
        single(4);
    }
}

extern "C" int vdx_conv3x3_gn_supported(int c1, int c2, int N) {
    // (c1 + c2 <= 1536: the block keeps its image's scale / shift table in LDS beside the tiles)
    return c1 > 0 && c1 % 64 == 0 && c2 >= 0 && c2 % 64 == 0 && c1 + c2 <= 1536 && N > 0 && N % 320 == 0 ? 1 : 0;
}

// Is K1 expected to beat the apply pass + conv GEMM?  One column tile (N = 320: the image is staged and normalised once), a
// width the 32-pixel patches tile, and at least two tiles per CU (profiles/r04_k1.md).
extern "C" int vdx_conv3x3_gn_preferred(int c1, int c2, int N, int n_img, int h, int w) {
    if (!vdx_conv3x3_gn_supported(c1, c2, N) || N != 320 || w % C1_PW != 0) return 0;
    // The choice is a function of ONE image's shape (not of n_img, not of the chip): K1 and the apply pass + conv GEMM differ
    // by a rounding of the normalised value, and a sample's bits must not depend on the batch or window it is computed in
    // (ADVICE r4).  48 tiles per image at level 0 of the XL UNet; n_img is kept in the signature for the C-ABI only.
    (void)n_img;
    const long long tiles_per_image = (long long)((h + C1_PH - 1) / C1_PH) * (w / C1_PW);
    return tiles_per_image >= 32 ? 1 : 0;
}

extern "C" int vdx_conv3x3_gn_f16(const void* a, int lda, const void* a2, int lda2, int c1, int c2, const float* scale_shift,
                                  const void* w, const void* bias, const void* bias2, int rows_per_bias2, int ldb2,
                                  const void* residual, int ldr, void* out, int ldo, int n_img, int h, int w_px, int N,
                                  vdx_stream_t stream) {
    VDX_CHECK(a && scale_shift && w && out, "conv3x3_gn: null pointer");
    VDX_CHECK((uintptr_t)scale_shift % 16 == 0, "conv3x3_gn: scale_shift must be 16-byte aligned (it is read with 16-byte loads)");
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
    const int lds = 2 * C1_IROWS * 128 + 2 * 320 * 128 + (c1 + c2) * 8;
    static const hipError_t attr_rc = hipFuncSetAttribute((const void*)conv3x3_gn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    if (attr_rc != hipSuccess) return vdx_fail("conv3x3_gn: cannot reserve 160 KB of LDS");
    const long long blocks = (long long)n_img * p.npy * p.npx * p.ntn;
    VDX_CHECK(blocks < (1ll << 31), "conv3x3_gn: grid too large");
    hipLaunchKernelGGL(conv3x3_gn_kernel, dim3((unsigned)blocks), dim3(512), lds, (hipStream_t)stream, p);
    return vdx_launch_status("vdx_conv3x3_gn_f16");
}

// Lab variants of this translation unit (phase stamps, ablations: timing only, some give WRONG results) are compiled in only
// under the macros below; a library that carries one says so through vdx_build_flags() and vdx/_lib.py refuses to load it
// as the product.
extern "C" int vdx_lab_conv_fused(void) {
#if defined(K1_ABL_NOPIN) || defined(K1_ABL_NONORM)
    return 64;
#else
    return 0;
#endif
}
