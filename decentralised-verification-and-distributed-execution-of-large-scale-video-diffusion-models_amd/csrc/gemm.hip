// gemm.hip — fp16 GEMM / implicit-GEMM family on the gfx950 matrix cores.
//
//   out[M][N] = epilogue( gather(A)[M][K] . W[N][K]^T )
//
// One kernel template serves every contraction of the 3D-UNet:
//   MODE 0  plain rows (Linear, 1x1 shortcut conv; optional 2-source channel concat)
//   MODE 1  3x3 conv, pad 1, stride 1|2, optional nearest-x2 upsample folded into the gather
//   MODE 2  temporal 3-tap conv (Conv3d (3,1,1)), zero padded at the chunk's first/last frame
// Data layout: activations channels-last rows [pixels][C] so a K-slice of 64 channels of one
// row is one 128-byte line; weights [N][K] with K = tap*C + c.
//
// Tiling: BM x BN x 64 block tile, waves own 64x64 (16 accumulators of v_mfma_f32_16x16x32_f16),
// LDS rows of 128 B XOR-swizzled by (row & 7) so every ds_read_b128 / ds_write_b128 is
// conflict-free, LDS double-buffered with ONE barrier per K-tile, next tile's global loads
// issued before the MFMA block and written to LDS after it (register staging: the gather can
// zero-fill padding taps, which an LDS-DMA load cannot).
// The MFMA is issued with the weight fragment as the A operand, so a lane ends up holding
// 8 consecutive output channels of one row -> 16-byte epilogue stores.
#include "vdx_common.h"

// 128 zero bytes: padding taps / rows past M read from here, so every gather load is unconditional.


struct GemmP {
    const f16 *a, *a2, *w, *bias, *bias2, *res;
    f16* out;
    int M, N, K, c1, c2;
    int lda, lda2, ldo, ldr;
    int h_in, w_in, h_out, w_out, stride, ups;
    int frames, hw, rpb2, ldb2;
    int ntn;
};

template <int BM, int BN, int WM, int WN, int MODE, bool GEGLU>
__global__ __launch_bounds__(WM* WN * 64) void gemm_kernel(const GemmP p) {
    constexpr int NT = WM * WN * 64;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int ACH = BM * 8 / NT, BCH = BN * 8 / NT;
    constexpr int STAGE = (BM + BN) * 128;
    static_assert(TN % 2 == 0 && ACH >= 1 && BCH >= 1, "tile shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (bid / p.ntn) * BM, n0 = (bid % p.ntn) * BN;

    // ---- per-thread gather descriptors (fixed across K tiles) ------------------------------
    const int cchunk = tid & 7;           // 16-byte chunk inside the 128-byte K-slice
    size_t a_off[ACH];                    // MODE 0: row offset in source 0;  MODE 1/2: see below
    size_t a_off2[ACH];                   // MODE 0: row offset in source 1
    int a_y[ACH], a_x[ACH];               // MODE 1: top-left tap coords;  MODE 2: a_y = frame idx
    bool a_ok[ACH];
    int a_lds[ACH];
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
        const int r = (i * NT + tid) >> 3;
        const int m = m0 + r;
        a_ok[i] = m < p.M;
        const int mm = a_ok[i] ? m : 0;
        a_lds[i] = r * 128 + ((cchunk ^ (r & 7)) << 4);
        if (MODE == 0) {
            a_off[i] = (size_t)mm * p.lda;
            a_off2[i] = (size_t)mm * p.lda2;
            a_y[i] = a_x[i] = 0;
        } else if (MODE == 1) {
            const int per = p.h_out * p.w_out;
            const int n = mm / per, rem = mm - n * per;
            const int yo = rem / p.w_out, xo = rem - yo * p.w_out;
            a_y[i] = yo * p.stride - 1;
            a_x[i] = xo * p.stride - 1;
            a_off[i] = (size_t)n * p.h_in * p.w_in;   // first source row of this image
            a_off2[i] = 0;
        } else {
            a_y[i] = (mm / p.hw) % p.frames;
            a_x[i] = 0;
            a_off[i] = (size_t)mm;
            a_off2[i] = 0;
        }
    }
    size_t b_off[BCH];
    int b_lds[BCH];
#pragma unroll
    for (int i = 0; i < BCH; ++i) {
        const int r = (i * NT + tid) >> 3;
        // LDS row permutation inside each 32-row group: global row 8q+4b+j -> LDS row 16b+4q+j,
        // so that after the MFMA a lane owns 8 consecutive output columns.
        const int pr = (r & ~31) | (((r >> 2) & 1) << 4) | (((r >> 3) & 3) << 2) | (r & 3);
        b_lds[i] = BM * 128 + pr * 128 + ((cchunk ^ (pr & 7)) << 4);
        b_off[i] = (size_t)(n0 + r) * p.K + cchunk * 8;
    }

    // ---- K-tile walker ---------------------------------------------------------------------
    const int ct = p.c1 + p.c2;           // channels per tap
    int tap = 0, kc = 0;                  // state of the NEXT tile to be loaded
    u32x4 ra[ACH], rb[BCH];
    const f16* zp = (const f16*)g_zero_page;
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < BCH; ++i) rb[i] = *(const u32x4*)(p.w + b_off[i] + (size_t)kt * 64);
        if (MODE == 0) {
            const bool first = kc < p.c1;
#pragma unroll
            for (int i = 0; i < ACH; ++i) {
                const f16* src = first ? p.a + a_off[i] + kc : p.a2 + a_off2[i] + (kc - p.c1);
                ra[i] = *(const u32x4*)(a_ok[i] ? src + cchunk * 8 : zp);
            }
        } else if (MODE == 1) {
            const int ky = tap / 3, kx = tap - ky * 3;
            const int hlim = p.h_in << p.ups, wlim = p.w_in << p.ups;
#pragma unroll
            for (int i = 0; i < ACH; ++i) {
                int y = a_y[i] + ky, x = a_x[i] + kx;
                const bool ok = a_ok[i] && (unsigned)y < (unsigned)hlim && (unsigned)x < (unsigned)wlim;
                y >>= p.ups;
                x >>= p.ups;
                const f16* src = p.a + (a_off[i] + (size_t)(y * p.w_in + x)) * p.lda + kc + cchunk * 8;
                ra[i] = *(const u32x4*)(ok ? src : zp);
            }
        } else {
#pragma unroll
            for (int i = 0; i < ACH; ++i) {
                const int f = a_y[i] + tap - 1;
                const bool ok = a_ok[i] && (unsigned)f < (unsigned)p.frames;
                const f16* src = p.a + (a_off[i] + (size_t)((tap - 1) * p.hw)) * p.lda + kc + cchunk * 8;
                ra[i] = *(const u32x4*)(ok ? src : zp);
            }
        }
        kc += 64;
        if (kc == ct) { kc = 0; ++tap; }
    };
    auto lstore = [&](int buf) {
        char* s = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < ACH; ++i) *(u32x4*)(s + a_lds[i]) = ra[i];
#pragma unroll
        for (int i = 0; i < BCH; ++i) *(u32x4*)(s + b_lds[i]) = rb[i];
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = p.K >> 6;
    const int frow = lane & 15, fq = lane >> 4;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
        const char* As = smem + cur * STAGE;
        const char* Bs = As + BM * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8 af[TM], bf[TN];
            const int c = ks * 4 + fq;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * WTM + i * 16 + frow;
                af[i] = *(const f16x8*)(As + row * 128 + ((c ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn * WTN + j * 16 + frow;
                bf[j] = *(const f16x8*)(Bs + row * 128 + ((c ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) lstore(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds rows m = ..+frow, 8 consecutive columns per accumulator pair --
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * WTM + i * 16 + frow;
        if (m >= p.M) continue;
        const f16* b2row = p.bias2 ? p.bias2 + (size_t)(m / p.rpb2) * p.ldb2 : nullptr;
#pragma unroll
        for (int a = 0; a < TN / 2; ++a) {
            const int n = n0 + wn * WTN + a * 32 + fq * 8;
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

// ---- host side ------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int MODE, bool GEGLU>
static int launch(const GemmP& p, hipStream_t st) {
    constexpr int lds = 2 * (BM + BN) * 128;
    auto kern = gemm_kernel<BM, BN, WM, WN, MODE, GEGLU>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return vdx_fail("gemm: cannot reserve %d bytes of LDS", lds);
        attr_set = true;
    }
    GemmP q = p;
    q.ntn = p.N / BN;
    const int ntm = (p.M + BM - 1) / BM;
    hipLaunchKernelGGL(kern, dim3(ntm * q.ntn), dim3(WM * WN * 64), lds, st, q);
    return vdx_launch_status("vdx_gemm_f16");
}

template <int MODE, bool GEGLU>
static int pick_tile(const GemmP& p, hipStream_t st) {
    if (p.N % 128 == 0) return launch<128, 128, 2, 2, MODE, GEGLU>(p, st);
    return launch<256, 64, 4, 1, MODE, GEGLU>(p, st);
}

extern "C" int vdx_gemm_f16(const vdx_gemm_args* a, vdx_stream_t stream) {
    VDX_CHECK(a && a->a && a->w && a->out, "gemm: null pointer");
    VDX_CHECK(a->M > 0 && a->N > 0 && a->K > 0, "gemm: empty problem M=%d N=%d K=%d", a->M, a->N, a->K);
    VDX_CHECK(a->N % 64 == 0 && a->K % 64 == 0, "gemm: N=%d and K=%d must be multiples of 64", a->N, a->K);
    VDX_CHECK(a->c1 > 0 && a->c1 % 64 == 0 && a->c2 % 64 == 0, "gemm: c1=%d c2=%d must be multiples of 64", a->c1, a->c2);
    VDX_CHECK((a->c2 == 0) == (a->a2 == nullptr), "gemm: a2/c2 mismatch");
    const int taps = a->mode == VDX_GEMM_CONV3X3 ? 9 : a->mode == VDX_GEMM_TCONV3 ? 3 : 1;
    VDX_CHECK(a->K == taps * (a->c1 + a->c2), "gemm: K=%d != taps*(c1+c2)=%d", a->K, taps * (a->c1 + a->c2));
    VDX_CHECK(a->lda % 8 == 0 && a->ldo % 8 == 0 && (a->a2 == nullptr || a->lda2 % 8 == 0) &&
                  (a->residual == nullptr || a->ldr % 8 == 0),
              "gemm: leading dimensions must be multiples of 8 elements");
    VDX_CHECK(a->lda >= a->c1, "gemm: lda < c1");
    GemmP p;
    p.a = (const f16*)a->a; p.a2 = (const f16*)a->a2; p.w = (const f16*)a->w;
    p.bias = (const f16*)a->bias; p.bias2 = (const f16*)a->bias2; p.res = (const f16*)a->residual;
    p.out = (f16*)a->out;
    p.M = a->M; p.N = a->N; p.K = a->K; p.c1 = a->c1; p.c2 = a->c2;
    p.lda = a->lda; p.lda2 = a->lda2; p.ldo = a->ldo; p.ldr = a->ldr;
    p.h_in = a->h_in; p.w_in = a->w_in; p.h_out = a->h_out; p.w_out = a->w_out;
    p.stride = a->stride; p.ups = a->upsample ? 1 : 0;
    p.frames = a->frames; p.hw = a->hw; p.rpb2 = a->rows_per_bias2 > 0 ? a->rows_per_bias2 : 1;
    p.ldb2 = a->ldb2 > 0 ? a->ldb2 : a->N;
    VDX_CHECK(p.ldb2 % 8 == 0, "gemm: ldb2 must be a multiple of 8");
    p.ntn = 0;
    hipStream_t st = (hipStream_t)stream;
    const bool geglu = (a->epilogue & VDX_EPI_GEGLU) != 0;
    if (geglu) {
        VDX_CHECK(a->mode == VDX_GEMM_PLAIN && !a->bias2 && !a->residual, "gemm: GEGLU epilogue is plain-mode only");
        VDX_CHECK(a->ldo % 4 == 0, "gemm: GEGLU ldo must be a multiple of 4");
        return pick_tile<0, true>(p, st);
    }
    switch (a->mode) {
        case VDX_GEMM_PLAIN:
            return pick_tile<0, false>(p, st);
        case VDX_GEMM_CONV3X3:
            VDX_CHECK(a->c2 == 0, "gemm: conv3x3 takes one source");
            VDX_CHECK(a->stride == 1 || a->stride == 2, "gemm: stride %d", a->stride);
            VDX_CHECK(a->h_in > 0 && a->w_in > 0 && a->h_out > 0 && a->w_out > 0, "gemm: conv geometry");
            VDX_CHECK(a->M % (a->h_out * a->w_out) == 0, "gemm: M=%d not a whole number of %dx%d images", a->M, a->h_out, a->w_out);
            {
                const int he = a->h_in << p.ups, we = a->w_in << p.ups;
                VDX_CHECK(a->h_out == (he + 2 - 3) / a->stride + 1 && a->w_out == (we + 2 - 3) / a->stride + 1,
                          "gemm: conv output %dx%d inconsistent with input %dx%d stride %d", a->h_out, a->w_out, he, we, a->stride);
            }
            return pick_tile<1, false>(p, st);
        case VDX_GEMM_TCONV3:
            VDX_CHECK(a->c2 == 0, "gemm: tconv3 takes one source");
            VDX_CHECK(a->frames > 0 && a->hw > 0 && a->M % (a->frames * a->hw) == 0, "gemm: tconv geometry M=%d F=%d HW=%d", a->M, a->frames, a->hw);
            return pick_tile<2, false>(p, st);
    }
    return vdx_fail("gemm: unknown mode %d", a->mode);
}
