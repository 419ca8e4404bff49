// tconv_fused.hip — K3 (SURVEY.md §2.3, row TemporalConvLayer; App. A.4): the temporal 3-tap convolution with the
// GroupNorm apply and the SiLU of its Sequential FUSED IN,
//
//     out[b, f, p, :] = bias + residual + sum_{kt = 0..2} W_kt . silu( x[b, f + kt - 1, p, :] * a[b, :] + s[b, :] )      (zero rows
//     for frames outside the clip — the padding applies to the NORMALISED tensor),
//
// reached four times per TemporalConvLayer from fsdp_chunked_coherent.py:140.  (a, s) = the per-(sample, channel) scale
// and shift the statistics pass leaves behind (norm.hip: gn_partial + gn_finalize, unchanged).  What it replaces: the
// apply pass (one read + one write of the activation, 88 launches per forward) followed by gemm_kernel<.., MODE 2, ..>,
// which stages the normalised rows three times (once per tap).
//
// Tile = 16 pixels x FT frames of ONE batch item x 320 output channels.  Rows are ordered (frame, pixel), so the frame
// shift of a tap is a shift by whole 16-row MFMA tiles: per 64-channel slice the block stages ONE image of (FT + 2)
// frames x 16 pixels x 64 channels by LDS-DMA (frames -1 and FT: the neighbours' frames, or the zero page at the ends of
// the clip), normalises it IN PLACE in LDS (x * a + s, SiLU; once per element, not once per tap) while the previous
// slice's last tap runs on the matrix cores, and the three taps read their activation fragments from that one image at
// row-tile offsets 0 / 1 / 2.  Weights stream per (slice, tap) as [320][64] tiles, double-buffered, in the layout,
// swizzle and row permutation of gemm.hip (the same packed weights: K = (c / 64) * 192 + kt * 64 + c % 64), and the
// accumulators / fragment addressing / epilogue store format are gemm.hip's (16x16x32 MFMA, weight fragment as the A
// operand, 8 consecutive output channels per lane).
//
// LDS at FT = 16: 2 x 36 KB (image) + 2 x 40 KB (weights) = 152 KB, one 512-thread block per CU.
#include "gemm_common.h"

struct TcP {
    const f16 *x, *w, *bias, *res;
    f16* out;
    const float* ab;          // [B][C][2]: scale, shift
    int ldx, ldo, ldr;
    int B, F, S, C, N;
    int nfc, npb, ntn;        // frame chunks per clip (F / FT), pixel blocks (ceil(S / 16)), column tiles (N / 320)
};

template <int FT>
__global__ __launch_bounds__(512) void tconv_gn_kernel(const TcP p) {
    constexpr int BN = 320, WM = 4, WN = 2;
    constexpr int TM = FT / WM, TN = BN / WN / 16;           // a wave: TM frames (row tiles of 16 pixels) x 160 columns
    constexpr int IROWS = (FT + 2) * 16, IMG = IROWS * 128, WT = BN * 128;
    constexpr int IK = (IROWS + 63) / 64;                    // DMA instructions per thread slot for one image
    static_assert(FT % WM == 0 && TN % 2 == 0 && IROWS % 32 == 0, "tile shape");
    extern __shared__ __attribute__((aligned(128))) char smem[];
    char* img = smem;                   // [2][IROWS][128 B]
    char* wt = smem + 2 * IMG;          // [2][320][128 B]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int frow = lane & 15, fq = lane >> 4;
    // tile id -> (batch item, frame chunk, pixel block, column tile); column tiles of the same rows are neighbours
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int nt_ = bid % p.ntn;
    bid /= p.ntn;
    const int pb = bid % p.npb;
    bid /= p.npb;
    const int fc = bid % p.nfc, b = bid / p.nfc;
    const int f0 = fc * FT, p0 = pb * 16, n0 = nt_ * BN;
    const int nslices = p.C >> 6, K = 3 * p.C;

    // ---- image staging: slot `tid` fills 16-byte slot tid & 7 of image rows (tid >> 3) + 64 k; chunk swizzle = row & 7
    const f16* zp = (const f16*)g_zero_page;
    const int prow = tid >> 3;
    const int csw = ((tid & 7) ^ (prow & 7)) * 8;            // (rows 64 k apart share row & 7)
    long long xoff[IK];                                      // element offset of my source row, or -1: zero page
#pragma unroll
    for (int k = 0; k < IK; ++k) {
        const int r = prow + 64 * k;
        const int f = f0 + (r >> 4) - 1, px = min(p0 + (r & 15), p.S - 1);
        xoff[k] = (r < IROWS && (unsigned)f < (unsigned)p.F) ? ((long long)(b * p.F + f) * p.S + px) * p.ldx + csw : -1;
    }
    auto issue_img = [&](int s, int buf) __attribute__((always_inline)) {
        char* dst = img + buf * IMG + wave * 1024;
#pragma unroll
        for (int k = 0; k < IK; ++k) {
            if (k * 64 + wave * 8 < IROWS) {                 // (wave-uniform: the last instruction covers half the waves)
                const f16* src = xoff[k] >= 0 ? p.x + xoff[k] + s * 64 : zp;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + k * 8192), 16, 0, 0);
            }
        }
    };
    // ---- weight staging: gemm.hip's Stager (physical LDS row pr holds weight row 8q+4b+j, pr = 16b+4q+j per 32 rows)
    const int wrow0 = n0 + ((((prow >> 2) & 3) << 3) | (((prow >> 4) & 1) << 2) | (prow & 3)) + (prow & ~31);
    auto issue_w = [&](int step, int buf) __attribute__((always_inline)) {
        char* dst = wt + buf * WT + wave * 1024;
#pragma unroll
        for (int i = 0; i < BN / 64; ++i) {
            const int r = min(wrow0 + i * 64, p.N - 1);
            __builtin_amdgcn_global_load_lds((gptr_t)(p.w + (size_t)r * K + step * 64 + csw), (lptr_t)(dst + i * 8192), 16, 0, 0);
        }
    };
    // ---- normalisation of an image in place: thread = data chunk tid & 7 (8 channels) of rows (tid >> 3) + 64 k
    const int nch = tid & 7;
    auto normalise = [&](int s, int buf) __attribute__((always_inline)) {
        float a[8], sh[8];
        const float* ab = p.ab + ((size_t)b * p.C + s * 64 + nch * 8) * 2;
        const f32x4 v0 = *(const f32x4*)ab, v1 = *(const f32x4*)(ab + 4), v2 = *(const f32x4*)(ab + 8), v3 = *(const f32x4*)(ab + 12);
        a[0] = v0[0]; sh[0] = v0[1]; a[1] = v0[2]; sh[1] = v0[3]; a[2] = v1[0]; sh[2] = v1[1]; a[3] = v1[2]; sh[3] = v1[3];
        a[4] = v2[0]; sh[4] = v2[1]; a[5] = v2[2]; sh[5] = v2[3]; a[6] = v3[0]; sh[6] = v3[1]; a[7] = v3[2]; sh[7] = v3[3];
#pragma unroll
        for (int k = 0; k < IK; ++k) {
            const int r = prow + 64 * k;
            // (wave-uniform: a wave's 64 slots are 8 rows of one frame) rows of frames outside the clip stay zero
            const int f = f0 + ((wave * 8 + 64 * k) >> 4) - 1;
            if (wave * 8 + 64 * k < IROWS && (unsigned)f < (unsigned)p.F) {
                f16x8* q = (f16x8*)(img + buf * IMG + r * 128 + ((nch ^ (r & 7)) << 4));
                const f16x8 v = *q;
                f16x8 o;
#pragma unroll
                for (int j = 0; j < 8; j += 2) {      // channel pairs: v_pk_fma / v_pk_mul / v_pk_add_f32
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    const f32x2 x = {(float)v[j], (float)v[j + 1]};
                    const f32x2 y = x * (f32x2){a[j], a[j + 1]} + (f32x2){sh[j], sh[j + 1]};
                    const f32x2 u = y * -1.44269504088896341f;
                    const f32x2 d = (f32x2){__builtin_amdgcn_exp2f(u[0]), __builtin_amdgcn_exp2f(u[1])} + 1.0f;
                    const f32x2 r = y * (f32x2){__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
                    o[j] = (f16)r[0];
                    o[j + 1] = (f16)r[1];
                }
                *q = o;
            }
        }
    };

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
    const int nsteps = nslices * 3;
    for (int step = 0, s = 0, tap = 0; step < nsteps; ++step) {
        const int wcur = step & 1, icur = s & 1;
        if (step + 1 < nsteps) issue_w(step + 1, wcur ^ 1);                  // (buffer last read before the previous barrier)
        if (tap == 0 && s + 1 < nslices) issue_img(s + 1, icur ^ 1);        // landed two barriers from here, normalised in tap 2
        const char* As = img + icur * IMG + tap * 2048;                     // tap kt reads frame f + kt - 1 = image row tile + kt
        const char* Bs = wt + wcur * WT;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8 af[TM];
            const int c = ks * 4 + fq;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = (wm * TM + i) * 16 + frow;
                af[i] = *(const f16x8*)(As + row * 128 + ((c ^ (row & 7)) << 4));
            }
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
        if (tap == 2 && s + 1 < nslices) normalise(s + 1, icur ^ 1);        // its DMA was waited for by the barrier after tap 1
        if (++tap == 3) {
            tap = 0;
            ++s;
        }
        __syncthreads();                  // next weights (and image) landed, this step's buffers fully read
    }

    // ---- epilogue: bias + residual, 16-byte stores; row (i, frow) of this wave = frame f0 + wm TM + i, pixel p0 + frow
    constexpr int NA = TN / 2;
    const int nb = n0 + wn * (BN / WN);
    f16x8 bv[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        const int n = nb + a * 32 + fq * 8;
        bv[a] = *(const f16x8*)((p.bias && n < p.N) ? p.bias + n : zp);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int px = p0 + frow;
        const bool row_ok = px < p.S;
        const size_t m = ((size_t)(b * p.F + f0 + wm * TM + i)) * p.S + (row_ok ? px : 0);
        f16x8 rv[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const int n = nb + a * 32 + fq * 8;
            rv[a] = *(const f16x8*)((p.res && row_ok && n < p.N) ? p.res + m * p.ldr + n : zp);
        }
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const int n = nb + a * 32 + fq * 8;
            f16x8 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = (f16)(acc[i][2 * a][j] + (float)bv[a][j] + (float)rv[a][j]);
                o[4 + j] = (f16)(acc[i][2 * a + 1][j] + (float)bv[a][4 + j] + (float)rv[a][4 + j]);
            }
            if (row_ok && n < p.N) *(f16x8*)(p.out + m * p.ldo + n) = o;
        }
    }
}

// frames per tile: the largest of 16 / 12 / 8 that divides F (a chunk pays two halo frames: 16 -> 12.5 %, 12 -> 17 %, 8 -> 25 %)
static int tconv_ft(int F) { return F % 16 == 0 ? 16 : F % 12 == 0 ? 12 : F % 8 == 0 ? 8 : 0; }

extern "C" int vdx_tconv_gn_supported(int C, int N, int F) {
    return C > 0 && C % 64 == 0 && N > 0 && N % 320 == 0 && tconv_ft(F) != 0 ? 1 : 0;
}

// Is K3 expected to be faster than the apply pass + TCONV3 GEMM?  Measured (tools/tconv_bench.py, profiles/r04_k3.md): with ONE
// column tile (N = 320: the image is staged and normalised once) and enough tiles to fill the chip several times it is
// 1.16-1.25 x faster; with 2 or 4 column tiles every tile normalises the image again and it is 0.6-0.98 x.
extern "C" int vdx_tconv_gn_preferred(int C, int N, int B, int F, int S) {
    if (!vdx_tconv_gn_supported(C, N, F) || N != 320) return 0;
    // a function of ONE sample's shape (C, N, F, S) — not of B, not of the chip: see vdx_conv3x3_gn_preferred
    (void)B;
    const long long tiles_per_sample = (long long)(F / tconv_ft(F)) * ((S + 15) / 16);
    return tiles_per_sample >= 512 ? 1 : 0;
}

template <int FT>
static int tconv_launch(const TcP& p, hipStream_t st) {
    constexpr int lds = 2 * (FT + 2) * 16 * 128 + 2 * 320 * 128;
    auto kern = tconv_gn_kernel<FT>;
    static const hipError_t attr_rc = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (attr_rc != hipSuccess) return vdx_fail("tconv_gn: cannot reserve %d bytes of LDS", lds);
    TcP q = p;
    q.nfc = p.F / FT;
    const long long blocks = (long long)p.B * q.nfc * p.npb * p.ntn;
    VDX_CHECK(blocks < (1ll << 31), "tconv_gn: grid too large");
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), lds, st, q);
    return vdx_launch_status("vdx_tconv_gn_f16");
}

extern "C" int vdx_tconv_gn_f16(const void* x, int ldx, const float* scale_shift, const void* w, const void* bias,
                                const void* residual, int ldr, void* out, int ldo, int B, int F, int S, int C, int N,
                                vdx_stream_t stream) {
    VDX_CHECK(x && scale_shift && w && out, "tconv_gn: null pointer");
    VDX_CHECK((uintptr_t)scale_shift % 16 == 0, "tconv_gn: scale_shift must be 16-byte aligned (it is read with 16-byte loads)");
    VDX_CHECK(B > 0 && F > 0 && S > 0, "tconv_gn: empty problem");
    VDX_CHECK(vdx_tconv_gn_supported(C, N, F), "tconv_gn: C=%d (%% 64), N=%d (%% 320), F=%d (%% 8) not supported", C, N, F);
    VDX_CHECK(ldx % 8 == 0 && ldo % 8 == 0 && (!residual || ldr % 8 == 0) && ldx >= C && ldo >= N, "tconv_gn: leading dimensions");
    VDX_CHECK((long long)B * F * S < (1ll << 31), "tconv_gn: too many rows");
    TcP p;
    p.x = (const f16*)x; p.w = (const f16*)w; p.bias = (const f16*)bias; p.res = (const f16*)residual; p.out = (f16*)out;
    p.ab = scale_shift;
    p.ldx = ldx; p.ldo = ldo; p.ldr = ldr;
    p.B = B; p.F = F; p.S = S; p.C = C; p.N = N;
    p.npb = (S + 15) / 16; p.ntn = N / 320; p.nfc = 0;
    hipStream_t st = (hipStream_t)stream;
    switch (tconv_ft(F)) {
        case 16: return tconv_launch<16>(p, st);
        case 12: return tconv_launch<12>(p, st);
        default: return tconv_launch<8>(p, st);
    }
}
