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
// Tiling: BM x BN x 64 block tile, waves own 64x64 (16 accumulators of v_mfma_f32_16x16x32_f16).
// Staging is LDS-DMA (global_load_lds_dwordx4): every lane supplies its own 16-byte SOURCE
// address — which is what makes the conv gathers, the zero padding (source = a 128-byte zero
// page), the concat and the N/M tails free — while the LDS image stays lane-linear (1 KiB = 8 rows
// of 128 B per wave-instruction).  The XOR swizzle that makes every ds_read_b128 conflict-free is
// therefore applied to the source chunk index and to the read address, never to the destination.
// LDS is double-buffered with ONE barrier per K-tile: tile k+1 streams in while tile k feeds the
// MFMAs.  The MFMA is issued with the weight fragment as the A operand, so a lane ends up holding
// 8 consecutive output channels of one row -> 16-byte epilogue stores.
#include "gemm_common.h"
#include <stdlib.h>

#ifdef VDX_STAMPS   // diagnostic build only (make stamps): per-block phase clocks, never in the product library
#define STAMP_MAX 32768
static __device__ unsigned long long g_stamps[STAMP_MAX * 8];
extern "C" int vdx_debug_read_stamps(void* dst, int nblocks) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps), (size_t)nblocks * 64) == hipSuccess ? 0 : -1;
}
#define STAMP(i) st_[i] = __builtin_amdgcn_s_memtime()
#else
#define STAMP(i)
#endif

// ---- staging of one output tile's K tiles (LDS-DMA) -------------------------------------------
// Issue slot `it` (of IT per operand) fills, with its DMA instruction i, 16-byte slot it&7 of physical LDS
// row prow + i*RPP.  Slot s of physical row r must hold data chunk s ^ (r & 7); RPP is a multiple of 8, so
// the chunk is the same for every i.  Rows past M (and weight rows past N) are CLAMPED to the last valid
// row: their products land in accumulators the epilogue never stores.
//
// SPLIT (8-wave blocks): waves 0-3 stage the activation rows, waves 4-7 the weight rows, 256 issue slots
// each.  Waves w and w+4 share a SIMD; a DMA instruction costs its wave ~60-100 issue cycles, and with
// all eight waves issuing their share at the top of a K tile the matrix pipe sat idle for that long in
// every K tile (measured: DMA-only 1.0 us and MFMA-only 1.65 us per K tile, together 2.1 us).  Now the
// activation waves issue at the top while their partners run MFMAs, and the weight waves issue between
// their two K halves while the activation waves run MFMAs.  The weights are L2-resident (0.2-0.4 us to
// land), so half a K tile of lead is enough for them; the activation rows, which may come from HBM, keep
// the full K tile of lead.
template <int BM, int BN, int NT, int MODE, bool SPLIT, bool UPS2 = false>
struct Stager {
    static constexpr int IT = SPLIT ? NT / 2 : NT;        // issue slots per operand
    static constexpr int RPP = IT / 8;                    // LDS rows covered by one DMA instruction of all slots
    static constexpr int ACH = BM / RPP, BCH = BN / RPP;  // DMA instructions per slot and K tile: A rows, W rows
    static constexpr int STAGE = (BM + BN) * 128;
    static_assert(BM % RPP == 0 && BN % RPP == 0 && RPP % 32 == 0, "tile shape");
    static_assert(!SPLIT || NT == 512, "SPLIT pairs wave w with wave w+4");
    int d0[ACH], d1[ACH];   // MODE 0: source row | MODE 1: image base row, (y+1)<<16|(x+1) | MODE 2: row, frame
    int wrow0, csw, iwave;
    int tap, kc, kw;        // the NEXT K tile to be issued (kw = its first weight column)

    __device__ __forceinline__ void setup(const GemmP& p, int tid, int m0, int n0) {
        const int it = SPLIT ? (tid & (IT - 1)) : tid;
        const int prow = it >> 3;
        iwave = __builtin_amdgcn_readfirstlane(it >> 6);
        csw = ((it & 7) ^ (prow & 7)) * 8;    // element offset of the data chunk this slot fetches
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            const int mm = min(m0 + prow + i * RPP, p.M - 1);
            if (MODE == 0) {
                d0[i] = mm;
                d1[i] = 0;
            } else if (MODE == 1) {
                const int per = p.h_out * p.w_out;
                const int n = mm / per, rem = mm - n * per;
                const int yo = rem / p.w_out, xo = rem - yo * p.w_out;
                d0[i] = n * p.h_in * p.w_in;
                d1[i] = ((yo * p.stride) << 16) | (xo * p.stride);
            } else {
                d0[i] = mm;
                d1[i] = (mm / p.hw) % p.frames;
            }
        }
        // Physical LDS row pr holds weight row 8q+4b+j where pr = 16b+4q+j inside each 32-row group, so
        // that after the MFMA a lane owns 8 consecutive output columns (RPP % 32 == 0: same for every i).
        wrow0 = n0 + ((((prow >> 2) & 3) << 3) | (((prow >> 4) & 1) << 2) | (prow & 3)) + (prow & ~31);
        tap = kc = kw = 0;
    }
    // weight rows of the next K tile -> stage buffer `buf`
    __device__ __forceinline__ void issue_w(const GemmP& p, char* smem, int buf) {
        char* sb = smem + buf * STAGE + BM * 128 + iwave * 1024;
#pragma unroll
        for (int i = 0; i < BCH; ++i) {
            const int r = min(wrow0 + i * RPP, p.N - 1);
            __builtin_amdgcn_global_load_lds((gptr_t)(p.w + (size_t)r * p.K + kw + csw), (lptr_t)(sb + i * RPP * 128), 16, 0, 0);
        }
    }
    // (gathered) activation rows of the next K tile -> stage buffer `buf`
    __device__ __forceinline__ void issue_a(const GemmP& p, char* smem, int buf) {
        const f16* zp = (const f16*)g_zero_page;
        char* sa = smem + buf * STAGE + iwave * 1024;
        if (MODE == 0) {
            const bool first = kc < p.c1;
            const f16* base = first ? p.a + kc + csw : p.a2 + (kc - p.c1) + csw;
            const int ld = first ? p.lda : p.lda2;
#pragma unroll
            for (int i = 0; i < ACH; ++i)
                __builtin_amdgcn_global_load_lds((gptr_t)(base + (size_t)d0[i] * ld), (lptr_t)(sa + i * RPP * 128), 16, 0, 0);
        } else if (MODE == 1) {
            const int ky = tap / 3, kx = tap - ky * 3;
            const int hlim = p.h_up, wlim = p.w_up;
#pragma unroll
            for (int i = 0; i < ACH; ++i) {
                int y = (d1[i] >> 16) + ky - 1, x = (d1[i] & 0xffff) + kx - 1;
                const bool ok = (unsigned)y < (unsigned)hlim && (unsigned)x < (unsigned)wlim;
                if (UPS2) {                   // explicit target size (latent not divisible by 8): F.interpolate(size=, "nearest")
                    y = min((int)((float)y * p.usy), p.h_in - 1);
                    x = min((int)((float)x * p.usx), p.w_in - 1);
                } else {                      // (its own instantiation: the float index math costs the issue-bound K loop 11 %)
                    y >>= p.ups;
                    x >>= p.ups;
                }
                const f16* src = p.a + (size_t)(d0[i] + y * p.w_in + x) * p.lda + kc + csw;
                __builtin_amdgcn_global_load_lds((gptr_t)(ok ? src : zp), (lptr_t)(sa + i * RPP * 128), 16, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < ACH; ++i) {
                const int f = d1[i] + tap - 1;
                const bool ok = (unsigned)f < (unsigned)p.frames;
                const f16* src = p.a + (size_t)(d0[i] + (tap - 1) * p.hw) * p.lda + kc + csw;
                __builtin_amdgcn_global_load_lds((gptr_t)(ok ? src : zp), (lptr_t)(sa + i * RPP * 128), 16, 0, 0);
            }
        }
    }
    // start the K walk at K tile kt (split-K slices)
    __device__ __forceinline__ void seek(int kt) {
        kw = kt * 64;
        if (MODE == 0) {
            kc = kw;
        } else {
            constexpr int T = MODE == 1 ? 9 : 3;
            const int slice = kt / T;
            tap = kt - slice * T;
            kc = slice * 64;
        }
    }
    __device__ __forceinline__ void advance() {
        // K order of the gathers is (64-channel slice, tap, channel): all taps of one channel slice are
        // consumed back to back, so the shifted re-reads of the same source pixels hit the XCD's L2
        // (tap-major order streamed ~0.5 MB per CU between re-uses and thrashed it: 7.5x re-fetch).
        kw += 64;
        if (MODE == 0) {
            kc += 64;
        } else if (++tap == (MODE == 1 ? 9 : 3)) {
            tap = 0;
            kc += 64;
        }
    }
};

// VAR: 0 the product kernels | 1 a split-K slice (p.ksplit > 1) | 2 the 3x3 gather with nearest-to-size upsampling.
// (Variants 1 and 2 are their own instantiations so that the kernels of every ordinary step stay exactly the round-2
// code: run-time `ksplit` / `ups == 2` branches in the one template cost the 3x3-conv kernel 11 % — 894 -> 995 us per
// launch in the XL step, profiles/r03_kernel_stats.csv history.)
template <int BM, int BN, int WM, int WN, int MODE, bool GEGLU, bool SPLIT, int VAR>
__global__ __launch_bounds__(WM* WN * 64) void gemm_kernel(const GemmP p) {
    constexpr int NT = WM * WN * 64;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr bool KS = VAR == 1;
#ifdef VDX_GEMM_PLAIN_LOOP                  // timing build (tools/gemm_abl.sh): the K loop without the rolling fragment prefetch
    constexpr bool defined_gemm_plain_loop = true;
#else
    constexpr bool defined_gemm_plain_loop = false;
#endif
    typedef Stager<BM, BN, NT, MODE, SPLIT, VAR == 2> Stage;
    constexpr int STAGE = Stage::STAGE;
    static_assert(TN % 2 == 0, "tile shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
#ifdef VDX_STAMPS
    unsigned long long st_[4];
    const unsigned long long r0_ = __builtin_amdgcn_s_memrealtime();
    STAMP(0);
#endif
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int frow = lane & 15, fq = lane >> 4;
    const int nk = p.K >> 6;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    // split-K (p.ksplit > 1): block = (tile, slice); the slices of a tile are neighbours, so they share an XCD's L2
    int kt_lo = 0, kt_hi = nk, slab = 0;
    if (KS) {
        slab = bid;
        const int sl = bid % p.ksplit;
        bid /= p.ksplit;
        kt_lo = sl * nk / p.ksplit;
        kt_hi = (sl + 1) * nk / p.ksplit;
    }
    int mt_, nt_;
    gemm_tile_of(bid, p.ntm, p.ntn, mt_, nt_);
    const int m0 = p.m_begin + mt_ * BM, n0 = nt_ * BN;
    const bool does_a = !SPLIT || wave < 4, does_w = !SPLIT || wave >= 4;

    float2* gelu = (float2*)(smem + 2 * STAGE);    // GEGLU: table behind the stage buffers (first barrier publishes it)
    if (GEGLU) gelu_tab_init(gelu, tid, NT);
    Stage sg;
    sg.setup(p, tid, m0, n0);
    if (KS && kt_lo) sg.seek(kt_lo);
    if (does_w) sg.issue_w(p, smem, 0);
    if (does_a) sg.issue_a(p, smem, 0);
    sg.advance();

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();                      // LDS-DMA in flight: the barrier's fence waits vmcnt(0)
    STAMP(1);
    for (int kt = KS ? kt_lo : 0; kt < (KS ? kt_hi : nk); ++kt) {
        const int cur = KS ? (kt - kt_lo) & 1 : kt & 1;
        const bool more = kt + 1 < (KS ? kt_hi : nk);
#if !(defined(VDX_STAMPS) && VDX_ABL == 2)   // diagnostic ablation 2: no DMA inside the K loop
        // buffer cur^1 was last read before the previous barrier
#if !(defined(VDX_STAMPS) && VDX_ABL == 5)    // ablation 5: no activation DMA in the loop
        if (more && does_a) sg.issue_a(p, smem, cur ^ 1);
#endif
#if !(defined(VDX_STAMPS) && VDX_ABL == 6)    // ablation 6: no weight DMA in the loop
        if (more && !SPLIT) sg.issue_w(p, smem, cur ^ 1);
#endif
#endif
        const char* As = smem + cur * STAGE;
        const char* Bs = As + BM * 128;
#if !defined(VDX_STAMPS) || VDX_ABL == 7       // (the stamped builds keep the plain loop below unless ablation 7 asks for this one)
        if constexpr (TM == 4 && TN == 10 && !defined_gemm_plain_loop) {
            // ---- the 64x160 wave tile: ROLLING fragment prefetch.  A K half is two groups of 20 MFMAs (column tiles 0-4, 5-9)
            // walked column-major, so a weight fragment is dead after four MFMAs and the read of the fragment that takes its
            // place — the next group's, the next K half's — is issued right behind them, into the same registers: the LDS
            // latency of all but the K tile's first nine reads (nothing can be read before the barrier) and the four
            // activation fragments of the second K half (SPLIT: the nine of either K half) runs under MFMAs.  hipcc on its own reads a group's fragments,
            // drains the LDS queue, issues the 20 MFMAs, four times per K tile; the order here is pinned by
            // sched_group_barrier.  Same MFMAs on the same accumulators in the same K order: same bits.
            f16x8 af[TM], bf[5];
            const int arow = (wm * WTM + frow) * 128, brow = (wn * WTN + frow) * 128, sw = frow & 7;   // (tile rows are multiples of 16: row & 7 = frow & 7)
            auto rdA = [&](int ks) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = *(const f16x8*)(As + arow + i * 2048 + (((ks * 4 + fq) ^ sw) << 4));
            };
            auto rdB = [&](int j, int ks, int g) __attribute__((always_inline)) {
                bf[j] = *(const f16x8*)(Bs + brow + (g * 5 + j) * 2048 + (((ks * 4 + fq) ^ sw) << 4));
            };
            auto mm4 = [&](int j, int g) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    acc[i][g * 5 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][g * 5 + j], 0, 0, 0);
            };
            rdA(0);
#pragma unroll
            for (int j = 0; j < 5; ++j) rdB(j, 0, 0);
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                mm4(j, 0);
                rdB(j, 0, 1);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 9, 0);
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            if constexpr (!SPLIT) {
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    mm4(j, 1);
                    rdB(j, 1, 0);
                }
                rdA(1);
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            } else {
                // (the gathers' weight-staging waves issue their DMA between the K halves: no fragment is carried across —
                // the staging descriptors and 36 live fragment registers do not fit beside the accumulators)
#pragma unroll
                for (int j = 0; j < 5; ++j) mm4(j, 1);
                __builtin_amdgcn_sched_group_barrier(0x008, 20, 0);
                if (more && does_w) sg.issue_w(p, smem, cur ^ 1);
                rdA(1);
#pragma unroll
                for (int j = 0; j < 5; ++j) rdB(j, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 9, 0);
            }
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                mm4(j, 0);
                rdB(j, 1, 1);
            }
#pragma unroll
            for (int j = 0; j < 5; ++j) mm4(j, 1);
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 20, 0);
        } else
#endif
#if defined(VDX_STAMPS) && VDX_ABL == 1         // diagnostic ablation 1: DMA only, no LDS reads / MFMA
        if (false)
#endif
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#if !(defined(VDX_STAMPS) && (VDX_ABL == 2 || VDX_ABL == 6))
            if (SPLIT && ks == 1 && more && does_w) sg.issue_w(p, smem, cur ^ 1);
#endif
            f16x8 af[TM];
            const int c = ks * 4 + fq;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * WTM + i * 16 + frow;
                af[i] = *(const f16x8*)(As + row * 128 + ((c ^ (row & 7)) << 4));
            }
            // weight fragments in groups of <= 5 (bounds live registers on the 160-wide wave tile)
            constexpr int NG = TN > 5 ? 2 : 1, GS = TN / NG;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                f16x8 bf[GS];
#pragma unroll
                for (int j = 0; j < GS; ++j) {
                    const int row = wn * WTN + (g * GS + j) * 16 + frow;
                    bf[j] = *(const f16x8*)(Bs + row * 128 + ((c ^ (row & 7)) << 4));
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < GS; ++j)
                        acc[i][g * GS + j] =
                            __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][g * GS + j], 0, 0, 0);
            }
        }
        if (more) sg.advance();
        __syncthreads();                  // next K tile landed (vmcnt(0)) and this one is fully read
    }

    STAMP(2);
    if (KS) {
        // this slice's fp32 accumulators -> its slab, in register order ([accumulator][thread][4]: 16-byte coalesced
        // stores); vdx_gemm_reduce_kernel adds the slabs of a tile in slice order and runs the epilogue
        float* dst = p.partial + (size_t)slab * (NT * TM * TN * 4) + tid * 4;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) *(f32x4*)(dst + (i * TN + j) * NT * 4) = acc[i][j];
        return;
    }
    gemm_epilogue<TM, TN, GEGLU>(p, acc, m0 + wm * WTM, n0 + wn * WTN, frow, fq, gelu);
#ifdef VDX_STAMPS
    __builtin_amdgcn_s_waitcnt(0);        // stores drained
    STAMP(3);
    if (tid == 0 && blockIdx.x < STAMP_MAX) {
        unsigned long long* d = g_stamps + (size_t)blockIdx.x * 8;
        for (int i = 0; i < 4; ++i) d[i] = st_[i];
        d[4] = r0_;
        d[5] = __builtin_amdgcn_s_memrealtime();
        d[6] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)) |
               ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32);
        d[7] = (unsigned long long)bid;
    }
#endif
}

// ---- host side ------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int MODE, bool GEGLU, bool SPLIT = false, int VAR = 0>
static int launch(const GemmP& p, hipStream_t st) {
    constexpr int lds = 2 * (BM + BN) * 128 + (GEGLU ? GELU_TAB_BYTES : 0);
    auto kern = gemm_kernel<BM, BN, WM, WN, MODE, GEGLU, SPLIT, VAR>;
    // one-time LDS opt-in; a function-local static is initialised exactly once even under concurrent callers
    static const hipError_t attr_rc = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (attr_rc != hipSuccess) return vdx_fail("gemm: cannot reserve %d bytes of LDS", lds);
    GemmP q = p;
    q.ntn = (p.N + BN - 1) / BN;
    const int ntm = (p.M - p.m_begin + BM - 1) / BM;
    q.ntm = ntm;
    hipLaunchKernelGGL(kern, dim3(ntm * q.ntn), dim3(WM * WN * 64), lds, st, q);
    return vdx_launch_status("vdx_gemm_f16");
}

// ---- split-K tails -----------------------------------------------------------------------------
// A product whose 256x320 tiles do not fill a whole number of rounds of 256 ends in a mostly idle round.  Its tail
// (fewer than half a round of tiles) can instead be computed as `ksplit` K slices per tile — all CUs busy for 1/ksplit
// of a tile time — followed by this reduction: one wave per (tile, wave of the tile) sums the slices' fp32 slabs in
// slice order (fixed: deterministic) and runs the SAME epilogue code on the sum.  The 16-frame windows of BASELINE
// cfg4 / cfg5 are where it pays (level 2: 288 tiles = 1.125 rounds).  It changes the summation order of those rows
// (K slices accumulated separately, then added), so it is a per-call opt-in (vdx_gemm_args.ksplit).
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64) void gemm_reduce_kernel(const GemmP p) {
    constexpr int NT = WM * WN * 64, WTM = BM / WM, WTN = BN / WN, TM = WTM / 16, TN = WTN / 16;
    const int lane = threadIdx.x;
    const int wave = blockIdx.x % (WM * WN), tile = blockIdx.x / (WM * WN);
    int mt_, nt_;
    gemm_tile_of(tile, p.ntm, p.ntn, mt_, nt_);
    const int m0 = p.m_begin + mt_ * BM, n0 = nt_ * BN;
    f32x4 acc[TM][TN];
    const float* src = p.partial + (size_t)tile * p.ksplit * (NT * TM * TN * 4) + (wave * 64 + lane) * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = *(const f32x4*)(src + (i * TN + j) * NT * 4);
    for (int sl = 1; sl < p.ksplit; ++sl) {
        const float* s2 = src + (size_t)sl * (NT * TM * TN * 4);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const f32x4 v = *(const f32x4*)(s2 + (i * TN + j) * NT * 4);
                acc[i][j] = acc[i][j] + v;
            }
    }
    gemm_epilogue<TM, TN, false>(p, acc, m0 + (wave / WN) * WTM, n0 + (wave % WN) * WTN, lane & 15, lane >> 4);
}

static constexpr size_t KSPLIT_SLAB_BYTES = 256 * 320 * 4;     // fp32 accumulators of one 256x320 tile

template <int MODE>
static int launch_ksplit(const GemmP& p, int ksplit, float* ws, hipStream_t st) {
    constexpr int BM = 256, BN = 320, WM = 4, WN = 2;
    constexpr int lds = 2 * (BM + BN) * 128;
    auto kern = gemm_kernel<BM, BN, WM, WN, MODE, false, MODE != 0, 1>;
    static const hipError_t attr_rc = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (attr_rc != hipSuccess) return vdx_fail("gemm: cannot reserve %d bytes of LDS", lds);
    GemmP q = p;
    q.ntn = (p.N + BN - 1) / BN;
    q.ntm = (p.M - p.m_begin + BM - 1) / BM;
    q.ksplit = ksplit;
    q.partial = ws;
    const int tiles = q.ntm * q.ntn;
    hipLaunchKernelGGL(kern, dim3(tiles * ksplit), dim3(WM * WN * 64), lds, st, q);
    hipLaunchKernelGGL((gemm_reduce_kernel<BM, BN, WM, WN>), dim3(tiles * WM * WN), dim3(64), 0, st, q);
    return vdx_launch_status("vdx_gemm_f16 (split-K tail)");
}

// Kernel choice.  `force` (vdx_gemm_args.epilogue bits 8..11, a testing/tuning knob) pins a
// variant: 1 = 128x128 two-stage (eight waves; 9 = four waves), 2 = 256x320 two-stage (K-step 64), 3 = 256x320 four-stage ring
// (K-step 32), 4 = 128x320 two-stage ring with two blocks per CU, 8 = 128x320 four-stage ring with eight
// 32x160 waves, 5 = 256x64, 6 = variant 2 without the
// split staging roles (every wave issues its share of both operands at the top of the K tile); 7 (handled in
// vdx_gemm_f16) = the weights-stationary short-K kernels of gemm_ws.hip.
//
// Channel widths of this UNet are multiples of 320: the 320-wide tiles (64x160 per wave) halve LDS/L2 bytes per MFMA
// against 128x128.  Widths that are not multiples of 320 (transformer_in: 512/1536/4096) still take the 320-wide tile
// when the masked tail wastes < 25 % of the last column of tiles.  Among the candidates a launch costs
// (rounds of 256 tiles) x (time of one tile); tile times relative to the 256x320 tile (= 10), fitted to
// tools/gemm_bench.py at 24 / 16 / 12 frames (profiles/r02_tools.txt): 128x320 as eight 32x160 waves 8 (half the
// work at 62 % of the efficiency), 128x128 3 (a fifth of the work at 67 %).  The 16-frame windows of BASELINE cfg4/5
// are where this matters: M = 18 432 rows at level 2 is 288 tiles of 256x320 = two rounds for 1.125 rounds of work,
// and the 128x128 kernel (5.6 rounds of small tiles) is 10-16 % faster there; level 3 at 16 / 12 frames likewise.
// K-32 rings lose to K-64 on every large shape (profiles/r01_gemm_variants.txt).
struct TileChoice {
    int v;            // variant
    long long cost;   // in tenths of a 256x320 tile time
};
static TileChoice choose_tile(long long rows, int N) {
    const int nt320 = (N + 319) / 320;
    const bool fits = nt320 * 320 * 4 <= N * 5 && rows >= 1024;   // (swapped V^T products have M = C)
    const long long t1 = ((rows + 127) / 128) * ((N + 127) / 128);
    const long long c1 = 3 * ((t1 + 255) / 256);
    if (!fits) return TileChoice{N > 64 ? 1 : 5, c1};
    const long long t256 = ((rows + 255) / 256) * nt320, t128 = ((rows + 127) / 128) * nt320;
    const long long c2 = 10 * ((t256 + 255) / 256), c8 = 8 * ((t128 + 255) / 256);
    if (c2 <= c8 && c2 <= c1) return TileChoice{2, c2};
    return c8 <= c1 ? TileChoice{8, c8} : TileChoice{1, c1};
}
// Row at which to split a product into [begin, split) on 256x320 tiles (whole rounds of 256) + [split, end) on whatever
// suits the rest, or 0: the last round of a 256x320 launch is otherwise as slow as a full one however few tiles it has.
static int choose_split(int begin, int end, int N) {
    const long long rows = end - begin;
    const TileChoice whole = choose_tile(rows, N);
    const int nt320 = (N + 319) / 320;
    // (also when the WHOLE product prefers small tiles — 18 432 rows x 1280: 288 big tiles = two rounds for 1.125 — whole
    // rounds of big tiles + a small-tile tail can still win: one round at 1.2 PFLOP/s + 160 small tiles)
    if (whole.v != 2 && !(nt320 * 320 * 4 <= N * 5 && rows >= 1024)) return 0;
    const long long t256 = ((rows + 255) / 256) * nt320;
    const long long full = t256 / 256;
    // The main launch is EXACT-FIT: `full` rounds of one tile per CU.  Beside a collective whose channel kernels hold a few CUs
    // (a multi-GPU run) it runs a whole round more: +12...+38 % on the 100 launches of exactly 256 tiles a 16-frame window has
    // (profiles/r06_rccl_contention.md, section 2b).  With a reserve in force (vdx_set_reserved_cus; 0 on one GPU: nothing
    // changes) a round of the main launch fills only the unreserved CUs and the tail takes the rows that are left.  Everything
    // else stays priced on the full chip: pricing ALL rounds on the reserved count changes tile families everywhere (+6.5 %).
    const long long nc = vdx_grid_cus() < 256 ? vdx_grid_cus() : 256;
    if (full == 0 || (t256 % 256 == 0 && nc == 256)) return 0;
    const long long mt_main = full * nc / nt320;                  // m tiles of the main launch (<= `full` rounds)
    if (mt_main == 0) return 0;
    const int split = begin + (int)(mt_main * 256);
    if (split >= end) return 0;
    const TileChoice tail = choose_tile(end - split, N);
    const long long cost = 10 * ((mt_main * nt320 + 255) / 256) + tail.cost + 1;   // + 1: the second launch's ramp
    return cost * 100 <= whole.cost * 94 ? split : 0;      // (measured in the step at 24 / 16 / 12 frames: 94 >= 100 >= 104 > 110)
}

template <int MODE, bool GEGLU>
static int pick_tile(const GemmP& p, int force, hipStream_t st) {
    const int v = force ? force : choose_tile(p.M - p.m_begin, p.N).v;
    if constexpr (MODE == 1 && !GEGLU) {
        if (p.ups == 2)     // nearest-to-size gather: two instantiations of its own (latents not divisible by 8 only)
            return v == 1 ? launch<128, 128, 2, 2, 1, false, false, 2>(p, st) : launch<256, 320, 4, 2, 1, false, true, 2>(p, st);
    }
    switch (v) {
        case 1: return launch<128, 128, 4, 2, MODE, GEGLU>(p, st);     // EIGHT waves of 32x64: two waves per SIMD (3-10 % over four 64x64 waves on the tails, same bits)
        case 2: return launch<256, 320, 4, 2, MODE, GEGLU, MODE != 0>(p, st);   // split roles pay on the gathers only
        case 6: return launch<256, 320, 4, 2, MODE, GEGLU, false>(p, st);
        case 3: return vdx_gemm_ring_launch(p, MODE, GEGLU, 0, st);
        case 4: return vdx_gemm_ring_launch(p, MODE, GEGLU, 1, st);
        case 8: return vdx_gemm_ring_launch(p, MODE, GEGLU, 2, st);
        case 5: return launch<256, 64, 4, 1, MODE, GEGLU>(p, st);
        case 9: return launch<128, 128, 2, 2, MODE, GEGLU>(p, st);     // the four-wave form (64x64 per wave), kept for comparison
    }
    return vdx_fail("gemm: unknown kernel variant %d", v);
}

// validation + the kernel-side parameter block, shared by the launch and by vdx_gemm_plan
static int gemm_prepare(const vdx_gemm_args* a, GemmP& p, bool& geglu, int& force, int& ws_family) {
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
    p.a = (const f16*)a->a; p.a2 = (const f16*)a->a2; p.w = (const f16*)a->w;
    p.bias = (const f16*)a->bias; p.bias2 = (const f16*)a->bias2; p.res = (const f16*)a->residual;
    p.out = (f16*)a->out;
    VDX_CHECK(a->row_begin >= 0 && (a->row_end == 0 || (a->row_end > a->row_begin && a->row_end <= a->M)) && a->row_begin < a->M,
              "gemm: rows [%d, %d) of %d", a->row_begin, a->row_end, a->M);
    // (kernels mask rows >= p.M: the end of the row range; geometry checks below use the whole product's a->M)
    p.M = a->row_end ? a->row_end : a->M; p.m_begin = a->row_begin; p.N = a->N; p.K = a->K; p.c1 = a->c1; p.c2 = a->c2;
    p.lda = a->lda; p.lda2 = a->lda2; p.ldo = a->ldo; p.ldr = a->ldr;
    p.h_in = a->h_in; p.w_in = a->w_in; p.h_out = a->h_out; p.w_out = a->w_out;
    p.stride = a->stride; p.ups = a->upsample;     // (the gathers shift by ups: 0 | 1; ups == 2 runs the VAR = 2 kernels)
    p.h_up = a->upsample == 1 ? 2 * a->h_in : a->upsample == 2 ? a->h_out : a->h_in;
    p.w_up = a->upsample == 1 ? 2 * a->w_in : a->upsample == 2 ? a->w_out : a->w_in;
    p.usy = a->h_in > 0 && p.h_up > 0 ? (float)a->h_in / (float)p.h_up : 1.0f;
    p.usx = a->w_in > 0 && p.w_up > 0 ? (float)a->w_in / (float)p.w_up : 1.0f;
    p.frames = a->frames; p.hw = a->hw; p.rpb2 = a->rows_per_bias2 > 0 ? a->rows_per_bias2 : 1;
    p.ldb2 = a->ldb2 > 0 ? a->ldb2 : a->N;
    VDX_CHECK(p.ldb2 % 8 == 0, "gemm: ldb2 must be a multiple of 8");
    p.ntn = 0; p.ntm = 0; p.ksplit = 0; p.partial = nullptr;
    p.wset_rows = a->wset_rows; p.wset_bias = a->wset_bias;
    geglu = (a->epilogue & VDX_EPI_GEGLU) != 0;
    force = (a->epilogue >> 8) & 15;   // kernel variant override (0 = automatic)
    if (geglu) {
        VDX_CHECK(a->mode == VDX_GEMM_PLAIN && !a->bias2 && !a->residual, "gemm: GEGLU epilogue is plain-mode only");
        VDX_CHECK(a->ldo % 4 == 0, "gemm: GEGLU ldo must be a multiple of 4");
    }
    // short-K Linear layers on many rows (levels 0/1, transformer_in): weights-stationary streaming kernels
    // (variant 7 pins them); they walk whole products only
    const bool whole = a->row_begin == 0 && (a->row_end == 0 || a->row_end == a->M);
    ws_family = 0;
    if (whole && (force == 7 || (force == 0 && a->M >= 16384))) {
        ws_family = vdx_gemm_ws_family(p, a->mode, geglu);
        VDX_CHECK(ws_family || force != 7, "gemm: variant 7 (weights-stationary) needs plain single-source rows, K in {320, 512, 640}, N %% 32 == 0, M %% 64 == 0");
    } else {
        VDX_CHECK(force != 7, "gemm: variant 7 (weights-stationary) computes whole products (row_begin / row_end unset)");
    }
    if (a->wset_rows != 0) {        // a weight set per row range (GroupNorm folded into the Linear): weights-stationary kernels only
        VDX_CHECK(a->wset_rows > 0 && a->wset_bias && a->M % a->wset_rows == 0, "gemm: wset_rows=%d needs wset_bias and M %% wset_rows == 0", a->wset_rows);
        VDX_CHECK(!a->bias && !a->bias2 && !a->residual && !geglu && a->mode == VDX_GEMM_PLAIN && a->ksplit <= 1,
                  "gemm: weight sets take no bias / bias2 / residual / GEGLU / split-K (fold them into wset_bias)");
        if (!ws_family && whole) ws_family = vdx_gemm_ws_family(p, a->mode, geglu);      // (also below the automatic row threshold)
        VDX_CHECK(ws_family == 1 || ws_family == 2 || ws_family == 4, "gemm: weight sets run on the weights-stationary kernels (K = 320 / 640, whole products)");
        VDX_CHECK(a->wset_rows % 64 == 0, "gemm: wset_rows must be a multiple of 64");
    }
    switch (a->mode) {
        case VDX_GEMM_PLAIN:
            break;
        case VDX_GEMM_CONV3X3:
            VDX_CHECK(!geglu, "gemm: GEGLU epilogue is plain-mode only");
            VDX_CHECK(a->c2 == 0, "gemm: conv3x3 takes one source");
            VDX_CHECK(a->stride == 1 || a->stride == 2, "gemm: stride %d", a->stride);
            VDX_CHECK(a->h_in > 0 && a->w_in > 0 && a->h_out > 0 && a->w_out > 0, "gemm: conv geometry");
            VDX_CHECK(a->M % (a->h_out * a->w_out) == 0, "gemm: M=%d not a whole number of %dx%d images", a->M, a->h_out, a->w_out);
            {
                VDX_CHECK(a->upsample >= 0 && a->upsample <= 2 && (a->upsample == 0 || a->stride == 1), "gemm: upsample %d with stride %d", a->upsample, a->stride);
                VDX_CHECK(a->upsample != 2 || (a->h_out >= a->h_in && a->w_out >= a->w_in), "gemm: upsample-to-size target smaller than the source");
                const int he = p.h_up, we = p.w_up;
                VDX_CHECK(a->h_out == (he + 2 - 3) / a->stride + 1 && a->w_out == (we + 2 - 3) / a->stride + 1,
                          "gemm: conv output %dx%d inconsistent with input %dx%d stride %d", a->h_out, a->w_out, he, we, a->stride);
            }
            break;
        case VDX_GEMM_TCONV3:
            VDX_CHECK(!geglu, "gemm: GEGLU epilogue is plain-mode only");
            VDX_CHECK(a->c2 == 0, "gemm: tconv3 takes one source");
            VDX_CHECK(a->frames > 0 && a->hw > 0 && a->M % (a->frames * a->hw) == 0, "gemm: tconv geometry M=%d F=%d HW=%d", a->M, a->frames, a->hw);
            break;
        default:
            return vdx_fail("gemm: unknown mode %d", a->mode);
    }
    return 0;
}

extern "C" int vdx_gemm_f16(const vdx_gemm_args* a, vdx_stream_t stream) {
    GemmP p;
    bool geglu;
    int force, ws_family;
    if (const int rc = gemm_prepare(a, p, geglu, force, ws_family)) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (a->ksplit > 1) {
        const int nt320 = (p.N + 319) / 320;
        const long long tiles = (long long)((p.M - p.m_begin + 255) / 256) * nt320;
        VDX_CHECK(!geglu && (force == 0 || force == 2), "gemm: split-K runs the 256x320 tile without GEGLU");
        VDX_CHECK(a->ksplit <= 16 && (p.K >> 6) / a->ksplit >= 2, "gemm: ksplit %d leaves fewer than two K tiles per slice (K = %d)", a->ksplit, p.K);
        VDX_CHECK(nt320 * 320 * 4 <= p.N * 5, "gemm: split-K needs N = %d to fill 320-wide tiles", p.N);
        VDX_CHECK(a->workspace && (uintptr_t)a->workspace % 16 == 0, "gemm: split-K needs a 16-byte aligned workspace");
        VDX_CHECK(tiles * a->ksplit <= (1 << 20), "gemm: split-K grid too large");
        VDX_CHECK(a->workspace_bytes >= (size_t)tiles * a->ksplit * KSPLIT_SLAB_BYTES,
                  "gemm: split-K workspace of %zu bytes is smaller than %lld tiles x %d slices x %d bytes", a->workspace_bytes, tiles,
                  a->ksplit, (int)KSPLIT_SLAB_BYTES);
        // the split-K kernels are VAR = 1: their gather shifts by p.ups (0 | 1) and has no nearest-to-size source map
        VDX_CHECK(a->upsample != 2, "gemm: split-K does not take the upsample-to-size gather (upsample = 2)");
        switch (a->mode) {
            case VDX_GEMM_PLAIN: return launch_ksplit<0>(p, a->ksplit, (float*)a->workspace, st);
            case VDX_GEMM_CONV3X3: return launch_ksplit<1>(p, a->ksplit, (float*)a->workspace, st);
            default: return launch_ksplit<2>(p, a->ksplit, (float*)a->workspace, st);
        }
    }
    if (ws_family) return vdx_gemm_ws_launch(p, ws_family, geglu, st);
    if (geglu) return pick_tile<0, true>(p, force, st);
    switch (a->mode) {
        case VDX_GEMM_PLAIN: return pick_tile<0, false>(p, force, st);
        case VDX_GEMM_CONV3X3: return pick_tile<1, false>(p, force, st);
        default: return pick_tile<2, false>(p, force, st);
    }
}

extern "C" int vdx_gemm_plan(const vdx_gemm_args* a, int32_t* variant, int32_t* split_row) {
    VDX_CHECK(variant && split_row, "gemm_plan: null pointer");
    GemmP p;
    bool geglu;
    int force, ws_family;
    if (const int rc = gemm_prepare(a, p, geglu, force, ws_family)) return rc;
    *split_row = 0;
    if (ws_family) {
        *variant = 7;
        return 0;
    }
    *variant = force ? force : choose_tile(p.M - p.m_begin, p.N).v;
    if (!force) *split_row = choose_split(p.m_begin, p.M, p.N);
    return 0;
}

// Split-K plan for a whole product (row_begin = row_end = 0): *split_row / *ksplit / *workspace_bytes such that rows
// [0, split_row) as one ordinary call and rows [split_row, M) as one call with vdx_gemm_args.ksplit = *ksplit (and a
// workspace of that many bytes) are expected to be faster than vdx_gemm_plan's best; *ksplit = 0: no.
extern "C" int vdx_gemm_plan_ksplit(const vdx_gemm_args* a, int32_t* split_row, int32_t* ksplit, size_t* workspace_bytes) {
    VDX_CHECK(split_row && ksplit && workspace_bytes, "gemm_plan_ksplit: null pointer");
    GemmP p;
    bool geglu;
    int force, ws_family;
    if (const int rc = gemm_prepare(a, p, geglu, force, ws_family)) return rc;
    *split_row = 0; *ksplit = 0; *workspace_bytes = 0;
    if (ws_family || geglu || force || p.m_begin != 0 || p.M != a->M) return 0;
    if (a->upsample == 2) return 0;     // no split-K instantiation carries the nearest-to-size gather (vdx_gemm_f16 refuses it)
    const long long rows = p.M;
    const int nt320 = (p.N + 319) / 320, nk = p.K >> 6;
    if (!(nt320 * 320 * 4 <= p.N * 5 && rows >= 1024)) return 0;        // (choose_tile's `fits`: N fills 320-wide tiles)
    const long long t256 = ((rows + 255) / 256) * nt320, full = t256 / 256;
    if (full == 0 || t256 % 256 == 0) return 0;
    const long long mt_main = full * 256 / nt320;
    const long long split = mt_main * 256;
    if (split >= rows) return 0;
    const long long t = ((rows - split + 255) / 256) * nt320;          // tiles of the tail
    int S = (int)(128 / t);            // (t * S <= 128 slabs of 327 680 bytes: the workspace stays <= 42 MB)
    if (S > 8) S = 8;
    if (S > nk / 4) S = nk / 4;
    if (S < 2) return 0;
    // costs in HUNDREDTHS of a 256x320 tile time (choose_tile counts tenths).  Slab traffic + reduction + its launch:
    // ~ 25 us at 128 slabs (measured: 2048 x 1280 x 11520 as 32 tiles x 8 slices 90 us for 54 us of slices), against a
    // tile time of 0.0375 us per unit of K (fitted at 1.1 PFLOP/s); a second launch ~ a tenth of a tile.
    const long long main_c = 100 * ((mt_main * nt320 + 255) / 256);
    const long long cost = main_c + (100 + S - 1) / S + (5 + 20 * t * S / 128) * 100 * 1000 / (37 * p.K) + 10;
    // what vdx_gemm_plan would do: the whole product on its best tile, or whole rounds + a tail on small tiles
    long long best = 10 * choose_tile(rows, p.N).cost;
    const long long two = main_c + 10 * choose_tile(rows - split, p.N).cost + 10;
    if (two < best) best = two;
    if (cost * 100 > best * 97) return 0;
    *split_row = (int32_t)split;
    *ksplit = S;
    *workspace_bytes = (size_t)t * S * KSPLIT_SLAB_BYTES;
    return 0;
}

// Lab variants of this translation unit (phase stamps, ablations: timing only, some give WRONG results) are compiled in only
// under the macros below; a library that carries one says so through vdx_build_flags() and vdx/_lib.py refuses to load it
// as the product.
extern "C" int vdx_lab_gemm(void) {
#if defined(VDX_STAMPS) || (defined(VDX_ABL) && VDX_ABL) || defined(VDX_GEMM_PLAIN_LOOP)
    return 1;
#else
    return 0;
#endif
}
