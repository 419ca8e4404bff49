// gemm_ring.hip — the large-shape GEMM / implicit-GEMM kernel: 256x320 block tile, 8 waves
// (4 x 2, 64x160 per wave = 40 accumulators of v_mfma_f32_16x16x32_f16), K-step 32, and a
// FOUR-stage LDS ring fed by LDS-DMA with COUNTED vmcnt waits and raw s_barrier:
//
//   prologue: stages 0,1,2 in flight
//   stage s : s_waitcnt vmcnt(2 x my DMA count)   -> my part of stage s has landed
//             s_barrier                           -> everyone's part landed; everyone is past
//                                                    the MFMAs of stage s-1
//             issue DMA of stage s+3 into the ring slot stage s-1 just vacated
//             14 ds_read_b128 + 40 MFMA on stage s
//
// so two stages of loads stay in flight across every barrier and the matrix pipe never waits for
// a vmcnt(0) drain (the 2-stage kernel in gemm.hip does, once per K tile).  One block per CU
// (144 KB of LDS, 8 waves = 2 per SIMD).  Same operand conventions, gather modes, swizzle idea
// (64-byte rows: slot s of row r holds chunk s ^ 3*((r>>2)&1), conflict-free ds_read_b128) and
// epilogue as gemm.hip.
#include "gemm_common.h"

namespace {

constexpr int BN = 320;
constexpr int TN = 10;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// WMB = waves along M, RPW = rows per wave (a wave's tile is RPW x 160), NS = ring stages.
//   <4, 64, 4>: 256x320 tile, 8 waves, 144 KB LDS, one block per CU, two stages of DMA in flight across each barrier.
//   <2, 64, 2>: 128x320 tile, 4 waves, 56 KB LDS, TWO blocks per CU — one block's prologue/epilogue (short-K Linear
//               layers, the erf-heavy GEGLU epilogue) overlaps the other's MFMA phase.
//   <4, 32, 4>: 128x320 tile, EIGHT waves of 32x160, 112 KB LDS, four-stage ring: the level-3 shapes (M = 6912:
//               216 tiles of 128x320 on 256 CUs, one block per CU whatever the kernel) get two waves per SIMD and
//               two stages of DMA in flight instead of one wave per SIMD behind a vmcnt(0) drain.
template <int WMB, int RPW, int NS, int MODE, bool GEGLU>
__global__ __launch_bounds__(WMB * 128) void gemm_ring_kernel(const GemmP p) {
    constexpr int BM = WMB * RPW, NW = WMB * 2, TM = RPW / 16;
    constexpr int NA = BM / 16 / NW;               // A pieces per wave (2, or 1 for the 32-row waves)
    constexpr int STAGE = (BM + BN) * 64;          // 32 K-elements (64 B) per row
    constexpr int NB = (20 + NW - 1) / NW;         // B pieces per wave (the last may be absent)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    int mt_, nt_;
    gemm_tile_of(bid, p.ntm, p.ntn, mt_, nt_);
    const int m0 = p.m_begin + mt_ * BM, n0 = nt_ * BN;
    float2* gelu = (float2*)(smem + NS * STAGE);   // GEGLU: table behind the ring (the stage barriers publish it)
    if (GEGLU) gelu_tab_init(gelu, tid, NW * 64);

    // ---- DMA roles.  A stage is BM/16 + 20 wave-instructions of 1 KiB (16 rows x 64 B).  Wave w
    // issues A pieces w, w+NW and B pieces w, w+NW, ... (< 20).
    const int drow = lane >> 2, dslot = lane & 3;
    const int csw = (dslot ^ (3 * ((drow >> 2) & 1))) * 8;      // element offset of my data chunk
    size_t a_off[NA], a_off2[NA];
    int a_y[NA], a_x[NA];
    bool a_ok[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int m = m0 + 16 * (wave + NW * i) + drow;
        a_ok[i] = m < p.M;
        const int mm = a_ok[i] ? m : 0;
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
            a_off[i] = (size_t)n * p.h_in * p.w_in;
            a_off2[i] = 0;
        } else {
            a_y[i] = (mm / p.hw) % p.frames;
            a_x[i] = 0;
            a_off[i] = (size_t)mm;
            a_off2[i] = 0;
        }
    }
    const f16* zp = (const f16*)g_zero_page;
    const bool has_last = wave + NW * (NB - 1) < 20;             // wave-uniform
    const f16* b_src[NB];
    int b_step[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int pr = 16 * (wave + NW * i) + drow;              // physical LDS row (< 320 when used)
        const int r = (pr & ~31) | (((pr >> 2) & 3) << 3) | (((pr >> 4) & 1) << 2) | (pr & 3);
        const bool ok = pr < BN && n0 + r < p.N;
        b_src[i] = ok ? p.w + (size_t)(n0 + r) * p.K + csw : zp;
        b_step[i] = ok ? 32 : 0;
    }

    const int ct = p.c1 + p.c2;
    int tap = 0, kc = 0;
    auto issue = [&](int slot) {
        char* sa = smem + slot * STAGE + wave * 1024;
        char* sb = smem + slot * STAGE + BM * 64 + wave * 1024;
#pragma unroll
        for (int i = 0; i < NB - 1; ++i) {
            __builtin_amdgcn_global_load_lds((gptr_t)b_src[i], (lptr_t)(sb + i * NW * 1024), 16, 0, 0);
            b_src[i] += b_step[i];
        }
        if (has_last) {
            __builtin_amdgcn_global_load_lds((gptr_t)b_src[NB - 1], (lptr_t)(sb + (NB - 1) * NW * 1024), 16, 0, 0);
            b_src[NB - 1] += b_step[NB - 1];
        }
        if (MODE == 0) {
            const bool first = kc < p.c1;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const f16* src = first ? p.a + a_off[i] + kc : p.a2 + a_off2[i] + (kc - p.c1);
                __builtin_amdgcn_global_load_lds((gptr_t)(a_ok[i] ? src + csw : zp), (lptr_t)(sa + i * NW * 1024), 16, 0, 0);
            }
        } else if (MODE == 1) {
            const int ky = tap / 3, kx = tap - ky * 3;
            const int hlim = p.h_up, wlim = p.w_up;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                int y = a_y[i] + ky, x = a_x[i] + kx;
                const bool ok = a_ok[i] && (unsigned)y < (unsigned)hlim && (unsigned)x < (unsigned)wlim;
                y >>= p.ups;                  // 0 | 1 (nearest-to-size, ups == 2, runs the tiled kernels' own instantiation: gemm.hip pick_tile)
                x >>= p.ups;
                const f16* src = p.a + (a_off[i] + (size_t)(y * p.w_in + x)) * p.lda + kc + csw;
                __builtin_amdgcn_global_load_lds((gptr_t)(ok ? src : zp), (lptr_t)(sa + i * NW * 1024), 16, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int f = a_y[i] + tap - 1;
                const bool ok = a_ok[i] && (unsigned)f < (unsigned)p.frames;
                const f16* src = p.a + (a_off[i] + (size_t)((tap - 1) * p.hw)) * p.lda + kc + csw;
                __builtin_amdgcn_global_load_lds((gptr_t)(ok ? src : zp), (lptr_t)(sa + i * NW * 1024), 16, 0, 0);
            }
        }
        // K order of the gathers: (64-channel slice, tap, channel) as in gemm.hip; a K-step covers
        // half a slice, so the half index toggles fastest.
        if (MODE == 0) {
            kc += 32;
        } else {
            kc ^= 32;
            if ((kc & 32) == 0 && ++tap == (MODE == 1 ? 9 : 3)) {
                tap = 0;
                kc += 64;
            }
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = p.K >> 5;
    const int frow = lane & 15, fq = lane >> 4;
    const int coff = (fq ^ (3 * ((frow >> 2) & 1))) << 4;        // my 16-byte slot inside a 64-byte row
    const int a_rd = (wm * RPW + frow) * 64 + coff;
    const int b_rd = BM * 64 + (wn * 160 + frow) * 64 + coff;

    int issued = 0;
    for (; issued < NS - 1 && issued < nk; ++issued) issue(issued);
    for (int s = 0; s < nk; ++s) {
        // my DMA of stage s is complete once at most `ahead` younger stages of mine are outstanding
        const int ahead = min(issued - 1 - s, NS - 2);
        if (NS == 2) {
            wait_vmcnt<0>();
        } else if (has_last) {
            if (ahead == 2) wait_vmcnt<2 * (NB + NA)>(); else if (ahead == 1) wait_vmcnt<NB + NA>(); else wait_vmcnt<0>();
        } else {
            if (ahead == 2) wait_vmcnt<2 * (NB - 1 + NA)>(); else if (ahead == 1) wait_vmcnt<NB - 1 + NA>(); else wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
        if (issued < nk) {
            issue(issued & (NS - 1));
            ++issued;
        }
        const char* st = smem + (s & (NS - 1)) * STAGE;
        f16x8 af[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *(const f16x8*)(st + a_rd + i * 1024);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            f16x8 bf[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) bf[j] = *(const f16x8*)(st + b_rd + (g * 5 + j) * 1024);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 5; ++j)
                    acc[i][g * 5 + j] =
                        __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][g * 5 + j], 0, 0, 0);
        }
    }
    gemm_epilogue<TM, TN, GEGLU>(p, acc, m0 + wm * RPW, n0 + wn * 160, frow, fq, gelu);
}

template <int WMB, int RPW, int NS, int MODE, bool GEGLU>
int launch_ring(const GemmP& p, hipStream_t st) {
    constexpr int BM = WMB * RPW;
    constexpr int lds = NS * (BM + BN) * 64 + (GEGLU ? GELU_TAB_BYTES : 0);
    auto kern = gemm_ring_kernel<WMB, RPW, NS, MODE, GEGLU>;
    // one-time LDS opt-in; a function-local static is initialised exactly once even under concurrent callers
    static const hipError_t attr_rc = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (attr_rc != hipSuccess) return vdx_fail("gemm: cannot reserve %d bytes of LDS", lds);
    GemmP q = p;
    q.ntn = (p.N + BN - 1) / BN;
    const int ntm = (p.M - p.m_begin + BM - 1) / BM;
    q.ntm = ntm;
    hipLaunchKernelGGL(kern, dim3(ntm * q.ntn), dim3(WMB * 128), lds, st, q);
    return vdx_launch_status("vdx_gemm_f16");
}

}  // namespace

template <int WMB, int RPW, int NS>
static int dispatch(const GemmP& p, int mode, bool geglu, hipStream_t st) {
    if (geglu) return launch_ring<WMB, RPW, NS, 0, true>(p, st);
    switch (mode) {
        case VDX_GEMM_PLAIN: return launch_ring<WMB, RPW, NS, 0, false>(p, st);
        case VDX_GEMM_CONV3X3: return launch_ring<WMB, RPW, NS, 1, false>(p, st);
        default: return launch_ring<WMB, RPW, NS, 2, false>(p, st);
    }
}

int vdx_gemm_ring_launch(const GemmP& p, int mode, bool geglu, int variant, hipStream_t st) {
    if (variant == 0) return dispatch<4, 64, 4>(p, mode, geglu, st);
    if (variant == 1) return dispatch<2, 64, 2>(p, mode, geglu, st);
    return dispatch<4, 32, 4>(p, mode, geglu, st);
}
