// gemm_ws.hip — weights-stationary streaming GEMM for the level-0 Linear layers (K = 320).
//
//   out[M][N] = epilogue( A[M][320] . W[N][320]^T ),   N a multiple of 320, M a multiple of 64
//
// Why a second kernel: at K = 320 the tiled kernel (gemm.hip) runs five K tiles per output tile and then
// pays a DMA prologue (~3 us), an epilogue (5-15 us) and a block turn-around (1-2 us) that nothing
// overlaps with — one block per CU owns all the LDS — so the matrix pipe is busy ~30 % of the time
// (tools/gemm_stamps.py).  Every output tile also re-streams its 205 KB weight panel from L2.  Here the
// weight panel never moves:
//
//   * a block owns a 320-column panel of W for its whole life; each of its 10 waves keeps its
//     32 columns x 320 K slice in REGISTERS (80 VGPRs, MFMA A-operand layout), loaded once;
//   * activations stream through a 3-stage LDS ring in chunks of 64 rows (40 KB, LDS-DMA, counted
//     vmcnt waits, one barrier per chunk); per chunk a wave does 40 ds_read_b128 + 80 MFMA
//     (v_mfma_f32_16x16x32_f16) and stores its 64 x 32 outputs (16-byte stores, fused bias / residual /
//     GEGLU) while the other waves of its SIMD keep the matrix pipe busy;
//   * the N/320 blocks that walk the same row range sit on ONE XCD and advance together, so a chunk
//     comes from HBM once and from that XCD's L2 for the other panels.
//
// LDS traffic per MFMA is the same as in the tiled kernel; L2 -> LDS traffic drops from (A + W) per tile to
// A only, and there is no per-tile prologue/epilogue bubble: the kernel is MFMA-bound for N >= 640 and
// HBM-bound (A + out) for N = 320.
#include "gemm_common.h"

namespace {

constexpr int WS_K = 320, WS_BN = 320, WS_ROWS = 64;
constexpr int WS_NW = 10;                       // waves per block, 32 output columns each
constexpr int WS_KS = WS_K / 32;                // MFMA K steps
constexpr int WS_STAGE = WS_ROWS * WS_K * 2;    // 40 KB: 5 sub-tiles of [64 rows][128 B]
constexpr int WS_NS = 3;                        // ring stages
constexpr int WS_PIECES = WS_STAGE / 1024 / WS_NW;   // DMA instructions per wave and chunk (4)
constexpr int WS_STORES = 4;                    // global stores per wave and chunk (one per 16-row group)
static_assert(WS_PIECES * WS_NW * 1024 == WS_STAGE, "chunk must split evenly over the waves");

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct WsP {
    GemmP g;
    int nt;         // 320-column panels (N / 320)
    int gpx;        // row groups per XCD (32 / nt)
    int nch;        // 64-row chunks in all (M / 64)
};

template <bool GEGLU, bool RES>
__global__ __launch_bounds__(WS_NW * 64) void gemm_ws_kernel(const WsP q) {
    const GemmP& p = q.g;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;

    // block -> (XCD, panel, row group): blocks b and b+8 share an XCD; the nt panels of one row group are
    // neighbours there, so they read the same activation chunks at about the same time (L2 hits)
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int panel = idx % q.nt, group = xcd * q.gpx + idx / q.nt, groups = 8 * q.gpx;
    const int c0 = (int)((long long)q.nch * group / groups), c1 = (int)((long long)q.nch * (group + 1) / groups);
    const int nch = c1 - c0;
    if (nch <= 0) return;
    const int n0 = panel * WS_BN + wave * 32;

    // ---- activation staging: chunk = 5 sub-tiles (64 K-elements each) of [64 rows][128 B]; DMA instruction
    // i of wave w fills sub-tile w>>1, rows 8*(4*(w&1)+i) .. +7; lane l lands at row +(l>>3), slot l&7, and
    // slot s of row r must hold data chunk s ^ (r & 7) (conflict-free ds_read_b128, as in gemm.hip)
    const int drow = 32 * (wave & 1) + (lane >> 3);
    const f16* a_src = p.a + (size_t)drow * p.lda + 64 * (wave >> 1) + ((lane & 7) ^ (lane >> 3)) * 8;
    const int d_off = (wave >> 1) * 8192 + (wave & 1) * 4096;
    auto issue = [&](int k) {   // chunk c0 + k -> ring slot k % 3
        const f16* src = a_src + (size_t)(c0 + k) * WS_ROWS * p.lda;
        char* dst = smem + (k % WS_NS) * WS_STAGE + d_off;
#pragma unroll
        for (int i = 0; i < WS_PIECES; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + (size_t)(8 * i) * p.lda), (lptr_t)(dst + i * 1024), 16, 0, 0);
    };
    issue(0);
    if (nch > 1) issue(1);

    // ---- this wave's weight slice, MFMA A-operand layout.  MFMA row i of 16-column tile j carries
    // weight row n0 + 8*(i>>2) + 4*j + (i&3): the lane then owns 8 CONSECUTIVE output columns
    // (n0 + 8*fq .. +7) of row frow -> 16-byte stores.
    f16x8 wf[WS_KS][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const f16* wrow = p.w + (size_t)(n0 + 8 * (frow >> 2) + 4 * j + (frow & 3)) * WS_K + 8 * fq;
#pragma unroll
        for (int s = 0; s < WS_KS; ++s) wf[s][j] = *(const f16x8*)(wrow + 32 * s);
    }
    f16x8 bv;
#pragma unroll
    for (int j = 0; j < 8; ++j) bv[j] = (f16)0.f;
    if (p.bias) bv = *(const f16x8*)(p.bias + n0 + 8 * fq);

    const int a_rd = frow * 128;   // + sub-tile, + 16-row group, + swizzled slot
    for (int k = 0; k < nch; ++k) {
        // chunk k has landed once at most the operations issued after its DMA are outstanding:
        // the stores of chunks k-2 and k-1 and the DMA of chunk k+1 (vector-memory operations retire in order)
        const int younger = (k + 1 < nch ? WS_PIECES : 0) + (k >= 1 ? WS_STORES : 0) + (k >= 2 ? WS_STORES : 0);
        if (younger >= WS_PIECES + 2 * WS_STORES) wait_vmcnt<WS_PIECES + 2 * WS_STORES>();
        else if (younger >= 2 * WS_STORES) wait_vmcnt<2 * WS_STORES>();
        else if (younger >= WS_PIECES) wait_vmcnt<(WS_PIECES < WS_STORES ? WS_PIECES : WS_STORES)>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();     // everyone's part of chunk k landed; everyone is past chunk k-1
        if (k + 2 < nch) issue(k + 2);    // into the ring slot chunk k-1 just vacated

        const char* st = smem + (k % WS_NS) * WS_STAGE + a_rd;
        f32x4 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][0] = acc[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < WS_KS; ++s) {
            const int c = (s & 1) * 4 + fq;
            f16x8 af[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)   // row = 16*i + frow, so row & 7 == frow & 7
                af[i] = *(const f16x8*)(st + (s >> 1) * 8192 + i * 2048 + ((c ^ (frow & 7)) << 4));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[s][0], af[i], acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[s][1], af[i], acc[i][1], 0, 0, 0);
            }
        }

        // ---- epilogue of this chunk: exactly WS_STORES stores per wave (the wait above counts them)
        const size_t m = (size_t)(c0 + k) * WS_ROWS + frow;
        f16x8 rv[4];
        if (RES) {
#pragma unroll
            for (int i = 0; i < 4; ++i) rv[i] = *(const f16x8*)(p.res + (m + 16 * i) * p.ldr + n0 + 8 * fq);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = acc[i][0][j] + (float)bv[j];
                v[4 + j] = acc[i][1][j] + (float)bv[4 + j];
            }
            if (GEGLU) {
                f16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (f16)(v[j] * gelu_erf_f(v[4 + j]));
                *(f16x4*)(p.out + (m + 16 * i) * p.ldo + ((n0 + 8 * fq) >> 1)) = o;
            } else {
                f16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (f16)(RES ? v[j] + (float)rv[i][j] : v[j]);
                *(f16x8*)(p.out + (m + 16 * i) * p.ldo + n0 + 8 * fq) = o;
            }
        }
    }
}

template <bool GEGLU, bool RES>
int launch_ws(const GemmP& p, hipStream_t st) {
    constexpr int lds = WS_NS * WS_STAGE;
    auto kern = gemm_ws_kernel<GEGLU, RES>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return vdx_fail("gemm_ws: cannot reserve %d bytes of LDS", lds);
        attr_set = true;
    }
    WsP q;
    q.g = p;
    q.nt = p.N / WS_BN;
    q.gpx = 32 / q.nt;                 // 256 CUs = 8 XCDs x 32: one block per CU
    q.nch = p.M / WS_ROWS;
    hipLaunchKernelGGL(kern, dim3(8 * q.gpx * q.nt), dim3(WS_NW * 64), lds, st, q);
    return vdx_launch_status("vdx_gemm_f16 (weights-stationary)");
}

}  // namespace

// usable(): plain single-source rows, K = 320, whole 320-column panels (at most 32: one XCD holds all
// panels of a row group), whole 64-row chunks, no per-row-block bias
bool vdx_gemm_ws_usable(const GemmP& p, int mode) {
    return mode == VDX_GEMM_PLAIN && p.K == WS_K && p.c2 == 0 && p.N % WS_BN == 0 && p.N / WS_BN <= 32 &&
           p.M % WS_ROWS == 0 && p.bias2 == nullptr;
}

int vdx_gemm_ws_launch(const GemmP& p, bool geglu, hipStream_t st) {
    if (geglu) return launch_ws<true, false>(p, st);
    return p.res ? launch_ws<false, true>(p, st) : launch_ws<false, false>(p, st);
}
