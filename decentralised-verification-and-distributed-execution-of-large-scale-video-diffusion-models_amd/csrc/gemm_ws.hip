// gemm_ws.hip — weights-stationary streaming GEMM for the Linear layers with a short K (320 / 512 / 640:
// levels 0 and 1 and transformer_in of the 3D-UNet).
//
//   out[M][N] = epilogue( A[M][K] . W[N][K]^T ),   N a multiple of 32, M a multiple of the chunk height
//
// Why a second kernel: at K = 320 the tiled kernel (gemm.hip) runs five K tiles per output tile and then
// pays a DMA prologue (~3 us), an epilogue (5-15 us) and a block turn-around (1-2 us) that nothing
// overlaps with — one block per CU owns all the LDS — so the matrix pipe is busy ~30 % of the time
// (tools/gemm_stamps.py).  Every output tile also re-streams its weight panel from L2.  Here the weight
// panel never moves:
//
//   * a block owns a panel of 32*NW columns of W for its whole life; each of its NW waves keeps its
//     32 columns x K slice in REGISTERS (K/4 VGPRs, MFMA A-operand layout), loaded once;
//   * activations stream through a 3-stage LDS ring in chunks of ROWS rows (LDS-DMA, counted vmcnt waits,
//     one barrier per chunk); per chunk a wave does ROWS/16 ds_read_b128 and ROWS/8 MFMA
//     (v_mfma_f32_16x16x32_f16) per K step and stores its ROWS x 32 outputs (16-byte stores, fused bias /
//     residual / GEGLU);
//   * PIPE variants (8 waves, 256 registers each) keep TWO accumulator sets: the epilogue of chunk k-1
//     (for GEGLU ~270 VALU instructions per lane: as long as the chunk's MFMAs) is interleaved with the
//     MFMAs of chunk k instead of running while the matrix pipe idles behind the chunk barrier;
//   * the blocks that walk the same row range are neighbours on one XCD and advance together, so a chunk
//     comes from HBM once and from that XCD's L2 for the other panels.
//
// LDS traffic per MFMA is the same as in the tiled kernel; L2 -> LDS traffic drops from (A + W) per tile to
// A only, and there is no per-tile prologue/epilogue bubble.
#include "gemm_common.h"

#ifdef VDX_STAMPS   // diagnostic build only: per-block cycle totals of wave 0 (wait+barrier, DMA issue, compute+epilogue)
static __device__ unsigned long long g_ws_stamps[256 * 4];
extern "C" int vdx_debug_read_ws_stamps(void* dst) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_ws_stamps), sizeof(unsigned long long) * 256 * 4) == hipSuccess ? 0 : -1;
}
#define WS_T() __builtin_amdgcn_s_memtime()
#endif

namespace {

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct WsP {
    GemmP g;
    int nt;         // column panels
    int groups;     // row groups (256 / nt): blocks of a group walk the same chunks
    int nch;        // chunks in all (M / ROWS)
    int per_xcd;    // blocks per XCD (grid / 8): 32 when the grid fills the 256 CUs
};

// WSET: one weight set per p.wset_rows rows (a GroupNorm folded into this Linear, norm.hip gn_fold_kernel): the register
// slice is reloaded when the walk enters the next set (a multiple of the chunk height), and the set's fp32 bias is the
// initial accumulator of its chunks.
template <int K, int NW, int ROWS, bool GEGLU, bool RES, bool PIPE, bool WSET = false>
struct Ws {
    static constexpr int KS = K / 32;               // MFMA K steps
    static constexpr int MT = ROWS / 16;            // 16-row groups per chunk
    static constexpr int SUBT = ROWS * 128;         // one sub-tile: [ROWS rows][64 K-elements]
    static constexpr int STAGE = ROWS * K * 2;      // K/64 sub-tiles
    // Ring stages: FOUR for K = 640 where they fit beside the GELU table / the weight sets' bias page (32 rows x 640: 4 x 40 KB
    // = the CU's 160 KB exactly), else three (K = 320 with its 64-row chunks measured 2-4 % SLOWER on four: profiles/r06_ws_ring.md).  The kernel is bound by HBM LATENCY, not bandwidth: the blocks of a
    // row group fetch the same chunk, so the chip has groups x (NS - 1) x 40 KB of reads in flight — 6.8 MB with three stages,
    // which at ~2 us of loaded latency is the ~3 TB/s these launches measured (round 5 PMC: 2.2-3.0 TB/s, 24-40 % MFMA busy);
    // a fourth stage puts half as much again in flight (round 6, profiles/r06_ws_ring.md).
#ifdef VDX_WS_NS3
    static constexpr int NS = 3;
#else
    static constexpr int NS = (K == 640 && 4 * STAGE + (GEGLU ? GELU_TAB_BYTES : 0) + (WSET ? NW * 64 * 32 : 0) <= 160 * 1024) ? 4 : 3;
#endif
    static constexpr int PF = NS - 1;               // chunks requested ahead of the one being computed
    static constexpr int PPW = STAGE / 1024 / NW;   // DMA instructions per wave and chunk
    static constexpr int RB = ROWS / 8;             // 8-row DMA blocks per sub-tile
    static_assert(PPW * NW * 1024 == STAGE, "chunk must split evenly over the waves");
    static_assert(NS * STAGE <= 160 * 1024, "ring exceeds the LDS");
    typedef f32x4 Acc[MT][2];

    const GemmP& p;
    char* smem;
    const float2* gelu;          // GEGLU: table of the normal CDF behind the ring (published by the first chunk barrier)
    int lane, wave, frow, fq, n0, c0, nch;
    bool active;                 // this wave's 32 columns exist (last panel of an N that is no panel multiple)
    const f16* a_src;            // this lane's DMA source for row (lane>>3) of 8-row block 0, sub-tile 0
    f16x8 wf[KS][2];
    f16x8 bv;
    int set_left;                // WSET: chunks of the current weight set still to walk (block-uniform)
    int cur_set;
#ifdef VDX_STAMPS
    unsigned long long t_bar = 0, t_iss = 0, t_wait = 0, t_last = 0;
#endif

    __device__ __forceinline__ Ws(const GemmP& p_, char* smem_) : p(p_), smem(smem_) {}

    // chunk c0 + k -> ring slot k % NS.  DMA instruction i of wave w is piece w*PPW + i: sub-tile
    // piece / RB, rows 8*(piece % RB) .. +7; lane l lands at row +(l>>3), slot l&7, and slot s of row r
    // must hold data chunk s ^ (r & 7) (conflict-free ds_read_b128, as in gemm.hip)
    __device__ __forceinline__ void issue(int k) {
        const f16* src = a_src + (size_t)(c0 + k) * ROWS * p.lda;
        char* dst = smem + (k % NS) * STAGE;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int piece = wave * PPW + i, q = piece / RB, rb = piece % RB;
            __builtin_amdgcn_global_load_lds((gptr_t)(src + (size_t)(8 * rb) * p.lda + 64 * q),
                                             (lptr_t)(dst + q * SUBT + rb * 1024), 16, 0, 0);
        }
    }
    // Chunk k has landed once at most the operations issued after its DMA are outstanding (vector-memory
    // operations retire in order): the DMA of chunks k+1 .. k+PF-1 and the stores of the PF epilogues issued since
    // (chunks k-PF .. k-1; with PIPE, whose epilogue runs one chunk later: k-PF-1 .. k-2).  Residual loads are
    // not counted: they have been consumed, and counting fewer only waits longer.
    template <int N>
    __device__ __forceinline__ void wait_le(int n) {      // s_waitcnt vmcnt(n) for a run-time n in [0, N]: the count is an immediate
        if constexpr (N == 0) wait_vmcnt<0>();
        else {
            if (n >= N) wait_vmcnt<N>();
            else wait_le<N - 1>(n);
        }
    }
    __device__ __forceinline__ void sync(int k) {
        constexpr int LAG = PIPE ? 1 : 0;
        int nd = 0, ns = 0;                                  // DMA batches / epilogues issued after chunk k's DMA (block-uniform but for `active`)
#pragma unroll
        for (int j = 1; j < PF; ++j) nd += (k + j < nch);
#pragma unroll
        for (int j = 1; j <= PF; ++j) ns += (active && k >= j + LAG);     // idle waves store nothing
        // RES + PIPE: every step since has also requested its chunk's residual rows (MT loads, steps k-PF .. k-1); the
        // newest of them are still in flight here and must be allowed to stay so
        int nr = 0;
        if (RES && PIPE) {
#pragma unroll
            for (int j = 1; j <= PF; ++j) nr += (active && k >= j);
        }
        // (a wave-uniform value: readfirstlane keeps the chain of compares scalar)
        wait_le<(PF - 1) * PPW + PF * MT * ((RES && PIPE) ? 2 : 1)>(__builtin_amdgcn_readfirstlane(nd * PPW + (ns + nr) * MT));
#ifdef VDX_STAMPS
        const unsigned long long t0 = WS_T();
#endif
        __builtin_amdgcn_s_barrier();     // everyone's part of chunk k landed; everyone is past chunk k-1
#ifdef VDX_STAMPS
        const unsigned long long t1 = WS_T();
#endif
        if (k + PF < nch) issue(k + PF);  // into the ring slot chunk k-1 just vacated
#ifdef VDX_STAMPS
        const unsigned long long t2 = WS_T();
        t_bar += t1 - t0;
        t_iss += t2 - t1;
        t_wait += t0 - t_last;    // (includes the compute of the previous chunk: subtract t_cmp)
        t_last = t2;
#endif
    }
    __device__ __forceinline__ void zero(Acc& acc) {
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[i][0] = acc[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __device__ __forceinline__ void load_af(const char* st, int s, f16x8 (&af)[MT]) {
        const int c = (s & 1) * 4 + fq;
#pragma unroll
        for (int i = 0; i < MT; ++i)   // row = 16*i + frow, so row & 7 == frow & 7
            af[i] = *(const f16x8*)(st + (s >> 1) * SUBT + i * 2048 + ((c ^ (frow & 7)) << 4));
    }
    // K step 0 starts from the constant 0 (an inline operand of the MFMA): no accumulator clearing
    __device__ __forceinline__ void mfma_step(int s, const f16x8 (&af)[MT], Acc& acc) {
        f32x4 z0 = (f32x4){0.f, 0.f, 0.f, 0.f}, z1 = z0;
        if (WSET && s == 0) {       // the set's bias = the initial accumulators; it lives in LDS (8 more registers held across the
            int o = bias_off();     // chunk loop spill the K = 640 kernel), re-read per chunk through an address the optimiser cannot hoist
            asm volatile("" : "+v"(o));
            z0 = *(const f32x4*)(smem + o);
            z1 = *(const f32x4*)(smem + o + 16);
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[s][0], af[i], s == 0 ? z0 : acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[s][1], af[i], s == 0 ? z1 : acc[i][1], 0, 0, 0);
        }
    }
    // this wave's weight slice, MFMA A-operand layout.  MFMA row i of 16-column tile j carries weight
    // row n0 + 8*(i>>2) + 4*j + (i&3): the lane then owns 8 CONSECUTIVE output columns
    // (n0 + 8*fq .. +7) of row frow -> 16-byte stores.
    __device__ __forceinline__ int bias_off() const { return NS * STAGE + (int)threadIdx.x * 32; }      // WSET: this lane's 32 bytes of LDS, behind the ring
    __device__ __forceinline__ void load_weights(int set) {
        const int nw = active ? n0 : 0;
        const f16* wbase = p.w + (WSET ? (size_t)set * p.N * K : 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const f16* wrow = wbase + (size_t)(nw + 8 * (frow >> 2) + 4 * j + (frow & 3)) * K + 8 * fq;
#pragma unroll
            for (int s = 0; s < KS; ++s) wf[s][j] = *(const f16x8*)(wrow + 32 * s);
        }
        if (WSET) {                  // this lane's 8 bias values -> its own 32 bytes of LDS (read back by the same lane only)
            const float* b = p.wset_bias + (size_t)set * p.N + nw + 8 * fq;
            *(f32x4*)(smem + bias_off()) = *(const f32x4*)b;
            *(f32x4*)(smem + bias_off() + 16) = *(const f32x4*)(b + 4);
        }
    }
    __device__ __forceinline__ void load_res(int k, f16x8 (&rv)[MT]) {
        const size_t m = (size_t)(c0 + k) * ROWS + frow;
#pragma unroll
        for (int i = 0; i < MT; ++i) rv[i] = *(const f16x8*)(p.res + (m + 16 * i) * p.ldr + n0 + 8 * fq);
    }
    // row group i of chunk k: exactly one store (the waits in sync() count them)
    __device__ __forceinline__ void store_group(int k, int i, const Acc& acc, const f16x8 (&rv)[MT]) {
        const size_t m = (size_t)(c0 + k) * ROWS + frow + 16 * i;
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = acc[i][0][j] + (float)bv[j];
            v[4 + j] = acc[i][1][j] + (float)bv[4 + j];
        }
        if (GEGLU) {
            f16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (f16)(v[j] * gelu_tab(v[4 + j], gelu));
            *(f16x4*)(p.out + m * p.ldo + ((n0 + 8 * fq) >> 1)) = o;
        } else {
            f16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (f16)(RES ? v[j] + (float)rv[i][j] : v[j]);
            *(f16x8*)(p.out + m * p.ldo + n0 + 8 * fq) = o;
        }
    }
    // MFMAs of chunk k into `cur`, with the epilogue of chunk k-1 (from `prev`) spread between them
    // rv_next (PIPE only): the residual set this step LOADS (chunk k's rows, for the epilogue that runs in the next step);
    // the epilogue of chunk k-1 inside this step uses `rv`, the other set, requested one whole step ago.  Round 5
    // requested chunk k-1's residual at the top of the step that stores it: ~1 450 of the chunk's 4 540 cycles were that
    // load's latency (tools/ws_stamps.py: 4 542 against 3 090 without a residual).
    typedef f16x8 Res[MT];
    template <bool HAS_PREV>
    __device__ __forceinline__ void step(int k, Acc& cur, const Acc& prev, Res& rv_next, const Res& rv) {
        const char* st = smem + (k % NS) * STAGE + frow * 128;
        if (!active) zero(cur);
        if (!active) return;              // (idle waves of a last panel: no loads either — their columns do not exist)
        if (RES && PIPE) load_res(k, rv_next);
        if (WSET) {                                  // (block-uniform: every wave of the block walks the same chunks)
            if (set_left == 0) {
                cur_set += 1;
                set_left = p.wset_rows / ROWS;
                load_weights(cur_set);
            }
            set_left -= 1;
        }
        // (reading the fragments of K step s+1 before the MFMAs of step s was measured: no gain — the other
        // wave(s) of the SIMD already cover the LDS latency)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            f16x8 af[MT];
            load_af(st, s, af);
            mfma_step(s, af, cur);
            if (HAS_PREV) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    if (s + 1 == KS * (i + 1) / MT) store_group(k - 1, i, prev, rv);
            }
        }
    }
    __device__ __forceinline__ void drain(int k, const Acc& acc, Res& rv) {
        if (!active) return;
        if (RES && !PIPE) load_res(k, rv);            // (PIPE: chunk k's rows were requested by its own step)
#pragma unroll
        for (int i = 0; i < MT; ++i) store_group(k, i, acc, rv);
    }

    __device__ __forceinline__ void run(const WsP& q) {
        const int tid = threadIdx.x;
        gelu = (const float2*)(smem + NS * STAGE);
        if (GEGLU) gelu_tab_init((float2*)(smem + NS * STAGE), tid, NW * 64);
        lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        frow = lane & 15;
        fq = lane >> 4;
        // block -> (row group, panel): blocks b and b+8 share an XCD, so linear slot (b&7)*32 + (b>>3)
        // keeps the panels of one row group next to each other on one XCD (L2 hits on their common chunks)
        const int slot = (blockIdx.x & 7) * q.per_xcd + (blockIdx.x >> 3);
        const int group = slot / q.nt, panel = slot - group * q.nt;
        if (group >= q.groups) return;
        c0 = (int)((long long)q.nch * group / q.groups);
        nch = (int)((long long)q.nch * (group + 1) / q.groups) - c0;
        if (nch <= 0) return;
        n0 = panel * (32 * NW) + wave * 32;
        active = n0 < p.N;
        a_src = p.a + (size_t)(lane >> 3) * p.lda + ((lane & 7) ^ (lane >> 3)) * 8;
#pragma unroll
        for (int j = 0; j < PF; ++j)
            if (j < nch) issue(j);
#ifdef VDX_STAMPS
        t_last = WS_T();
#endif

        cur_set = 0;
        set_left = 0;
        if (WSET) {
            const int cps = p.wset_rows / ROWS;      // chunks per weight set
            cur_set = c0 / cps;
            set_left = cps - (c0 - cur_set * cps);
        }
        load_weights(cur_set);
#pragma unroll
        for (int j = 0; j < 8; ++j) bv[j] = (f16)0.f;
        if (p.bias && active) bv = *(const f16x8*)(p.bias + n0 + 8 * fq);

        Res r0, r1;
#pragma unroll
        for (int i = 0; i < MT; ++i) r0[i] = r1[i] = bv;         // (defined values for the paths that never load them)
        if (!PIPE) {
            for (int k = 0; k < nch; ++k) {
                sync(k);
                Acc acc;
                step<false>(k, acc, acc, r0, r0);
                drain(k, acc, r0);
            }
        } else {
            Acc a, b;
            sync(0);
            step<false>(0, a, a, r0, r1);
            int k = 1;
            for (; k + 1 < nch; k += 2) {  // odd chunks: accumulators b, residual set r1; even chunks: a, r0
                sync(k);
                step<true>(k, b, a, r1, r0);
                sync(k + 1);
                step<true>(k + 1, a, b, r0, r1);
            }
            if (k < nch) {                 // nch even: chunk nch-1 goes into b, then drain b
                sync(k);
                step<true>(k, b, a, r1, r0);
                drain(k, b, r1);
            } else {                       // nch odd: the last chunk is in a
                drain(nch - 1, a, r0);
            }
        }
#ifdef VDX_STAMPS
        if (threadIdx.x == 0) {
            unsigned long long* d = g_ws_stamps + blockIdx.x * 4;
            d[0] = t_bar; d[1] = t_iss; d[2] = t_wait; d[3] = (unsigned long long)nch;
        }
#endif
    }
};

template <int K, int NW, int ROWS, bool GEGLU, bool RES, bool PIPE, bool WSET>
__global__ __launch_bounds__(NW * 64) void gemm_ws_kernel(const WsP q) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Ws<K, NW, ROWS, GEGLU, RES, PIPE, WSET> ws(q.g, smem);
    ws.run(q);
}

template <int K, int NW, int ROWS, bool GEGLU, bool RES, bool PIPE, bool WSET = false>
int launch_ws(const GemmP& p, hipStream_t st) {
    typedef Ws<K, NW, ROWS, GEGLU, RES, PIPE, WSET> W;
    static_assert(!(WSET && GEGLU), "weight sets run the plain epilogue");
    constexpr int lds = W::NS * W::STAGE + (GEGLU ? GELU_TAB_BYTES : 0) + (WSET ? NW * 64 * 32 : 0);
    auto kern = gemm_ws_kernel<K, NW, ROWS, GEGLU, RES, PIPE, WSET>;
    // one-time LDS opt-in; a function-local static is initialised exactly once even under concurrent callers
    static const hipError_t attr_rc = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (attr_rc != hipSuccess) return vdx_fail("gemm_ws: cannot reserve %d bytes of LDS", lds);
    WsP q;
    q.g = p;
    q.nt = (p.N + 32 * NW - 1) / (32 * NW);
    int grid = vdx_grid_cus() < 256 ? vdx_grid_cus() : 256;       // one block per CU (minus the reserve); a multiple of 8
    if (grid < q.nt) grid = (q.nt + 7) & ~7;                      // (a reserve so large that one row group no longer fits)
    q.groups = grid / q.nt;
    q.nch = p.M / ROWS;
    q.per_xcd = grid / 8;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, st, q);
    return vdx_launch_status("vdx_gemm_f16 (weights-stationary)");
}

template <int K, int NW, int ROWS, bool PIPE>
int launch_ws_epi(const GemmP& p, bool geglu, hipStream_t st) {
    if (geglu) return launch_ws<K, NW, ROWS, true, false, PIPE>(p, st);
    return p.res ? launch_ws<K, NW, ROWS, false, true, PIPE>(p, st) : launch_ws<K, NW, ROWS, false, false, PIPE>(p, st);
}

}  // namespace

// Shape family: 0 = not usable; 1 = K 320, 10 waves (320-column panels, N % 320 == 0);
// 2 = K 320, 8 waves pipelined (256-column panels); 3 = K 512; 4 = K 640 (both 8 waves pipelined, 32-row chunks)
int vdx_gemm_ws_family(const GemmP& p, int mode, bool geglu) {
    if (mode != VDX_GEMM_PLAIN || p.c2 != 0 || p.bias2 != nullptr || p.N % 32 != 0 || p.N > 256 * 256) return 0;
    if (p.K == 320 && p.M % 64 == 0) {
        if (p.N % 320 == 0 && !(geglu && p.N % 256 == 0)) return 1;
        return 2;
    }
    if (p.K == 512 && p.M % 32 == 0 && p.N % 256 == 0) return 3;   // (512 -> 320: the 64-column last panel loses to the tiled kernel)
    if (p.K == 640 && p.M % 32 == 0) return 4;
    return 0;
}

int vdx_gemm_ws_launch(const GemmP& p, int family, bool geglu, hipStream_t st) {
    if (p.wset_rows > 0) {      // (gemm_prepare: plain epilogue, families 1 / 2 / 4)
        switch (family) {
            case 1: return launch_ws<320, 10, 64, false, false, false, true>(p, st);
            case 2: return launch_ws<320, 8, 64, false, false, true, true>(p, st);
            case 4: return launch_ws<640, 8, 32, false, false, true, true>(p, st);
        }
        return vdx_fail("gemm_ws: weight sets on family %d", family);
    }
    switch (family) {
        case 1: return launch_ws_epi<320, 10, 64, false>(p, geglu, st);
        case 2: return launch_ws_epi<320, 8, 64, true>(p, geglu, st);
        case 3: return launch_ws_epi<512, 8, 32, true>(p, geglu, st);
        case 4: return launch_ws_epi<640, 8, 32, true>(p, geglu, st);
    }
    return vdx_fail("gemm_ws: shape not supported");
}

// Lab variants of this translation unit (phase stamps, ablations: timing only, some give WRONG results) are compiled in only
// under the macros below; a library that carries one says so through vdx_build_flags() and vdx/_lib.py refuses to load it
// as the product.
extern "C" int vdx_lab_gemm_ws(void) {
#if defined(VDX_STAMPS) || defined(VDX_WS_NS3)
    return 2;
#else
    return 0;
#endif
}
