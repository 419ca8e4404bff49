// tattn_fused.hip — K7: one attention sub-block of diffusers' TransformerTemporalModel as ONE kernel
// (reached from fsdp_chunked_coherent.py:140):
//
//     t' = t + to_out( softmax_F( q k^T / 8 ) v ),   [q|k|v] = LayerNorm(t) . W_qkv^T     (per latent pixel, over its F frames)
//
// Un-fused this is LayerNorm -> q|k|v GEMM -> F x F attention -> output GEMM (+bias, +residual): 14 passes over the
// [pixels x frames][inner] activation matrix through HBM for 4*inner^2 MACs per row.  Here a workgroup owns a TILE of
// pixels with ALL their frames; the rows are read once and written once:
//
//   P0  the tile's rows (48 per row group = G pixels x F frames, G = 48 / F) are loaded, normalised (fp32 statistics,
//       two passes over registers) and parked in LDS as fp16 (the MFMA operand image, XOR-swizzled);
//   P1  per head: q, k, v = X . W^T on v_mfma_f32_16x16x32_f16 with the weights STREAMED through a ring of LDS
//       stages by LDS-DMA (host-packed stage images: the DMA is a linear copy, every fragment read is a
//       conflict-free ds_read_b128).  q and k are produced transposed ([d][row]: operands W, X) and v straight
//       ([row][d]: operands X, W), so that all three accumulators ARE the operands of the attention products
//       (v_mfma_f32_16x16x16_f16, k = 4*(lane>>4)+reg on both sides): S^T = K Q^T has the query on the lane, the
//       softmax over the keys is 12 registers + two cross-lane exchanges, P^T is the B operand of O^T = V^T P^T.
//       Rows of different pixels that share a 16-row tile are masked.  Nothing of q, k, v, S, P ever leaves registers;
//   P2  the heads' outputs (fp16) replace X in LDS;
//   P3  t' = O . W_o^T + b_o + t with W_o streamed through the same ring; 16-byte stores.
//
// Geometry: 4 waves, one per SIMD (up to 512 registers each: 144 accumulators + the heads' outputs + fragments),
// as NRG row groups x NCH column halves: a wave owns row group rg and, of every head pair / column range, part ch.
// All waves read every weight stage, so a weight byte fetched from L2 feeds NRG*48 rows.
//   inner 320 (levels 0): NRG 4, NCH 1 — 192 rows per block, 12 KB stages, 3-deep ring
//   inner 512 (transformer_in): NRG 2, NCH 2 — 96 rows per block, 24 KB stages, 2-deep ring
// F must divide 48 (the BASELINE chunks: 24, 16, 12; also 8, 6, 4, 3, 2, 1); other shapes take the un-fused kernels.
#include "vdx_common.h"

#ifdef VDX_STAMPS   // diagnostic build only (make stamps): per-block phase cycle totals of one wave; never in the product library
static __device__ unsigned long long g_k7_stamps[4096 * 8];
extern "C" int vdx_debug_read_k7_stamps(void* dst) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_k7_stamps), sizeof(unsigned long long) * 4096 * 8) == hipSuccess ? 0 : -1;
}
#define K7_T(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_[i] += now_ - last_; last_ = now_; }
#define K7_STAMP_ACC(i, a, b, c) { asm volatile("" ::"v"(a), "v"(b), "v"(c)); K7_T(i) }
#else
#define K7_T(i)
#define K7_STAMP_ACC(i, a, b, c)
#endif

#ifdef K7_ABL_NOMFMA      /* diagnostic: one MFMA column per step instead of all (keeps the fragment reads alive) */
#define K7_MFMA_J4 1
#define K7_MFMA_JCG 1
#else
#define K7_MFMA_J4 4
#define K7_MFMA_JCG CG
#endif
// Issue order inside a K step (a compile-time directive, LLVM sched_group_barrier): the step's 15 fragment reads for
// the NEXT stage and its 3 DMA pieces are spread between the MFMAs, two MFMAs per memory instruction, instead of
// all 15 reads being issued (~200 cycles of this wave's issue time, matrix pipe idle) before the first MFMA.
#define K7_INTERLEAVE(NMFMA)                                                                  \
    _Pragma("unroll") for (int g_ = 0; g_ < 3; ++g_) {                                         \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);     /* MFMA */                        \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     /* VMEM read: one DMA piece */    \
    }                                                                                          \
    _Pragma("unroll") for (int g_ = 0; g_ < 15; ++g_) {                                        \
        __builtin_amdgcn_sched_group_barrier(0x008, (NMFMA) >= 36 ? 2 : 1, 0);                 \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     /* DS read */                    \
    }
#ifdef K7_DBG_XBAR
#define K7_DBG_STEP_END { __builtin_amdgcn_s_waitcnt(0xC07F); k7_barrier(); }
#else
#define K7_DBG_STEP_END
#endif

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// workgroup barrier that the COMPILER also treats as a memory barrier: LLVM models the s_barrier builtin as touching no
// memory, so LDS reads of a stage could be scheduled above the barrier that publishes it (seen: corrupted rows).
__device__ __forceinline__ void k7_barrier() {
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct K7P {
    const f16* t;        // [M][ldt] rows, row = (b*F + f)*S + p
    const f16 *gamma, *beta;
    const f16* wqkv;     // packed stage images (packing.pack_k7_qkv)
    const f16* wo;       // packed stage images (packing.pack_k7_out)
    const f16* bo;       // [inner]
    f16* out;            // [M][ldo]
    int ldt, ldo;
    int B, F, S;         // batch items, frames per pixel, pixels per frame
    int G;               // pixels per row group = 48 / F
    int gpb;             // row groups per batch item = ceil(S / G)
    int ngroups;         // B * gpb
    int fmagic;          // ceil(65536 / F): row / F == (row * fmagic) >> 16 for row < 64
    float eps, c;        // LayerNorm eps; softmax scale * log2(e)
};

template <int INNER, int NRG, int NCH>
struct K7 {
    static constexpr int KS = INNER / 32;                 // MFMA k steps over the model width
    static constexpr int HEADS = INNER / 64;
    static constexpr int NHG = HEADS / NCH;               // head groups (NCH heads at a time, one per column half)
    static constexpr int ROWS = NRG * 48;
    static constexpr int RB = INNER * 2;                  // bytes of one X row
    static constexpr int XB = ROWS * RB;
    static constexpr int SB = NCH * 192 * 64;             // bytes of one weight stage ([NCH*192 rows][32 k])
    static constexpr int NS = (160 * 1024 - XB) / SB >= 3 ? 3 : 2;
    static constexpr int PPW = SB / 1024 / 4;             // DMA pieces per wave and stage
    static constexpr int WCOLS = INNER / NCH;             // output columns of P3 per wave
    static constexpr int CG = WCOLS % 160 == 0 ? 10 : 8;  // 16-column tiles per P3 column group
    static constexpr int NCG = WCOLS / (16 * CG);
    static constexpr int NST = (NHG + NCG) * KS;          // stages of the whole weight stream
    static constexpr int CPL = INNER / 64;                // 16-byte chunks per lane in P0 (8 lanes per row)
    static constexpr bool SW16 = (INNER / 8) % 16 == 0;   // row stride in 16-byte slots mod 16: 0 -> swizzle by row&15
    static_assert(HEADS % NCH == 0 && WCOLS % (16 * CG) == 0 && CG % 2 == 0 && CG * 16 * NCH * 64 <= SB, "geometry");
    static_assert(XB + NS * SB <= 160 * 1024 && SB % 4096 == 0, "LDS budget");
    static_assert(NRG * NCH == 4 && KS % 2 == 0 && CG <= 12, "four waves; the K loops alternate two fragment sets");

    __device__ static __forceinline__ int swz(int row) { return SW16 ? (row & 15) : ((row >> 1) & 7); }

    const K7P& p;
    char* smem;
    int lane, n16, q4, rg, ch, wave, rot;
    int woff;            // this lane's byte offset inside a 16-row weight fragment tile of a stage

    __device__ __forceinline__ K7(const K7P& p_, char* s) : p(p_), smem(s) {}

    // Stage s of the stream.  Blocks that share an XCD walk the head groups in ROTATED order (rot = position among the
    // XCD's blocks): all blocks take the same time per tile, so without the rotation the 32 CUs of an XCD ask for the
    // same weight lines within a few hundred cycles and then not again for a whole tile (~70 us), by when the rows
    // streaming through the 4 MB L2 have evicted them — every stage then pays the latency of a miss.  Rotated, every
    // weight line is re-read by some CU every few microseconds and stays in L2.
    __device__ __forceinline__ const char* stage_src(int s) const {
        if (s >= NHG * KS) return (const char*)p.wo + (size_t)(s - NHG * KS) * SB;
        const int hg = s / KS, ks = s - hg * KS;
        int h = hg + rot;
        if (h >= NHG) h -= NHG;
        return (const char*)p.wqkv + (size_t)(h * KS + ks) * SB;
    }
    __device__ __forceinline__ void issue(int s) {
#ifdef K7_ABL_NODMA
        return;
#endif
        const char* src = stage_src(s) + (wave * PPW) * 1024 + lane * 16;
        char* dst = smem + XB + (s % NS) * SB + (wave * PPW) * 1024;
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + i * 1024), (lptr_t)(dst + i * 1024), 16, 0, 0);
    }
    // Stream protocol.  At the top of K step s every wave holds the fragments of stage s in REGISTERS (they were read
    // during step s-1, behind that step's MFMAs).  step_sync(s): my DMA pieces of stage s+1 have landed; barrier — now
    // stage s+1 is complete in LDS for everyone, and nobody needs the slot of stage s any more, so stage s+NS is
    // issued into it.  The caller then reads the fragments of stage s+1 into the other register set and runs the
    // MFMAs of stage s: LDS latency, DMA issue and the DMA's flight all sit behind matrix work.
    __device__ __forceinline__ void step_sync(int s) {
        // (asm statements only order MEMORY operations: without this the waits and the barrier below were scheduled
        // above the previous step's register-only MFMAs, i.e. the wave waited for its fragment reads before multiplying)
        __builtin_amdgcn_sched_barrier(0);
#ifdef K7_DBG_VM0
        wait_vmcnt<0>();
#else
        if (NS == 3) {
            if (s + 2 < NST) wait_vmcnt<PPW>(); else wait_vmcnt<0>();     // (stage s+2 may still be in flight)
        } else {
            wait_vmcnt<0>();
        }
#endif
        // my LDS accesses are done before I signal.  The BUILTIN form: the compiler's wait-count pass sees it and knows
        // the fragment registers read last step are valid (behind an asm wait it re-waited lgkmcnt(0) after issuing
        // this step's reads, which put their latency in front of the MFMAs)
        __builtin_amdgcn_s_waitcnt(0xC07F);
        asm volatile("" ::: "memory");
#ifdef VDX_STAMPS
        const unsigned long long t0_ = __builtin_amdgcn_s_memtime();
#endif
        k7_barrier();
#ifdef VDX_STAMPS
        const unsigned long long t1_ = __builtin_amdgcn_s_memtime();
#endif
        if (s + NS < NST) issue(s + NS);
#ifdef VDX_STAMPS
        const unsigned long long t2_ = __builtin_amdgcn_s_memtime();
        acq_bar += t1_ - t0_;
        acq_iss += t2_ - t1_;
#endif
    }
#ifdef VDX_STAMPS
    unsigned long long acq_bar = 0, acq_iss = 0;
#endif
    struct Frags {
        f16x8 w[12], x[3];
    };
    __device__ __forceinline__ f16x8 xfrag(int i, int ks) const {
        const int row = rg * 48 + 16 * i + n16;
        return *(const f16x8*)(smem + row * RB + (((4 * ks + q4) ^ swz(row)) << 4));
    }
    __device__ __forceinline__ f16x8 wfrag(int slot, int tile) const {
        return *(const f16x8*)(smem + XB + slot * SB + tile * 1024 + woff);
    }

    template <int NT>
    __device__ __forceinline__ void read_w(Frags& f, int slot) const {
#pragma unroll
        for (int j = 0; j < NT; ++j) f.w[j] = wfrag(slot, ch * NT + j);
    }
    __device__ __forceinline__ void read_x(Frags& f, int ks) const {
#pragma unroll
        for (int i = 0; i < 3; ++i) f.x[i] = xfrag(i, ks);
    }

    // global row of local row r (0..47) of this wave's row group, or -1
    __device__ __forceinline__ long long grow_of(int gi, int r) const {
        const int g = r / p.F, f = r - g * p.F;
        const int b = gi / p.gpb, pix = (gi - b * p.gpb) * p.G + g;
        if (gi >= p.ngroups || pix >= p.S) return -1;
        return ((long long)b * p.F + f) * p.S + pix;
    }

    __device__ __forceinline__ void run() {
        const int tid = threadIdx.x;
        lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        rg = wave / NCH;
        ch = wave % NCH;
        n16 = lane & 15;
        q4 = lane >> 4;
        {
            const int g = (0x1320 >> (4 * (n16 >> 2))) & 3;      // g = [0, 2, 3, 1][n >> 2]
            woff = n16 * 64 + ((q4 ^ g) << 4);
        }
        const int gi = blockIdx.x * NRG + rg;                   // this wave's row group
        rot = (blockIdx.x >> 3) % NHG;                          // blocks b and b+8 share an XCD
        const f16* zp = (const f16*)g_zero_page;

#ifdef VDX_STAMPS
        unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memtime();
#endif
        // the weight stream starts before the rows are even loaded
#pragma unroll
        for (int s = 0; s < NS; ++s) issue(s);

        // ---- P0: rows -> LayerNorm -> X (fp16, swizzled) in LDS.  8 lanes per row, CPL chunks of 8 channels per lane
        {
            constexpr int RPW = 48 / NCH;                       // rows of the group this wave normalises
            const int sub = lane & 7;
            f16x8 gm[CPL], bt[CPL];
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                gm[j] = *(const f16x8*)(p.gamma + 8 * (sub + 8 * j));
                bt[j] = *(const f16x8*)(p.beta + 8 * (sub + 8 * j));
            }
            constexpr int NPS = RPW / 8;
            f16x8 v[NPS][CPL];
            long long grs[NPS];
            // every pass's loads are in flight before the first is used (a pass at a time is one HBM round trip each)
#pragma unroll
            for (int ps = 0; ps < NPS; ++ps) {
                const int r = ch * RPW + 8 * ps + (lane >> 3);
                grs[ps] = grow_of(gi, r);
                const f16* src = grs[ps] >= 0 ? p.t + grs[ps] * p.ldt + 8 * sub : zp;
                const int step = grs[ps] >= 0 ? 64 : 0;
#ifdef K7_ABL_NOP0
#pragma unroll
                for (int j = 0; j < CPL; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[ps][j][e] = (f16)(0.01f * (float)(lane + j + e));
#else
#pragma unroll
                for (int j = 0; j < CPL; ++j)
#ifdef K7_EXP_NT
                    v[ps][j] = __builtin_nontemporal_load((const f16x8*)(src + j * step));
#else
                    v[ps][j] = *(const f16x8*)(src + j * step);
#endif
#endif
            }
#pragma unroll
            for (int ps = 0; ps < NPS; ++ps) {
                const int r = ch * RPW + 8 * ps + (lane >> 3);
                float sum = 0.f;
#pragma unroll
                for (int j = 0; j < CPL; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e) sum += (float)v[ps][j][e];
                sum += __shfl_xor(sum, 1, 64);
                sum += __shfl_xor(sum, 2, 64);
                sum += __shfl_xor(sum, 4, 64);
                const float mean = sum * (1.0f / INNER);
                float var = 0.f;
#pragma unroll
                for (int j = 0; j < CPL; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float d = (float)v[ps][j][e] - mean;
                        var += d * d;
                    }
                var += __shfl_xor(var, 1, 64);
                var += __shfl_xor(var, 2, 64);
                var += __shfl_xor(var, 4, 64);
                const float rstd = rsqrtf(var * (1.0f / INNER) + p.eps);
                const int row = rg * 48 + r;
                char* dst = smem + row * RB;
                const int sw = swz(row);
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    f16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        o[e] = grs[ps] >= 0 ? (f16)(((float)v[ps][j][e] - mean) * rstd * (float)gm[j][e] + (float)bt[j][e]) : (f16)0.f;
                    *(f16x8*)(dst + (((sub + 8 * j) ^ sw) << 4)) = o;
                }
            }
        }
        // (the first acquire()'s barrier publishes X)
        K7_T(0)

        // ---- P1: heads.  This wave: head hg*NCH + ch of every head group hg.
        f16x4 ohead[NHG][3][4];                                  // [head group][query tile][d tile]: O^T, 4 consecutive d
        // pixel of my query rows / key rows inside the 48-row group (for the block-diagonal mask)
        // (row / F as a multiply: exact for rows < 64 and every F this kernel accepts)
        int qpix[3], kpix[3][4];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            qpix[i] = ((16 * i + n16) * p.fmagic) >> 16;
#pragma unroll
            for (int e = 0; e < 4; ++e) kpix[i][e] = ((16 * i + 4 * q4 + e) * p.fmagic) >> 16;
        }
        // 16-row tiles of the group that share no pixel need no score tile at all (F = 16: only the diagonal; F = 24:
        // 7 of 9): bit 3*qt + kt of `need` says query tile qt has a pixel in common with key tile kt (wave-uniform)
        int need = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int alo = 16 * a / p.F, ahi = (16 * a + 15) / p.F, blo = 16 * b / p.F, bhi = (16 * b + 15) / p.F;
                if (!(ahi < blo || bhi < alo)) need |= 1 << (3 * a + b);
                if (alo == ahi && blo == bhi && alo == blo) need |= 1 << (9 + 3 * a + b);     // both tiles inside ONE pixel: no mask
            }
        need = __builtin_amdgcn_readfirstlane(need);
        // stage 0 has landed for everyone (and X is complete): its fragments open the pipeline
        if (NS == 3) wait_vmcnt<2 * PPW>(); else wait_vmcnt<PPW>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        k7_barrier();
        const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
        Frags fa, fb;
        read_w<12>(fa, 0);
        read_x(fa, 0);
        int s_idx = 0, slot = 0;
        // one K step: synchronise, fetch the NEXT stage's fragments into `nxt`, multiply `cur`
#define K7_NEXT_SLOT (slot + 1 == NS ? 0 : slot + 1)
#define K7_P1_STEP(cur, nxt, ksv, Z)                                                                           \
        {                                                                                                     \
            step_sync(s_idx);                                                                                 \
            K7_T(1)                                                                                           \
            const int ns_ = K7_NEXT_SLOT;                                                                     \
            if ((ksv) + 1 < KS) { read_w<12>(nxt, ns_); read_x(nxt, (ksv) + 1); }                             \
            else if (hg + 1 < NHG) { read_w<12>(nxt, ns_); read_x(nxt, 0); }                                  \
            else read_w<CG>(nxt, ns_);            /* first stage of P3: its X operand (O) is not in LDS yet */  \
            _Pragma("unroll") for (int j = 0; j < K7_MFMA_J4; ++j)                                            \
                _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                               \
                    aq[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur.w[j], cur.x[i], (Z) ? zero4 : aq[i][j], 0, 0, 0);     /* [d][row] */ \
                    ak[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur.w[4 + j], cur.x[i], (Z) ? zero4 : ak[i][j], 0, 0, 0); /* [d][row] */ \
                    av[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur.x[i], cur.w[8 + j], (Z) ? zero4 : av[i][j], 0, 0, 0); /* [row][d] */ \
                }                                                                                             \
            K7_INTERLEAVE(36)                                                                                 \
            ++s_idx;                                                                                          \
            slot = ns_;                                                                                       \
            K7_STAMP_ACC(2, aq[0][0], ak[2][3], av[2][3])                                                     \
            K7_DBG_STEP_END                                                                                   \
        }
#pragma unroll
        for (int hg = 0; hg < NHG; ++hg) {                       // (unrolled: ohead must be indexed statically)
            f32x4 aq[3][4], ak[3][4], av[3][4];
            // K step 0 starts from the constant 0 (an inline operand of the MFMA): 144 accumulator registers are not cleared
            K7_P1_STEP(fa, fb, 0, true)
            K7_P1_STEP(fb, fa, 1, false)
#pragma unroll 1
            for (int ks = 2; ks < KS; ks += 2) {                 // two steps per trip: the register sets alternate statically
                K7_P1_STEP(fa, fb, ks, false)
                K7_P1_STEP(fb, fa, ks + 1, false)
            }
            // ---- attention of this head on the wave's 48 rows (G pixels x F frames), all in registers.
            // (S and P.V on v_mfma_f32_16x16x16_f16, whose operand layout IS the accumulator layout.)
#ifdef K7_ABL_NOATT
#pragma unroll
            for (int qt = 0; qt < 3; ++qt)
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) ohead[hg][qt][dt][e] = (f16)(aq[qt][dt][e] + ak[qt][dt][e] + av[qt][dt][e]);
#else
            f16x4 qh[3][4], kh[3][4], vh[3][4];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        qh[i][j][e] = (f16)aq[i][j][e];
                        kh[i][j][e] = (f16)ak[i][j][e];
                        vh[i][j][e] = (f16)av[i][j][e];
                    }
#pragma unroll
            for (int qt = 0; qt < 3; ++qt) {
                f32x4 st[3];
#pragma unroll
                for (int kt = 0; kt < 3; ++kt) {
                    st[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if ((need >> (3 * qt + kt)) & 1) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            st[kt] = __builtin_amdgcn_mfma_f32_16x16x16f16(kh[kt][j], qh[qt][j], st[kt], 0, 0, 0);
                    }
                }
                float mx = NEG_BIG_K7();
                bool ok[3][4];
#pragma unroll
                for (int kt = 0; kt < 3; ++kt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool pure = (need >> (9 + 3 * qt + kt)) & 1;       // wave-uniform
                        ok[kt][e] = pure || (((need >> (3 * qt + kt)) & 1) && kpix[kt][e] == qpix[qt]);
                        mx = fmaxf(mx, ok[kt][e] ? st[kt][e] : NEG_BIG_K7());
                    }
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float mc = mx * p.c;
                float rs = 0.f;
#pragma unroll
                for (int kt = 0; kt < 3; ++kt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        st[kt][e] = ok[kt][e] ? __builtin_amdgcn_exp2f(st[kt][e] * p.c - mc) : 0.f;
                        rs += st[kt][e];
                    }
                rs += __shfl_xor(rs, 16, 64);
                rs += __shfl_xor(rs, 32, 64);
                const float inv = 1.0f / rs;
                f16x4 pt[3];
#pragma unroll
                for (int kt = 0; kt < 3; ++kt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) pt[kt][e] = (f16)(st[kt][e] * inv);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kt = 0; kt < 3; ++kt)
                        if ((need >> (3 * qt + kt)) & 1) o = __builtin_amdgcn_mfma_f32_16x16x16f16(vh[kt][dt], pt[kt], o, 0, 0, 0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) ohead[hg][qt][dt][e] = (f16)o[e];
                }
            }
#endif
#ifdef VDX_STAMPS
            asm volatile("" ::"v"(ohead[hg][2][3]));
            K7_T(3)
#endif
        }

        // ---- P2: the heads' outputs replace X (same image: row-major, swizzled 16-byte chunks)
        k7_barrier();            // every wave is past its last read of X
#pragma unroll
        for (int hg = 0; hg < NHG; ++hg) {
            const int head = ((hg + rot) % NHG) * NCH + ch;      // (ohead is indexed by time; the head it holds is rotated)
#pragma unroll
            for (int qt = 0; qt < 3; ++qt) {
                const int row = rg * 48 + 16 * qt + n16;
                char* dst = smem + row * RB + 8 * (q4 & 1);
                const int sw = swz(row);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const int chunk = head * 8 + 2 * dt + (q4 >> 1);       // channels head*64 + 16*dt + 4*q4 .. +3
                    *(f16x4*)(dst + ((chunk ^ sw) << 4)) = ohead[hg][qt][dt];
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        k7_barrier();            // O is complete in LDS
        read_x(fa, 0);                           // (fa already holds the weight fragments of the first P3 stage)
        K7_T(4)

        // ---- P3: t' = O . Wo^T + bo + t.  This wave: rows rg, columns ch*WCOLS + cg*16*CG .. of every column group
        long long grow[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) grow[i] = grow_of(gi, 16 * i + n16);
#define K7_P3_STEP(cur, nxt, ksv, Z)                                                                           \
        {                                                                                                     \
            step_sync(s_idx);                                                                                 \
            K7_T(5)                                                                                           \
            const int ns_ = K7_NEXT_SLOT;                                                                     \
            if ((ksv) + 1 < KS) { read_w<CG>(nxt, ns_); read_x(nxt, (ksv) + 1); }                             \
            else if (cg + 1 < NCG) { read_w<CG>(nxt, ns_); read_x(nxt, 0); }                                  \
            _Pragma("unroll") for (int j = 0; j < K7_MFMA_JCG; ++j)                                           \
                _Pragma("unroll") for (int i = 0; i < 3; ++i)                                                 \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur.w[j], cur.x[i], (Z) ? zero4 : acc[i][j], 0, 0, 0);   /* [col][row] */ \
            K7_INTERLEAVE(3 * CG)                                                                             \
            ++s_idx;                                                                                          \
            slot = ns_;                                                                                       \
            K7_STAMP_ACC(6, acc[0][0], acc[2][CG - 1], acc[1][0])                                             \
            K7_DBG_STEP_END                                                                                   \
        }
#pragma unroll 1
        for (int cg = 0; cg < NCG; ++cg) {
            f32x4 acc[3][CG];
            // residual rows and bias of this column group: requested now, consumed after the K loop (rows that do not
            // exist read the zero page; nothing is stored for them)
            const int cb = ch * WCOLS + cg * 16 * CG + 8 * q4;
            f16x8 rv[3][CG / 2], bv[CG / 2];
#pragma unroll
            for (int a = 0; a < CG / 2; ++a) {
                bv[a] = *(const f16x8*)(p.bo + cb + 32 * a);
#pragma unroll
                for (int i = 0; i < 3; ++i)
#ifdef K7_ABL_NOEPI
                    rv[i][a] = bv[a];
#else
                    rv[i][a] = *(const f16x8*)(grow[i] >= 0 ? p.t + grow[i] * p.ldt + cb + 32 * a : zp);
#endif
            }
            K7_P3_STEP(fa, fb, 0, true)
            K7_P3_STEP(fb, fa, 1, false)
#pragma unroll 1
            for (int ks = 2; ks < KS; ks += 2) {
                K7_P3_STEP(fa, fb, ks, false)
                K7_P3_STEP(fb, fa, ks + 1, false)
            }
            // epilogue: tile pair (2a, 2a+1) gives this lane 8 consecutive columns 32a + 8*q4 .. +7 of row n16 (+16i)
#pragma unroll
            for (int a = 0; a < CG / 2; ++a) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    f16x8 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o[e] = (f16)(acc[i][2 * a][e] + (float)bv[a][e] + (float)rv[i][a][e]);
                        o[4 + e] = (f16)(acc[i][2 * a + 1][e] + (float)bv[a][4 + e] + (float)rv[i][a][4 + e]);
                    }
                    if (grow[i] >= 0) *(f16x8*)(p.out + grow[i] * p.ldo + cb + 32 * a) = o;
                }
            }
#ifdef VDX_STAMPS
            __builtin_amdgcn_s_waitcnt(0);
            K7_T(7)
#endif
        }
#ifdef VDX_STAMPS
        if (lane == 0 && (wave == 0) && blockIdx.x < 4096)
            for (int i = 0; i < 8; ++i) g_k7_stamps[blockIdx.x * 8 + i] = i == 4 ? acq_bar : (i == 0 ? st_[0] + (acq_iss << 32) : st_[i]);
#endif
    }

    __device__ static __forceinline__ float NEG_BIG_K7() { return -1.0e30f; }
};

template <int INNER, int NRG, int NCH>
__global__ __launch_bounds__(256, 1) void tattn_fused_kernel(const K7P p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    K7<INNER, NRG, NCH> k(p, smem);
    k.run();
}

template <int INNER, int NRG, int NCH>
int launch_k7(const K7P& p, hipStream_t st) {
    typedef K7<INNER, NRG, NCH> T;
    constexpr int lds = T::XB + T::NS * T::SB;
    auto kern = tattn_fused_kernel<INNER, NRG, NCH>;
    static const hipError_t attr_rc = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (attr_rc != hipSuccess) return vdx_fail("temporal_attn_block: cannot reserve %d bytes of LDS", lds);
    const int tiles = (p.ngroups + NRG - 1) / NRG;
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), lds, st, p);
    return vdx_launch_status("vdx_temporal_attn_block_f16");
}

}  // namespace

extern "C" int vdx_temporal_attn_block_supported(int inner, int F) {
    return (inner == 320 || inner == 512) && F >= 1 && F <= 48 && 48 % F == 0;
}

// bytes of the packed weight images (host packing must produce exactly these)
extern "C" size_t vdx_temporal_attn_block_wqkv_bytes(int inner) {
    if (inner == 320) return (size_t)K7<320, 4, 1>::NHG * K7<320, 4, 1>::KS * K7<320, 4, 1>::SB;
    if (inner == 512) return (size_t)K7<512, 2, 2>::NHG * K7<512, 2, 2>::KS * K7<512, 2, 2>::SB;
    return 0;
}
extern "C" size_t vdx_temporal_attn_block_wo_bytes(int inner) {
    if (inner == 320) return (size_t)K7<320, 4, 1>::NCG * K7<320, 4, 1>::KS * K7<320, 4, 1>::SB;
    if (inner == 512) return (size_t)K7<512, 2, 2>::NCG * K7<512, 2, 2>::KS * K7<512, 2, 2>::SB;
    return 0;
}

extern "C" int vdx_temporal_attn_block_f16(const void* t, int ldt, const void* gamma, const void* beta, float eps,
                                           const void* wqkv_packed, const void* wo_packed, const void* bo,
                                           void* out, int ldo, int B, int F, int HW, int inner, float scale,
                                           vdx_stream_t stream) {
    VDX_CHECK(t && gamma && beta && wqkv_packed && wo_packed && bo && out, "temporal_attn_block: null pointer");
    VDX_CHECK(B > 0 && F > 0 && HW > 0, "temporal_attn_block: empty problem");
    VDX_CHECK(vdx_temporal_attn_block_supported(inner, F), "temporal_attn_block: inner=%d F=%d not supported (inner 320/512, F | 48)", inner, F);
    VDX_CHECK(ldt % 8 == 0 && ldo % 8 == 0 && ldt >= inner && ldo >= inner, "temporal_attn_block: bad leading dims");
    VDX_CHECK((long long)B * F * HW < (1ll << 31), "temporal_attn_block: too many rows");
    K7P p;
    p.t = (const f16*)t; p.gamma = (const f16*)gamma; p.beta = (const f16*)beta;
    p.wqkv = (const f16*)wqkv_packed; p.wo = (const f16*)wo_packed; p.bo = (const f16*)bo; p.out = (f16*)out;
    p.ldt = ldt; p.ldo = ldo; p.B = B; p.F = F; p.S = HW;
    p.G = 48 / F;
    p.gpb = (HW + p.G - 1) / p.G;
    p.ngroups = B * p.gpb;
    p.fmagic = (65536 + F - 1) / F;
    p.eps = eps;
    p.c = scale * 1.44269504088896341f;
    hipStream_t st = (hipStream_t)stream;
    if (inner == 320) return launch_k7<320, 4, 1>(p, st);
    return launch_k7<512, 2, 2>(p, st);
}

// Lab variants of this translation unit (phase stamps, ablations: timing only, some give WRONG results) are compiled in only
// under the macros below; a library that carries one says so through vdx_build_flags() and vdx/_lib.py refuses to load it
// as the product.
extern "C" int vdx_lab_tattn_fused(void) {
#if defined(VDX_STAMPS) || defined(K7_ABL_NOP0) || defined(K7_ABL_NOMFMA) || defined(K7_ABL_NOEPI) || defined(K7_ABL_NODMA) || defined(K7_ABL_NOATT)
    return 4;
#else
    return 0;
#endif
}
