// tattn2.hip — K7, second design: one attention sub-block of diffusers' TransformerTemporalModel as ONE kernel
// (reached from fsdp_chunked_coherent.py:140):
//
//     t' = t + to_out( softmax_F( q k^T / 8 ) v ),   [q|k|v] = LayerNorm(t) . W_qkv^T     (per latent pixel, over its F frames)
//
// What changed against csrc/tattn_fused.hip (which still serves inner 512), and why (profiles/r02_k7.md: that kernel is
// issue-bound at one wave per SIMD — 2 680 MFMA against 8 190 other vector instructions per wave and tile, 495 registers):
//   * LayerNorm's affine and the softmax scale are folded into the weights on the host (W' = c.W.diag(gamma) for q,
//     W.diag(gamma) for k and v; the beta terms become a q bias, nothing for k — a per-query constant cancels in the
//     softmax — and a term of the output bias for v): the kernel only centres and scales the rows;
//   * a head is projected in two passes over the tile's rows — q|k (96 accumulator registers), then v (48) — so the
//     scores and the softmax of the head need no register beyond what the q|k pass already owns;
//   * the head's output O^T leaves the P.V product as [d][row] accumulators, which ARE the B operand of the output
//     projection once the projection's k index is permuted to match (the permutation is applied to W_o's columns on
//     the host): no trip through LDS, no second image of the tile, and the projection reads no activation fragment;
//   * the weight stream is cut into 8 KB units ([8 tiles][16 rows][32 k]) in a 5-unit ring that fills the 160 KB of
//     LDS exactly beside the 120 KB row image; a K step is 64 deep (q|k: 2 units, 48 MFMAs per barrier; v: 1 unit).
//
//   * the kernel is PERSISTENT: a workgroup walks tiles blockIdx.x, + gridDim.x, ...; while the output projection of a
//     tile runs (it reads no row image), the rows of the NEXT tile are copied into the image by LDS-DMA (per-lane source
//     address = row gather + XOR swizzle, no register) and centred / scaled in place behind the projection's MFMAs;
//     the weight stream runs on across the tile boundary;
//   * the scores and the softmax of a head (vector work) sit inside the first v steps of that head.
//
// Geometry (inner 320): 4 waves, one per SIMD, each owning a row group of 48 rows = G pixels x F frames (G = 48 / F)
// for the whole tile; nothing a wave reads of the row image is written by another wave (so the image needs no barrier).
#include "vdx_common.h"
#include <utility>

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((address_space(3))) char lchar;                 // LDS addresses as 32-bit arithmetic: a constant
typedef __attribute__((address_space(3))) f16x8 lf16x8;               // term then folds into the DS offset field
typedef __attribute__((address_space(1))) f16 gf16;                   // explicit global pointers: a pointer the optimiser
typedef __attribute__((address_space(1))) f16x8 gf16x8;               // cannot trace becomes a FLAT access (counts on both
typedef __attribute__((address_space(1))) f32x4 gf32x4;               // counters, completes out of order)

// workgroup barrier the COMPILER also treats as a memory barrier (LLVM models s_barrier as touching no memory)
__device__ __forceinline__ void wg_barrier() {
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct K7BP {
    const f16* t;        // [M][ldt] rows, row = (b*F + f)*S + p
    f16* out;            // [M][ldo]
    const char* wqkv;    // [head][15 units][8192 B]: q0 k0 q1 k1 ... q4 k4 v0 .. v4 (a unit: 64 rows x 64 k as 8 tiles)
    const char* wo;      // [cg A: head x 2 units][cg B: head x 2 units][cg C: head x 1 unit]
    const float* bq;     // [inner]  c . W_q . beta
    const float* bo2;    // [inner]  b_o + W_o . (W_v . beta)
    int ldt, ldo;
    int B, F, S;         // batch items, frames per pixel, pixels per frame
    int G;               // pixels per row group = 48 / F
    int gpb;             // row groups per batch item = ceil(S / G)
    int tps;             // tiles per batch item = ceil(gpb / 4): a tile never straddles two batch items
    int ntiles;          // B * tps
    int fmagic;          // ceil(65536 / F): row / F == (row * fmagic) >> 16 for row < 64
    float eps;
};

// stores of rows that do not exist (last tile) go here, so that every epilogue issues the same number of stores and the
// counted s_waitcnt of the next step stays exact
static __device__ __attribute__((aligned(16))) u32x4 g_dump_page[64 + 64];     // lane * 16 bytes + up to 2 * INNER bytes of column offset

__device__ __forceinline__ float dpp_add8(float v) {        // sum over the 8 lanes that share a row (lane & ~7)
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    return v;
}
// value of lanes l, l^16, l^32, l^48 combined (the four lane quads that hold one query's keys): two VALU swaps, no LDS
#ifdef K7B_SHFL
__device__ __forceinline__ float quad_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float quad_sum(float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }
#else
// v_permlane16_swap exchanges the odd 16-lane rows of its first operand with the even rows of its second;
// v_permlane32_swap the upper half of the first with the lower half of the second.  Fed the same value twice, the two
// results together hold the value of both partners in every lane.  Inline asm, not the builtins: hipcc (ROCm 7.2) folds
// `r[0] op r[1]` of the builtin to `r[0] op r[0]` (seen in the ISA as v_add_f32 v, a0, a0 — the reduction is then a
// no-op).  The s_nop covers the VALU-write -> permlane-read hazard (2 wait states), which nothing pads inside asm.
__device__ __forceinline__ void swap16(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void swap32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ float quad_max(float v) {
    float a = v, b = v;
    swap16(a, b);
    a = fmaxf(a, b);
    b = a;
    swap32(a, b);
    return fmaxf(a, b);
}
__device__ __forceinline__ float quad_sum(float v) {
    float a = v, b = v;
    swap16(a, b);
    a = a + b;
    b = a;
    swap32(a, b);
    return a + b;
}
#endif

#ifdef K7B_STAMPS   // diagnostic build only: cycle totals per (step kind, segment) of wave 0; never in the product library
static __device__ unsigned long long g_k7b_stamps[1024 * 16];
extern "C" int vdx_debug_read_k7b_stamps(void* dst) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_k7b_stamps), sizeof(g_k7b_stamps)) == hipSuccess ? 0 : -1;
}
#ifndef K7B_SS
#define K7B_SS 0
#define K7B_SE 1000
#define K7B_IDX(idx) (idx)
#else
#define K7B_IDX(idx) ((idx) % 4)           /* one range of steps: the four segments only */
#endif
#define K7B_T(idx)                                                                                     \
    if constexpr ((S >= K7B_SS && S <= K7B_SE) || S == (K7B_SS + 64) % 65) {                            \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        unsigned long long now_;                                                                       \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");                    \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        if constexpr (S >= K7B_SS && S <= K7B_SE) tsum[K7B_IDX(idx)] += now_ - tlast;                   \
        tlast = now_;                                                                                  \
    }
#else
#define K7B_T(idx)
#endif

// FS, the frame specialisation.  24 / 16 / 12 (the BASELINE chunk lengths): F is a compile-time constant — score tiles of
// 16 rows that share no pixel are not computed at all (F = 16: 3 of 9 tiles remain, F = 24: 7, F = 12: 7), tiles that lie
// inside one pixel need no mask.  4: any F that is a multiple of 4 — the 4 key rows a lane holds of a score tile (rows
// 16*kt + 4*q4 + e) then belong to ONE pixel, so the block-diagonal mask needs one compare per key tile instead of four
// (F = 8, 4, 48).  1: any F | 48.
template <int INNER, int FS>
struct K7B {
    static constexpr bool F4 = FS != 1;
    // tile relations at a compile-time F (16-row tiles a, b of the 48-row group): do they share a pixel / lie in ONE pixel
    static constexpr bool tiles_meet(int a, int b) {
        if (FS < 12) return true;
        const int alo = 16 * a / FS, ahi = (16 * a + 15) / FS, blo = 16 * b / FS, bhi = (16 * b + 15) / FS;
        return !(ahi < blo || bhi < alo);
    }
    static constexpr bool tiles_pure(int a, int b) {
        if (FS < 12) return false;
        const int alo = 16 * a / FS, ahi = (16 * a + 15) / FS, blo = 16 * b / FS, bhi = (16 * b + 15) / FS;
        return alo == ahi && blo == bhi && alo == blo;
    }
    static constexpr int KS = INNER / 32;                 // MFMA k steps over the model width
    static constexpr int HEADS = INNER / 64;
    static constexpr int KM = KS / 2;                     // K-64 steps over the model width
    static constexpr int ROWS = 192;
    static constexpr int RB = INNER * 2;                  // bytes of one row of the image
    static constexpr int XB = ROWS * RB;
    static constexpr int UB = 8192, NU = 5;               // ring: NU units of UB bytes
    static constexpr int HSTEPS = 2 * KM;                 // steps of one head: KM of q|k, KM of v
    static constexpr int P1S = HEADS * HSTEPS;
    static constexpr int NCGF = INNER / 128;              // full 128-column groups of the output projection
    static constexpr int NCG = (INNER + 127) / 128;
    static constexpr int NSTEP = P1S + NCG * HEADS;       // the output projection contracts head by head (K = 64)
    static constexpr int UPH = 3 * KM;                    // units of one head in the q|k|v stream
    static constexpr int CPL = INNER / 64;                // 16-byte chunks per lane in P0 (8 lanes per row)
    static constexpr int CPR = INNER / 8;                 // 16-byte chunks per row
    static constexpr int NPS = 6;                         // P0 passes of 8 rows
    static constexpr int PPP = 8 * RB / 1024;             // DMA pieces per pass
    static_assert(KS % 2 == 0 && (INNER % 128 == 0 || INNER % 128 == 64), "geometry");
    static_assert(XB + NU * UB <= 160 * 1024, "LDS budget");
    static constexpr int NCB = INNER / 64;                // column blocks of 64 channels (= PPP: one DMA piece each)
    static constexpr int RBB = NCB * 1024;                // bytes of one row block (8 rows)
    static_assert(PPP == NCB && XB == 24 * RBB, "a DMA piece is one (row block, column block): 8 rows x 128 bytes");
    static_assert(NCG == 3 && NPS == 6 && HEADS == 5, "the row prefetch schedule below is written for three column groups of five steps");

    // ---- the static schedule of a tile: step s consumes units [ub(s), ub(s+1)) of the weight stream; the stream runs
    // on into the next tile (s >= NSTEP: the same schedule again)
    static constexpr int kind(int s) { return s < P1S ? ((s % HSTEPS) < KM ? 0 : 1) : 2; }        // 0 q|k, 1 v, 2 out
    static constexpr int ub1(int s) {
        if (s <= P1S) {
            const int hs = s / HSTEPS, r = s % HSTEPS;
            return UPH * hs + (r < KM ? 2 * r : 2 * KM + (r - KM));
        }
        const int v = s - P1S, c = v / HEADS, m = v % HEADS;
        return UPH * HEADS + (c < NCGF ? 2 * HEADS * c + 2 * m : 2 * HEADS * NCGF + (v - NCGF * HEADS));
    }
    static constexpr int NUNITS = ub1(NSTEP);
    static constexpr int ub(int s) { return s <= NSTEP ? ub1(s) : NUNITS + ub1(s - NSTEP); }
    // units issued once step s has freed its own (s = -1: before the first step): as far ahead as the ring allows
    static constexpr int hm(int s) { return ub(s + 1) + NU; }
    static_assert(NUNITS % NU == 0, "the ring position of a unit must not depend on the tile");

    // ---- every vector-memory instruction a wave issues, in order, so that the step's s_waitcnt vmcnt(N) is exact:
    // N = the instructions YOUNGER than the last DMA piece the step needs.  After the weight units of step s-1's
    // issue point come, in this order: the row pieces of the next tile issued there, the stores of a column group's
    // epilogue (end of step s-1), and the loads at the top of step s (q bias of the next head, output bias of the next
    // column group, residual rows).  All of them are unconditional and opaque to the optimiser (see opaque()).
    // The next tile's rows are requested right AFTER the first column group's epilogue and have landed before the
    // second one's: an epilogue consumes plain loads, in front of which hipcc waits vmcnt(0) — every DMA in flight at
    // that point, HBM-latency row pieces included, would be waited for.  Passes 0-2 in step RS0, passes 3-5 in RS0 + 1.
    static constexpr int RS0 = P1S + HEADS;
    static constexpr int xp(int s) { return s == RS0 || s == RS0 + 1 ? 3 * PPP : 0; }               // row pieces issued in step s
    // P0 passes normalised in the first half of step s: bit ps of the result.  Each at least two step waits after its
    // pieces were issued (the waits retire every older DMA), none in the last step of the second column group.
    static constexpr int p0_mask_of(int s) {
        return s == RS0 + 3 ? 0x03 : s == RS0 + 5 ? 0x04 : s == RS0 + 6 ? 0x08 : s == RS0 + 7 ? 0x10 : s == RS0 + 8 ? 0x20 : 0;
    }
    static constexpr int nt_of(int c) { return c < NCGF ? 8 : 4; }
    static constexpr int first_of(int c) { return P1S + c * HEADS; }
    static constexpr int n_bq(int s) { return kind(s) == 1 && s % HSTEPS == HSTEPS - 1 && s + 1 < P1S ? 4 : (s == NSTEP - 1 ? 4 : 0); }
    static constexpr int n_bias(int s) {      // bias of column group c: one step before the group starts
        for (int c = 0; c < NCG; ++c) if (s == first_of(c) - 1) return nt_of(c);
        return 0;
    }
    static constexpr int n_res(int s) {       // residual rows of column group c: at the top of its second step
        for (int c = 0; c < NCG; ++c) if (s == first_of(c) + 1) return 3 * nt_of(c) / 2;
        return 0;
    }
    static constexpr int n_st(int s) {        // stores of the epilogue that ran at the end of step s
        for (int c = 0; c < NCG; ++c) if (s == first_of(c) + HEADS - 1) return 3 * nt_of(c) / 2;
        return 0;
    }
    static constexpr int prev(int s) { return s == 0 ? NSTEP - 1 : s - 1; }
    static constexpr int younger(int s) { return xp(prev(s)) + n_st(prev(s)) + n_bq(s) + n_bias(s) + n_res(s); }
    // DMA pieces (two per wave and unit) + other instructions that may still be in flight when step s waits for the
    // units of step s+1
    static constexpr int inflight(int s) { return 2 * (hm(s - 1) - ub(s + 2)) + younger(s); }


    struct Frag {
        f16x8 w[8], x[3];
    };
    struct State {
        Frag fa, fb;
        f32x4 aq[3][4], ak[3][4], av[3][4];      // q^T, k^T: [d][row]; v: [row][d]
        f32x4 acc[3][8];                         // output projection: [col][row]
        f16x8 oh[HEADS][3][2];                   // the heads' outputs as B operands: [time slot][row tile][k step of the head]
        f32x4 bqv[4];
        f16x4 qh[3][4], kh[3][4];                // q, k of the current head as fp16 MFMA operands
        f16x4 pt[3][3];                          // P^T of the current head: [query tile][key tile]
        f16x8 rv[3][4];                          // residual rows of the current column group
        f32x4 bv[8];                             // its output bias, the projection's initial accumulator
        const gf16* resp[3];                     // this lane's residual rows (+ 8*q4), or the dump page
        gf16* outp[3];                           // this lane's output rows (+ 8*q4), or the dump page
    };

    const K7BP& p;
    char* smem;
    lchar* lds;                                  // the same, as an LDS pointer
    int lane, n16, q4, wave;
    // order in which this tile / the next tile walks the heads: rotated by the tile's position INSIDE its batch item.
    // Why rotate at all: see tattn_fused.hip (workgroups that share an XCD would otherwise ask for the same weight lines at
    // the same time and then not again for a whole tile).  Why by that position: the output projection sums the heads in
    // walking order, so the order must be a function of the data alone — a sample's result then has the same bits
    // wherever it sits in the batch, whatever the grid (tests/test_unet_gpu.py: batch independence).
    int rot, rotn;
    int woffb, xb[2];                            // LDS byte addresses: weight fragment base, row-image fragment bases (k step parity)
    int tb, tg, tbn, tgn;                        // this wave's (batch item, row group inside it) in this tile / the next tile
    int qpix[3], kpix[3][F4 ? 1 : 4];
#ifdef K7B_STAMPS
    unsigned long long tsum[16], tlast;
#endif

    __device__ __forceinline__ K7B(const K7BP& p_, char* s) : p(p_), smem(s), lds((lchar*)s) {}

    // a value the optimiser cannot see through: loads addressed with it are neither hoisted out of the tile loop nor
    // merged — every source-level load below is exactly one instruction per tile (the wait counts rely on it)
    __device__ static __forceinline__ int opaque(int v) {
        asm volatile("" : "+v"(v));
        return v;
    }

    // ---- weight stream
    template <int U>
    __device__ __forceinline__ const char* unit_src() const {
        constexpr int u = U % NUNITS;
        const int r = U >= NUNITS ? rotn : rot;          // (the stream runs on into the next tile)
        if constexpr (u < UPH * HEADS) {
            constexpr int hs = u / UPH, w = u % UPH;
            int h = hs + r;
            if (h >= HEADS) h -= HEADS;
            return p.wqkv + (size_t)(h * UPH + w) * UB;
        } else {
            constexpr int v = u - UPH * HEADS;
            if constexpr (v < 2 * HEADS * NCGF) {
                constexpr int cg = v / (2 * HEADS), rr = v % (2 * HEADS), hs = rr / 2, kk = rr % 2;
                int h = hs + r;
                if (h >= HEADS) h -= HEADS;
                return p.wo + (size_t)(cg * 2 * HEADS + h * 2 + kk) * UB;
            } else {
                constexpr int hs = v - 2 * HEADS * NCGF;
                int h = hs + r;
                if (h >= HEADS) h -= HEADS;
                return p.wo + (size_t)(2 * HEADS * NCGF + h) * UB;
            }
        }
    }
    template <int U>
    __device__ __forceinline__ void issue_unit() {
#ifdef K7B_ABL_NOWDMA      /* diagnostic builds (timing only, wrong results): what each part of the tile costs */
        return;
#endif
        const char* src = unit_src<U>() + (2 * wave) * 1024 + lane * 16;
        char* dst = smem + XB + (U % NU) * UB + (2 * wave) * 1024;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(src + 1024), (lptr_t)(dst + 1024), 16, 0, 0);
    }
    template <int U0, int U1>
    __device__ __forceinline__ void issue_range() {
        if constexpr (U0 < U1) {
            issue_unit<U0>();
            issue_range<U0 + 1, U1>();
        }
    }

    // global row index of local row r (0..47) of row group g4 of batch item b (clamped to row 0 when it does not exist)
    // and whether it exists; branch-free: every lane computes an address, the caller selects
    __device__ __forceinline__ bool grow_of(int b, int g4, int r, long long& gr) const {
        const int g = (r * p.fmagic) >> 16, f = r - g * p.F;
        const int pix = g4 * p.G + g;
        const bool ok = b < p.B && g4 < p.gpb && pix < p.S;
        gr = ok ? (long long)((b * p.F + f) * p.S + pix) : 0ll;
        return ok;
    }

    // ---- The row image.  LDS layout: [row block of 8 rows][column block of 64 channels][8 rows][128 bytes]; inside the
    // 128 bytes of a row the 16-byte chunk c sits at position c ^ (row & 7).  A (row block, column block) is 1 KB = one
    // LDS-DMA piece whose lane L carries row L >> 3, position L & 7: the per-lane SOURCE address does the row gather (a
    // row's frames are S rows apart) and the XOR; the column block is an immediate offset.  For the MFMA fragment reads
    // (16 rows x 16 bytes per lane quad) the XOR makes every ds_read_b128 lane group hit 16 distinct slots of the
    // 256-byte bank row; the in-place normalisation reads and writes whole pieces.
    //
    // rows of row group g4, pass PS (its 8 rows) -> the wave's part of the image, by LDS-DMA.  Rows that do not exist
    // read the zero page (their values must stay finite: a masked key still multiplies a zero probability).
    template <int PS>
    __device__ __forceinline__ void issue_rows(int b, int g4) {
#ifdef K7B_ABL_NOROWS
        return;
#endif
        const int r = 8 * PS + (lane >> 3);
        long long gr;
        const bool ok = grow_of(b, g4, r, gr);
        const char* rowp = (const char*)(p.t + gr * p.ldt) + (((lane & 7) ^ (lane >> 3)) << 4);
        const char* zp = (const char*)g_zero_page;
        const char* src = ok ? rowp : zp;
        const int cstep = ok ? 128 : 0;
        char* dst = smem + (wave * 6 + PS) * RBB;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + cb * cstep), (lptr_t)(dst + cb * 1024), 16, 0, 0);
    }
    // P0 of pass PS, in place: centre and scale the 8 rows (8 lanes per row, one chunk of every column block per lane).
    // fp32 statistics: the mean from the row sum, the variance from the squares of (x - mean_h) with mean_h the mean
    // rounded to fp16 (the differences are then exact to fp16 relative precision, whatever the mean) corrected by
    // (mean - mean_h)^2; the result x * rstd - mean * rstd is formed in fp32 and rounded once.  gamma / beta live in
    // the weights.
    template <int PS>
    __device__ __forceinline__ void p0_pass() {
#ifdef K7B_ABL_NOROWS
        return;
#endif
        lchar* base = lds + (wave * 6 + PS) * RBB + lane * 16;
        f16x8 v[NCB];
#pragma unroll
        for (int j = 0; j < NCB; ++j) v[j] = *(const lf16x8*)(base + 1024 * j);
        const f16x2 ones = (f16x2){(f16)1.f, (f16)1.f};
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < NCB; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) sum = __builtin_amdgcn_fdot2((f16x2){v[j][2 * e], v[j][2 * e + 1]}, ones, sum, false);
        sum = dpp_add8(sum);
        const float mean = sum * (1.0f / INNER);
        const f16 mh = (f16)mean;
        const float dm = mean - (float)mh;
        const f16x2 nm = (f16x2){(f16)-mh, (f16)-mh};
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < NCB; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f16x2 d = (f16x2){v[j][2 * e], v[j][2 * e + 1]} + nm;
                ss = __builtin_amdgcn_fdot2(d, d, ss, false);
            }
        ss = dpp_add8(ss);
        const float var = fmaxf(ss * (1.0f / INNER) - dm * dm, 0.f);
        const float rstd = rsqrtf(var + p.eps);
        const float nmr = -mean * rstd;
#pragma unroll
        for (int j = 0; j < NCB; ++j) {
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (f16)fmaf((float)v[j][e], rstd, nmr);
            *(lf16x8*)(base + 1024 * j) = o;
        }
    }

    // row-image fragment: row 16*i + n16 of the wave's group, chunk 4*ks + q4: row block 2i + (n16 >> 3), column block
    // ks >> 1, position (4*(ks & 1) + q4) ^ (n16 & 7): xb[ks & 1] + a constant
    __device__ __forceinline__ f16x8 xfrag(int i, int ks) const {
        return *(const lf16x8*)(lds + xb[ks & 1] + (2 * i * RBB + 1024 * (ks >> 1)));
    }
    __device__ __forceinline__ f16x8 wfrag(int unit, int tile) const {
        return *(const lf16x8*)(lds + woffb + ((unit % NU) * UB + tile * 1024));
    }

    // fragments of half KK (one MFMA k step) of step S (S may be NSTEP: step 0 of the next tile)
    template <int S_, int KK>
    __device__ __forceinline__ void read_half(Frag& f) const {
        constexpr int S = S_ % NSTEP;
        constexpr int kd = kind(S), u0 = ub(S);
        if constexpr (kd == 0) {
            constexpr int m = S % HSTEPS;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f.w[j] = wfrag(u0, 4 * KK + j);
                f.w[4 + j] = wfrag(u0 + 1, 4 * KK + j);
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) f.x[i] = xfrag(i, 2 * m + KK);
        } else if constexpr (kd == 1) {
            constexpr int m = S % HSTEPS - KM;
#pragma unroll
            for (int j = 0; j < 4; ++j) f.w[j] = wfrag(u0, 4 * KK + j);
#pragma unroll
            for (int i = 0; i < 3; ++i) f.x[i] = xfrag(i, 2 * m + KK);
        } else {
            constexpr int c = (S - P1S) / HEADS;
            if constexpr (c < NCGF) {
#pragma unroll
                for (int j = 0; j < 8; ++j) f.w[j] = wfrag(u0 + KK, j);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) f.w[j] = wfrag(u0, 4 * KK + j);
            }
        }
    }

    // the MFMAs of half KK of step S, with the memory instructions that were issued in front of them spread between
    template <int S, int KK, int NDS, int NVM>
    __device__ __forceinline__ void mma_half(State& st, const Frag& f) {
        constexpr int kd = kind(S);
        const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (kd == 0) {
            constexpr bool Z = (S % HSTEPS) == 0 && KK == 0;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    st.aq[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.w[j], f.x[i], Z ? st.bqv[j] : st.aq[i][j], 0, 0, 0);
                    st.ak[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.w[4 + j], f.x[i], Z ? zero4 : st.ak[i][j], 0, 0, 0);
                }
        } else if constexpr (kd == 1) {
            constexpr bool Z = (S % HSTEPS) == KM && KK == 0;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    st.av[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.x[i], f.w[j], Z ? zero4 : st.av[i][j], 0, 0, 0);
        } else {
            constexpr int v = S - P1S, c = v / HEADS, hs = v % HEADS;
            constexpr bool Z = hs == 0 && KK == 0;
            constexpr int NT = nt_of(c);
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    st.acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.w[j], st.oh[hs][i][KK], Z ? st.bv[j] : st.acc[i][j], 0, 0, 0);
        }
        // issue order (a compile-time directive): one memory instruction after every MFMA until they are used up
#pragma unroll
        for (int g = 0; g < NVM; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
#pragma unroll
        for (int g = 0; g < NDS; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
    }

    static constexpr int nds(int s_) {
        const int s = s_ % NSTEP;
        return kind(s) == 0 ? 11 : kind(s) == 1 ? 7 : nt_of((s - P1S) / HEADS);
    }

    // ---- attention of the head in time slot HS on the wave's 48 rows, all in registers.
    // After the q|k pass: q, k as fp16 operands (the accumulators are dead from here on, the v pass runs with 48).
    __device__ __forceinline__ void attn_cvt_qk(State& st) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    st.qh[i][j][e] = (f16)st.aq[i][j][e];
                    st.kh[i][j][e] = (f16)st.ak[i][j][e];
                }
    }
    // Scores of query tile QT: S^T = K Q^T (query on the lane), softmax over the keys -> P^T as fp16.  Runs inside the
    // head's v steps, beside their MFMAs: straight-line code (a branch would cut the step into scheduling regions) —
    // key tiles that share no pixel with the query tile are computed and masked like any other key.
    template <int QT>
    __device__ __forceinline__ void attn_scores(State& st) {
#ifdef K7B_ABL_NOATT
        for (int kt = 0; kt < 3; ++kt) st.pt[QT][kt] = st.qh[QT][kt] + st.kh[kt][QT];
        return;
#endif
        f32x4 sc[3];
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
            if (!tiles_meet(QT, kt)) continue;
            sc[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                sc[kt] = __builtin_amdgcn_mfma_f32_16x16x16f16(st.kh[kt][j], st.qh[QT][j], sc[kt], 0, 0, 0);
        }
        const int qp = opaque(qpix[QT]);      // (recomputed per head: 36 compare masks kept across the heads do not fit the SGPRs)
        float mx = -1.0e30f;
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
            if (!tiles_meet(QT, kt)) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (!tiles_pure(QT, kt))
                    sc[kt][e] = kpix[kt][F4 ? 0 : e] == qp ? sc[kt][e] : -1.0e30f;      // keys of other pixels: exp2 below gives exactly 0
                mx = fmaxf(mx, sc[kt][e]);
            }
        }
        mx = quad_max(mx);
        float rs = 0.f;
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
            if (!tiles_meet(QT, kt)) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sc[kt][e] = __builtin_amdgcn_exp2f(sc[kt][e] - mx);         // (the scale is in W_q)
                rs += sc[kt][e];
            }
        }
        rs = quad_sum(rs);
        const float inv = 1.0f / rs;
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
            if (!tiles_meet(QT, kt)) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) st.pt[QT][kt][e] = (f16)(sc[kt][e] * inv);
        }
    }
    // After the v pass: O^T[d][query] = V^T P^T: lane = query row, registers e = d 16*dt + 4*q4 + e.  Two d tiles side
    // by side are one B operand of the output projection (k index of W_o permuted to match: packing.pack_k7b).
    template <int HS>
    __device__ __forceinline__ void attn_pv(State& st) {
#ifdef K7B_ABL_NOATT
        for (int qt = 0; qt < 3; ++qt)
            for (int kk = 0; kk < 2; ++kk)
                for (int e = 0; e < 8; ++e) st.oh[HS][qt][kk][e] = (f16)st.av[qt][2 * kk + (e >> 2)][e & 3] + st.pt[qt][kk][e & 3];
        return;
#endif
        f16x4 vh[3][4];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) vh[i][j][e] = (f16)st.av[i][j][e];
#pragma unroll
        for (int qt = 0; qt < 3; ++qt)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                f32x4 o0 = (f32x4){0.f, 0.f, 0.f, 0.f}, o1 = o0;
#pragma unroll
                for (int kt = 0; kt < 3; ++kt) {
                    if (!tiles_meet(qt, kt)) continue;
                    o0 = __builtin_amdgcn_mfma_f32_16x16x16f16(vh[kt][2 * kk], st.pt[qt][kt], o0, 0, 0, 0);
                    o1 = __builtin_amdgcn_mfma_f32_16x16x16f16(vh[kt][2 * kk + 1], st.pt[qt][kt], o1, 0, 0, 0);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    st.oh[HS][qt][kk][e] = (f16)o0[e];
                    st.oh[HS][qt][kk][4 + e] = (f16)o1[e];
                }
            }
    }

    // q bias of the head in time slot HS: the initial accumulator of its q (d = 16*j + 4*q4 + e on the registers)
    template <int HS, bool NEXT_TILE = false>
    __device__ __forceinline__ void load_bq(State& st) {
        int h = HS + (NEXT_TILE ? rotn : rot);
        if (h >= HEADS) h -= HEADS;
        const int o = opaque(h * 64 + 4 * q4);
#pragma unroll
        for (int j = 0; j < 4; ++j) st.bqv[j] = *(const gf32x4*)((const __attribute__((address_space(1))) float*)p.bq + o + 16 * j);
    }
    // output bias of column group C: the initial accumulator of the projection (tile 2a + jj, register e: column
    // 32a + 8*q4 + 4*jj + e of the group)
    template <int C>
    __device__ __forceinline__ void load_bias(State& st) {
        const int o = opaque(C * 128 + 8 * q4);
#pragma unroll
        for (int j = 0; j < nt_of(C); ++j) st.bv[j] = *(const gf32x4*)((const __attribute__((address_space(1))) float*)p.bo2 + o + 32 * (j / 2) + 4 * (j % 2));
    }
    // residual rows of column group C: requested at the top of the group's second K step (HBM latency), consumed after it
    template <int C>
    __device__ __forceinline__ void load_residual(State& st) {
#ifdef K7B_ABL_NOEPI
        for (int i = 0; i < 3; ++i) for (int a = 0; a < 4; ++a) st.rv[i][a] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
        return;
#endif
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const gf16* src = st.resp[i] + opaque(0);
#pragma unroll
            for (int a = 0; a < nt_of(C) / 2; ++a) st.rv[i][a] = *(const gf16x8*)(src + C * 128 + 32 * a);
        }
    }
    // tile pair (2a, 2a+1) gives this lane 8 consecutive columns 32a + 8*q4 .. +7 of row n16 (+16i).  The projection
    // (bias included: it was the initial accumulator) is rounded to fp16 and the residual added in fp16 — the
    // reference's order (to_out returns fp16, `attn_output + hidden_states` is an fp16 add).
    template <int C>
    __device__ __forceinline__ void epilogue(State& st) {
#ifdef K7B_ABL_NOEPI
        for (int a = 0; a < nt_of(C); ++a) for (int i = 0; i < 3; ++i) asm volatile("" ::"v"(st.acc[i][a]));
        return;
#endif
#pragma unroll
        for (int a = 0; a < nt_of(C) / 2; ++a)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                f16x8 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = (f16)st.acc[i][2 * a][e];
                    o[4 + e] = (f16)st.acc[i][2 * a + 1][e];
                }
                o = o + st.rv[i][a];
                *(gf16x8*)(st.outp[i] + C * 128 + 32 * a) = o;
            }
    }

    // ---- one step of the tile.  At its top the fragments of its first half are in registers (st.fa).
    template <int S>
    __device__ __forceinline__ void step(State& st) {
        constexpr int kd = kind(S);
        // loads of later steps, requested here (counted by younger(S))
        if constexpr (n_bq(S) > 0) load_bq<(S == NSTEP - 1 ? 0 : S / HSTEPS + 1), S == NSTEP - 1>(st);
        if constexpr (n_bias(S) > 0) load_bias<(S + 1 - P1S) / HEADS>(st);
        if constexpr (n_res(S) > 0) load_residual<(S - P1S) / HEADS>(st);
        // vector work that rides on this step's MFMAs: the scores of the head (v steps 0..2), the next tile's rows
        if constexpr (kd == 1 && S % HSTEPS - KM < 3) attn_scores<S % HSTEPS - KM>(st);
        p0_passes<p0_mask_of(S)>(std::make_integer_sequence<int, NPS>{});
        // second half's fragments behind the first half's MFMAs
        read_half<S, 1>(st.fb);
        mma_half<S, 0, nds(S), 0>(st, st.fa);
        // the units of step S+1 have landed for everyone, and nobody reads the units of step S any more
        __builtin_amdgcn_sched_barrier(0);
        K7B_T(4 * kd + 0)
        wait_vm<inflight(S)>();
        __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): my reads of step S's units are done (builtin: the
        asm volatile("" ::: "memory");           // compiler then knows st.fb is valid)
        wg_barrier();
        K7B_T(4 * kd + 1)
        issue_range<hm(S - 1), hm(S)>();
        if constexpr (xp(S) > 0) {
            issue_rows<3 * (S - RS0)>(tbn, tgn);
            issue_rows<3 * (S - RS0) + 1>(tbn, tgn);
            issue_rows<3 * (S - RS0) + 2>(tbn, tgn);
        }
        read_half<S + 1, 0>(st.fa);
        mma_half<S, 1, nds(S + 1), 2 * (hm(S) - hm(S - 1)) + xp(S)>(st, st.fb);
        __builtin_amdgcn_sched_barrier(0);
        K7B_T(4 * kd + 2)
        if constexpr (kd == 0 && S % HSTEPS == KM - 1) attn_cvt_qk(st);
        if constexpr (kd == 1 && S % HSTEPS == HSTEPS - 1) attn_pv<S / HSTEPS>(st);
        if constexpr (n_st(S) > 0) epilogue<(S - P1S) / HEADS>(st);
#ifdef K7B_STAMPS
        if constexpr ((kd == 0 && S % HSTEPS == KM - 1) || (kd == 1 && S % HSTEPS == HSTEPS - 1) || n_st(S) > 0) { asm volatile("" ::"v"(st.oh[0][0][0]), "v"(st.qh[0][0]), "v"(st.kh[2][3])); K7B_T(4 * kd + 3) }
#endif
    }
    template <int... S>
    __device__ __forceinline__ void steps(State& st, std::integer_sequence<int, S...>) {
        (step<S>(st), ...);
    }
    template <int MASK, int... PS>
    __device__ __forceinline__ void p0_passes(std::integer_sequence<int, PS...>) {
        ((MASK >> PS & 1 ? p0_pass<PS>() : void()), ...);
    }
    template <int... PS>
    __device__ __forceinline__ void first_rows(std::integer_sequence<int, PS...>) {
        (issue_rows<PS>(tb, tg), ...);
        wait_vm<0>();
        (p0_pass<PS>(), ...);
    }

    __device__ __forceinline__ void set_lane_constants() {
        n16 = lane & 15;
        q4 = lane >> 4;
        const int g = (0x1320 >> (4 * (n16 >> 2))) & 3;          // g = [0, 2, 3, 1][n >> 2]
        woffb = XB + n16 * 64 + ((q4 ^ g) << 4);
        const int rr = n16 & 7, xrow = (wave * 6 + (n16 >> 3)) * RBB + rr * 128;
        xb[0] = xrow + ((q4 ^ rr) << 4);
        xb[1] = xrow + (((4 + q4) ^ rr) << 4);
        // pixel of my query rows / key rows inside the 48-row group (for the block-diagonal mask)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            qpix[i] = ((16 * i + n16) * p.fmagic) >> 16;
#pragma unroll
            for (int e = 0; e < (F4 ? 1 : 4); ++e) kpix[i][e] = ((16 * i + 4 * q4 + e) * p.fmagic) >> 16;
        }
    }

    // tile -> (batch item, this wave's row group inside it, head rotation); tiles are aligned to batch items
    __device__ __forceinline__ void set_tile(int tile, int& b, int& g4, int& r) const {
        b = tile / p.tps;
        const int tt = tile - b * p.tps;
        g4 = tt * 4 + wave;
        r = tt % HEADS;
    }

    __device__ __forceinline__ void run() {
        const int tid = threadIdx.x;
        lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        set_lane_constants();

        // ---- first tile: the weight stream, the tile's rows, the first q bias; everything has landed before the first
        // step (so its counted wait finds nothing outstanding), the rows are centred and scaled in place
        int tile = blockIdx.x;
        set_tile(tile, tb, tg, rot);
        rotn = rot;
        State st;
        issue_range<0, NU>();
        load_bq<0>(st);
        first_rows(std::make_integer_sequence<int, NPS>{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wg_barrier();
        read_half<0, 0>(st.fa);
#ifdef K7B_STAMPS
        for (int i = 0; i < 16; ++i) tsum[i] = 0;
        tlast = __builtin_amdgcn_s_memtime();
        const unsigned long long tstart = tlast;
#endif

        for (;;) {
            // Everything below is straight-line code per tile.  The lane constants are made opaque once per tile: the
            // optimiser otherwise hoists every address that depends only on them (hundreds) out of this loop and
            // spills them around it.
            asm volatile("" : "+v"(lane));
            asm volatile("" : "+s"(wave));
            set_lane_constants();
            const int next = tile + gridDim.x;
            set_tile(next, tbn, tgn, rotn);      // (past the last tile: every row reads the zero page)
            // rows that do not exist (last tile) read and write the dump page: every tile issues the same instructions
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                long long gr;
                const bool ok = grow_of(tb, tg, 16 * i + n16, gr);
                gf16* dump = (gf16*)g_dump_page + lane * 8;
                const gf16* rp = (const gf16*)p.t + gr * p.ldt + 8 * q4;
                gf16* op = (gf16*)p.out + gr * p.ldo + 8 * q4;
                st.resp[i] = ok ? rp : dump;
                st.outp[i] = ok ? op : dump;
            }
            steps(st, std::make_integer_sequence<int, NSTEP>{});
            if (next >= p.ntiles) break;
            tile = next;
            tb = tbn;
            tg = tgn;
            rot = rotn;
        }
#ifdef K7B_STAMPS
        tsum[12] = __builtin_amdgcn_s_memtime() - tstart;
        if (lane == 0 && wave == 0 && blockIdx.x < 1024)
            for (int i = 0; i < 16; ++i) g_k7b_stamps[blockIdx.x * 16 + i] = tsum[i];
#endif
        wait_vm<0>();        // (the stream ran on into a tile that does not exist: let its copies land before the LDS is released)
    }
};

template <int INNER, int FS>
__global__ __launch_bounds__(256, 1) void tattn2_kernel(const K7BP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    K7B<INNER, FS> k(p, smem);
    k.run();
}

}  // namespace

extern "C" int vdx_temporal_attn_block2_supported(int inner, int F) {
    return inner == 320 && F >= 1 && F <= 48 && 48 % F == 0;
}
// bytes of the packed blob (vdx/packing.py pack_k7b): q|k|v units, output-projection units, fp32 q bias, fp32 output bias
extern "C" size_t vdx_temporal_attn_block2_pack_bytes(int inner) {
    if (inner != 320) return 0;
    return (size_t)K7B<320, 1>::NUNITS * K7B<320, 1>::UB + 2 * 320 * sizeof(float);
}

extern "C" int vdx_temporal_attn_block2_f16(const void* t, int ldt, const void* packed, float eps, void* out, int ldo,
                                            int B, int F, int HW, int inner, vdx_stream_t stream) {
    VDX_CHECK(t && packed && out, "temporal_attn_block2: null pointer");
    VDX_CHECK(B > 0 && F > 0 && HW > 0, "temporal_attn_block2: empty problem");
    VDX_CHECK(vdx_temporal_attn_block2_supported(inner, F), "temporal_attn_block2: inner=%d F=%d not supported (inner 320, F | 48)", inner, F);
    VDX_CHECK(ldt % 8 == 0 && ldo % 8 == 0 && ldt >= inner && ldo >= inner, "temporal_attn_block2: bad leading dims");
    VDX_CHECK((long long)B * F * HW < (1ll << 31), "temporal_attn_block2: too many rows");
    VDX_CHECK(((uintptr_t)t % 16 == 0) && ((uintptr_t)out % 16 == 0) && ((uintptr_t)packed % 16 == 0), "temporal_attn_block2: pointers must be 16-byte aligned");
    typedef K7B<320, 1> T;
    K7BP p;
    p.t = (const f16*)t; p.out = (f16*)out;
    p.wqkv = (const char*)packed;
    p.wo = p.wqkv + (size_t)T::UPH * T::HEADS * T::UB;
    p.bq = (const float*)(p.wqkv + (size_t)T::NUNITS * T::UB);
    p.bo2 = p.bq + 320;
    p.ldt = ldt; p.ldo = ldo; p.B = B; p.F = F; p.S = HW;
    p.G = 48 / F;
    p.gpb = (HW + p.G - 1) / p.G;
    p.tps = (p.gpb + 3) / 4;
    p.ntiles = B * p.tps;
    p.fmagic = (65536 + F - 1) / F;
    p.eps = eps;
    constexpr int lds = T::XB + T::NU * T::UB;
    void (*kern)(const K7BP) = F == 24 ? tattn2_kernel<320, 24> : F == 16 ? tattn2_kernel<320, 16> : F == 12 ? tattn2_kernel<320, 12>
                               : F % 4 == 0 ? tattn2_kernel<320, 4> : tattn2_kernel<320, 1>;
    static const hipError_t attr_rc = [] {
        hipError_t e = hipSuccess;
        for (auto k : {tattn2_kernel<320, 24>, tattn2_kernel<320, 16>, tattn2_kernel<320, 12>, tattn2_kernel<320, 4>, tattn2_kernel<320, 1>}) {
            const hipError_t r = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (r != hipSuccess) e = r;
        }
        return e;
    }();
    if (attr_rc != hipSuccess) return vdx_fail("temporal_attn_block2: cannot reserve %d bytes of LDS", lds);
    // persistent grid: every workgroup walks the same number of tiles (+-1), one workgroup per CU at most
    const int ncu = vdx_grid_cus();
    const int rounds = (p.ntiles + ncu - 1) / ncu;
    const int grid = (p.ntiles + rounds - 1) / rounds;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, (hipStream_t)stream, p);
    return vdx_launch_status("vdx_temporal_attn_block2_f16");
}

// Lab variants of this translation unit (phase stamps, ablations: timing only, some give WRONG results) are compiled in only
// under the macros below; a library that carries one says so through vdx_build_flags() and vdx/_lib.py refuses to load it
// as the product.
extern "C" int vdx_lab_tattn2(void) {
#if defined(K7B_STAMPS) || defined(K7B_ABL_NOROWS) || defined(K7B_ABL_NOEPI) || defined(K7B_ABL_NOATT) || defined(K7B_ABL_NOWDMA)
    return 8;
#else
    return 0;
#endif
}
