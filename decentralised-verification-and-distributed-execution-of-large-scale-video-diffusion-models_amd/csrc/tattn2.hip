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
// Geometry (inner 320): 4 waves, one per SIMD, each owning a row group of 48 rows = G pixels x F frames (G = 48 / F)
// for the whole tile; nothing a wave reads of the row image is written by another wave.
#include "vdx_common.h"
#include <utility>

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

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
    int ngroups;         // B * gpb
    int fmagic;          // ceil(65536 / F): row / F == (row * fmagic) >> 16 for row < 64
    float eps;
};

template <int INNER>
struct K7B {
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
    static constexpr bool SW16 = (INNER / 8) % 16 == 0;
    static_assert(KS % 2 == 0 && (INNER % 128 == 0 || INNER % 128 == 64), "geometry");
    static_assert(XB + NU * UB <= 160 * 1024, "LDS budget");

    // ---- the static schedule of a tile: step s consumes units [ub(s), ub(s+1)) of the weight stream
    static constexpr int kind(int s) { return s < P1S ? ((s % HSTEPS) < KM ? 0 : 1) : 2; }        // 0 q|k, 1 v, 2 out
    static constexpr int ub(int s) {
        if (s <= P1S) {
            const int hs = s / HSTEPS, r = s % HSTEPS;
            return UPH * hs + (r < KM ? 2 * r : 2 * KM + (r - KM));
        }
        const int v = s - P1S, c = v / HEADS, m = v % HEADS;
        return UPH * HEADS + (c < NCGF ? 2 * HEADS * c + 2 * m : 2 * HEADS * NCGF + (v - NCGF * HEADS));
    }
    static constexpr int NUNITS = ub(NSTEP);
    // units issued once step s has freed its own (s = -1: before the first step): as far ahead as the ring allows
    static constexpr int hm(int s) { return ub(s + 1) + NU < NUNITS ? ub(s + 1) + NU : NUNITS; }
    // DMA pieces (two per wave and unit) that may still be in flight when step s waits for the units of step s+1
    static constexpr int inflight(int s) { return hm(s - 1) > ub(s + 2) ? 2 * (hm(s - 1) - ub(s + 2)) : 0; }
    static_assert(NUNITS % NU == 0, "the ring position of a unit must not depend on the tile");

    __device__ static __forceinline__ int swz(int row) { return SW16 ? (row & 15) : ((row >> 1) & 7); }

    struct Frag {
        f16x8 w[8], x[3];
    };
    struct State {
        Frag fa, fb;
        f32x4 aq[3][4], ak[3][4], av[3][4];      // q^T, k^T: [d][row]; v: [row][d]
        f32x4 acc[3][8];                         // output projection: [col][row]
        f16x8 oh[HEADS][3][2];                   // the heads' outputs as B operands: [time slot][row tile][k step of the head]
        f32x4 bqv[4];
        f16x4 pt[3][3];                          // P^T of the current head: [query tile][key tile]
        f16x8 rv[3][4];                          // residual rows of the current column group
        f32x4 bv[4][2];                          // its output bias
        long long grow[3];
    };

    const K7BP& p;
    char* smem;
    int lane, n16, q4, wave, rot, woff, gi;
    int qpix[3], kpix[3][4], need;

    __device__ __forceinline__ K7B(const K7BP& p_, char* s) : p(p_), smem(s) {}

    // ---- weight stream
    template <int U>
    __device__ __forceinline__ const char* unit_src() const {
        if constexpr (U < UPH * HEADS) {
            constexpr int hs = U / UPH, w = U % UPH;
            int h = hs + rot;
            if (h >= HEADS) h -= HEADS;
            return p.wqkv + (size_t)(h * UPH + w) * UB;
        } else {
            constexpr int v = U - UPH * HEADS;
            if constexpr (v < 2 * HEADS * NCGF) {
                constexpr int cg = v / (2 * HEADS), r = v % (2 * HEADS), hs = r / 2, kk = r % 2;
                int h = hs + rot;
                if (h >= HEADS) h -= HEADS;
                return p.wo + (size_t)(cg * 2 * HEADS + h * 2 + kk) * UB;
            } else {
                constexpr int hs = v - 2 * HEADS * NCGF;
                int h = hs + rot;
                if (h >= HEADS) h -= HEADS;
                return p.wo + (size_t)(2 * HEADS * NCGF + h) * UB;
            }
        }
    }
    template <int U>
    __device__ __forceinline__ void issue_unit() {
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

    __device__ __forceinline__ f16x8 xfrag(int i, int ks) const {
        const int row = wave * 48 + 16 * i + n16;
        return *(const f16x8*)(smem + row * RB + (((4 * ks + q4) ^ swz(row)) << 4));
    }
    __device__ __forceinline__ f16x8 wfrag(int unit, int tile) const {
        return *(const f16x8*)(smem + XB + (unit % NU) * UB + tile * 1024 + woff);
    }

    // fragments of half KK (one MFMA k step) of step S
    template <int S, int KK>
    __device__ __forceinline__ void read_half(Frag& f) const {
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
        int nm = 0;
        if constexpr (kd == 0) {
            constexpr bool Z = (S % HSTEPS) == 0 && KK == 0;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    st.aq[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.w[j], f.x[i], Z ? st.bqv[j] : st.aq[i][j], 0, 0, 0);
                    st.ak[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.w[4 + j], f.x[i], Z ? zero4 : st.ak[i][j], 0, 0, 0);
                }
            nm = 24;
        } else if constexpr (kd == 1) {
            constexpr bool Z = (S % HSTEPS) == KM && KK == 0;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    st.av[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.x[i], f.w[j], Z ? zero4 : st.av[i][j], 0, 0, 0);
            nm = 12;
        } else {
            constexpr int v = S - P1S, c = v / HEADS, hs = v % HEADS;
            constexpr bool Z = hs == 0 && KK == 0;
            constexpr int NT = c < NCGF ? 8 : 4;
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    st.acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.w[j], st.oh[hs][i][KK], Z ? zero4 : st.acc[i][j], 0, 0, 0);
            nm = 3 * NT;
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
        (void)nm;
    }

    static constexpr int nds(int s) { return s >= NSTEP ? 0 : kind(s) == 0 ? 11 : kind(s) == 1 ? 7 : ((s - P1S) / HEADS < NCGF ? 8 : 4); }

    // global row of local row r (0..47) of this wave's row group, or -1
    __device__ __forceinline__ long long grow_of(int r) const {
        const int g = (r * p.fmagic) >> 16, f = r - g * p.F;
        const int b = gi / p.gpb, pix = (gi - b * p.gpb) * p.G + g;
        if (gi >= p.ngroups || pix >= p.S) return -1;
        return ((long long)b * p.F + f) * p.S + pix;
    }

    // ---- attention of the head in time slot HS on the wave's 48 rows, all in registers.
    // Part 1, after the q|k pass: S^T = K Q^T (query on the lane), softmax over the keys -> P^T as fp16 (18 registers);
    // the q and k accumulators are dead from here on, the v pass runs with 48.
    template <int HS>
    __device__ __forceinline__ void attn_scores(State& st) {
        f16x4 qh[3][4], kh[3][4];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    qh[i][j][e] = (f16)st.aq[i][j][e];
                    kh[i][j][e] = (f16)st.ak[i][j][e];
                }
#pragma unroll
        for (int qt = 0; qt < 3; ++qt) {
            f32x4 sc[3];
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) {
                sc[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if ((need >> (3 * qt + kt)) & 1) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        sc[kt] = __builtin_amdgcn_mfma_f32_16x16x16f16(kh[kt][j], qh[qt][j], sc[kt], 0, 0, 0);
                }
            }
            float mx = -1.0e30f;
            bool ok[3][4];
#pragma unroll
            for (int kt = 0; kt < 3; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool pure = (need >> (9 + 3 * qt + kt)) & 1;       // wave-uniform
                    ok[kt][e] = pure || (((need >> (3 * qt + kt)) & 1) && kpix[kt][e] == qpix[qt]);
                    mx = fmaxf(mx, ok[kt][e] ? sc[kt][e] : -1.0e30f);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float rs = 0.f;
#pragma unroll
            for (int kt = 0; kt < 3; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sc[kt][e] = ok[kt][e] ? __builtin_amdgcn_exp2f(sc[kt][e] - mx) : 0.f;     // (the scale is in W_q)
                    rs += sc[kt][e];
                }
            rs += __shfl_xor(rs, 16, 64);
            rs += __shfl_xor(rs, 32, 64);
            const float inv = 1.0f / rs;
#pragma unroll
            for (int kt = 0; kt < 3; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) st.pt[qt][kt][e] = (f16)(sc[kt][e] * inv);
        }
    }
    // Part 2, after the v pass: O^T[d][query] = V^T P^T: lane = query row, registers e = d 16*dt + 4*q4 + e.  Two d tiles
    // side by side are one B operand of the output projection (k index of W_o permuted to match: packing.pack_k7b).
    template <int HS>
    __device__ __forceinline__ void attn_pv(State& st) {
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
                for (int kt = 0; kt < 3; ++kt)
                    if ((need >> (3 * qt + kt)) & 1) {
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

    // q bias of the head in time slot HS (its d = 16*j + 4*q4 + e on the accumulator registers)
    template <int HS>
    __device__ __forceinline__ void load_bq(State& st) {
        int h = HS + rot;
        if (h >= HEADS) h -= HEADS;
#pragma unroll
        for (int j = 0; j < 4; ++j) st.bqv[j] = *(const f32x4*)(p.bq + h * 64 + 16 * j + 4 * q4);
    }

    // residual rows of column group C: requested at the start of the group's K loop, consumed after it
    template <int C>
    __device__ __forceinline__ void load_residual(State& st) {
        constexpr int NT = C < NCGF ? 8 : 4;
        const f16* zp = (const f16*)g_zero_page;
        const int cb = C * 128 + 8 * q4;
#pragma unroll
        for (int a = 0; a < NT / 2; ++a)
#pragma unroll
            for (int i = 0; i < 3; ++i)
                st.rv[i][a] = *(const f16x8*)(st.grow[i] >= 0 ? p.t + st.grow[i] * p.ldt + cb + 32 * a : zp);
    }
    template <int C>
    __device__ __forceinline__ void load_bias(State& st) {
        constexpr int NT = C < NCGF ? 8 : 4;
        const int cb = C * 128 + 8 * q4;
#pragma unroll
        for (int a = 0; a < NT / 2; ++a) {
            st.bv[a][0] = *(const f32x4*)(p.bo2 + cb + 32 * a);
            st.bv[a][1] = *(const f32x4*)(p.bo2 + cb + 32 * a + 4);
        }
    }
    // tile pair (2a, 2a+1) gives this lane 8 consecutive columns 32a + 8*q4 .. +7 of row n16 (+16i)
    template <int C>
    __device__ __forceinline__ void epilogue(State& st) {
        constexpr int NT = C < NCGF ? 8 : 4;
        const int cb = C * 128 + 8 * q4;
#pragma unroll
        for (int a = 0; a < NT / 2; ++a) {
            const f32x4 b0 = st.bv[a][0], b1 = st.bv[a][1];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                f16x8 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = (f16)(st.acc[i][2 * a][e] + b0[e] + (float)st.rv[i][a][e]);
                    o[4 + e] = (f16)(st.acc[i][2 * a + 1][e] + b1[e] + (float)st.rv[i][a][4 + e]);
                }
                if (st.grow[i] >= 0) *(f16x8*)(p.out + st.grow[i] * p.ldo + cb + 32 * a) = o;
            }
        }
    }

    // ---- one step of the tile.  At its top the fragments of its first half are in registers (st.fa).
    template <int S>
    __device__ __forceinline__ void step(State& st) {
        constexpr int kd = kind(S);
        // q bias of the NEXT head: requested a step before its first MFMAs start from it
        if constexpr (kd == 1 && S % HSTEPS == HSTEPS - 1 && S + 1 < P1S) load_bq<S / HSTEPS + 1>(st);
        if constexpr (kd == 2 && (S - P1S) % HEADS == 0) load_residual<(S - P1S) / HEADS>(st);
        if constexpr (kd == 2 && (S - P1S) % HEADS == HEADS - 1) load_bias<(S - P1S) / HEADS>(st);
        // second half's fragments behind the first half's MFMAs
        read_half<S, 1>(st.fb);
        mma_half<S, 0, nds(S), 0>(st, st.fa);
        // the units of step S+1 have landed for everyone, and nobody reads the units of step S any more
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (S + 1 < NSTEP) {
            wait_vm<inflight(S)>();
            __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): my reads of step S's units are done (builtin: the
            asm volatile("" ::: "memory");           // compiler then knows st.fb is valid)
            wg_barrier();
            issue_range<hm(S - 1), hm(S)>();
            read_half<S + 1, 0>(st.fa);
        }
        mma_half<S, 1, nds(S + 1), 2 * (hm(S) - hm(S - 1))>(st, st.fb);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (kd == 0 && S % HSTEPS == KM - 1) attn_scores<S / HSTEPS>(st);
        if constexpr (kd == 1 && S % HSTEPS == HSTEPS - 1) attn_pv<S / HSTEPS>(st);
        if constexpr (kd == 2 && (S - P1S) % HEADS == HEADS - 1) epilogue<(S - P1S) / HEADS>(st);
    }
    template <int... S>
    __device__ __forceinline__ void steps(State& st, std::integer_sequence<int, S...>) {
        (step<S>(st), ...);
    }

    __device__ __forceinline__ void run() {
        const int tid = threadIdx.x;
        lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        n16 = lane & 15;
        q4 = lane >> 4;
        {
            const int g = (0x1320 >> (4 * (n16 >> 2))) & 3;      // g = [0, 2, 3, 1][n >> 2]
            woff = n16 * 64 + ((q4 ^ g) << 4);
        }
        gi = blockIdx.x * 4 + wave;                             // this wave's row group
        rot = (blockIdx.x >> 3) % HEADS;                        // blocks b and b+8 share an XCD (see tattn_fused.hip)
        const f16* zp = (const f16*)g_zero_page;

        // the weight stream starts before the rows are even loaded
        issue_range<0, hm(-1)>();
        State st;
        load_bq<0>(st);

        // ---- P0: rows -> centred and scaled (fp32 statistics, two passes over registers) -> X (fp16, swizzled) in LDS.
        // 8 lanes per row, CPL chunks of 8 channels per lane.  gamma / beta live in the weights.
        {
            const int sub = lane & 7;
            constexpr int NPS = 48 / 8;
            f16x8 v[NPS][CPL];
            long long grs[NPS];
#pragma unroll
            for (int ps = 0; ps < NPS; ++ps) {
                const int r = 8 * ps + (lane >> 3);
                grs[ps] = grow_of(r);
                const f16* src = grs[ps] >= 0 ? p.t + grs[ps] * p.ldt + 8 * sub : zp;
                const int stp = grs[ps] >= 0 ? 64 : 0;
#pragma unroll
                for (int j = 0; j < CPL; ++j) v[ps][j] = *(const f16x8*)(src + j * stp);
            }
#pragma unroll
            for (int ps = 0; ps < NPS; ++ps) {
                const int r = 8 * ps + (lane >> 3);
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
                const float nmr = -mean * rstd;
                const int row = wave * 48 + r;
                char* dst = smem + row * RB;
                const int sw = swz(row);
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    f16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = grs[ps] >= 0 ? (f16)fmaf((float)v[ps][j][e], rstd, nmr) : (f16)0.f;
                    *(f16x8*)(dst + (((sub + 8 * j) ^ sw) << 4)) = o;
                }
            }
        }

        // pixel of my query rows / key rows inside the 48-row group (for the block-diagonal mask)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            qpix[i] = ((16 * i + n16) * p.fmagic) >> 16;
#pragma unroll
            for (int e = 0; e < 4; ++e) kpix[i][e] = ((16 * i + 4 * q4 + e) * p.fmagic) >> 16;
        }
        // 16-row tiles of the group that share no pixel need no score tile at all (F = 16: only the diagonal; F = 24:
        // 7 of 9): bit 3*qt + kt of `need` says query tile qt has a pixel in common with key tile kt (wave-uniform);
        // bit 9 + 3*qt + kt: both tiles lie inside ONE pixel, no mask
        need = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int alo = 16 * a / p.F, ahi = (16 * a + 15) / p.F, blo = 16 * b / p.F, bhi = (16 * b + 15) / p.F;
                if (!(ahi < blo || bhi < alo)) need |= 1 << (3 * a + b);
                if (alo == ahi && blo == bhi && alo == blo) need |= 1 << (9 + 3 * a + b);
            }
        need = __builtin_amdgcn_readfirstlane(need);

#pragma unroll
        for (int i = 0; i < 3; ++i) st.grow[i] = grow_of(16 * i + n16);

        // the units of step 0 have landed for everyone (X is private to the wave: its own writes only need lgkmcnt)
        wait_vm<2 * (hm(-1) - ub(1))>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wg_barrier();
        read_half<0, 0>(st.fa);
        steps(st, std::make_integer_sequence<int, NSTEP>{});
    }
};

template <int INNER>
__global__ __launch_bounds__(256, 1) void tattn2_kernel(const K7BP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    K7B<INNER> k(p, smem);
    k.run();
}

}  // namespace

extern "C" int vdx_temporal_attn_block2_supported(int inner, int F) {
    return inner == 320 && F >= 1 && F <= 48 && 48 % F == 0;
}
// bytes of the packed blob (vdx/packing.py pack_k7b): q|k|v units, output-projection units, fp32 q bias, fp32 output bias
extern "C" size_t vdx_temporal_attn_block2_pack_bytes(int inner) {
    if (inner != 320) return 0;
    return (size_t)K7B<320>::NUNITS * K7B<320>::UB + 2 * 320 * sizeof(float);
}

extern "C" int vdx_temporal_attn_block2_f16(const void* t, int ldt, const void* packed, float eps, void* out, int ldo,
                                            int B, int F, int HW, int inner, vdx_stream_t stream) {
    VDX_CHECK(t && packed && out, "temporal_attn_block2: null pointer");
    VDX_CHECK(B > 0 && F > 0 && HW > 0, "temporal_attn_block2: empty problem");
    VDX_CHECK(vdx_temporal_attn_block2_supported(inner, F), "temporal_attn_block2: inner=%d F=%d not supported (inner 320, F | 48)", inner, F);
    VDX_CHECK(ldt % 8 == 0 && ldo % 8 == 0 && ldt >= inner && ldo >= inner, "temporal_attn_block2: bad leading dims");
    VDX_CHECK((long long)B * F * HW < (1ll << 31), "temporal_attn_block2: too many rows");
    VDX_CHECK(((uintptr_t)t % 16 == 0) && ((uintptr_t)out % 16 == 0) && ((uintptr_t)packed % 16 == 0), "temporal_attn_block2: pointers must be 16-byte aligned");
    typedef K7B<320> T;
    K7BP p;
    p.t = (const f16*)t; p.out = (f16*)out;
    p.wqkv = (const char*)packed;
    p.wo = p.wqkv + (size_t)T::UPH * T::HEADS * T::UB;
    p.bq = (const float*)(p.wqkv + (size_t)T::NUNITS * T::UB);
    p.bo2 = p.bq + 320;
    p.ldt = ldt; p.ldo = ldo; p.B = B; p.F = F; p.S = HW;
    p.G = 48 / F;
    p.gpb = (HW + p.G - 1) / p.G;
    p.ngroups = B * p.gpb;
    p.fmagic = (65536 + F - 1) / F;
    p.eps = eps;
    constexpr int lds = T::XB + T::NU * T::UB;
    auto kern = tattn2_kernel<320>;
    static const hipError_t attr_rc = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (attr_rc != hipSuccess) return vdx_fail("temporal_attn_block2: cannot reserve %d bytes of LDS", lds);
    const int tiles = (p.ngroups + 3) / 4;
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), lds, (hipStream_t)stream, p);
    return vdx_launch_status("vdx_temporal_attn_block2_f16");
}
