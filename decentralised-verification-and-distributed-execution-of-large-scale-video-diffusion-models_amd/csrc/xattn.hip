// xattn.hip — K5: the cross-attention sub-block of diffusers' BasicTransformerBlock (Transformer2DModel, level 0 of the
// UNet; reached from fsdp_chunked_coherent.py:140) as ONE kernel:
//
//     t' = t + to_out( softmax( q K^T / 8 ) V ),   q = LayerNorm(t) . W_q^T,   K, V = the sample's text keys / values (77 x inner)
//
// Un-fused this is LayerNorm, a q GEMM, the flash kernel on 77 keys and an output GEMM + residual: four round trips of a
// [442 368][320] tensor per block.  Here a row is read once and written once.  The kernel is tattn2.hip's machinery (K7,
// second design) with the roles changed:
//   * rows: 192 consecutive rows per tile (4 waves x 48), never straddling two batch items; centred / scaled in LDS,
//     LayerNorm's affine and the softmax scale folded into W_q on the host (packing.pack_k5);
//   * per head: five K-64 steps project q (24 MFMAs each), then ONE attention step: the head's text keys and values — 80
//     key slots, 2 x 10 KB in MFMA-fragment order, packed once per prompt (packing.pack_k5_kv) — arrive as three units of
//     the same weight ring; the scores S^T = K q^T (key on the registers, query on the lane), the softmax over the 80 slots
//     (slots >= kv_len masked) and O^T = V^T P^T run in registers; O^T is the B operand of the output projection as it
//     stands (tattn2's permuted W_o);
//   * the output projection, bias, residual, epilogue, the prefetch of the next tile's rows behind the projection, the
//     5-unit weight ring and the counted s_waitcnt scheme are tattn2's (see the comments there).
// Built for inner 320 (level 0: 5 cross-attentions per forward).
#include "vdx_common.h"
#include <utility>

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((address_space(3))) char lchar;
typedef __attribute__((address_space(3))) f16x8 lf16x8;
typedef __attribute__((address_space(3))) f16x4 lf16x4;
typedef __attribute__((address_space(1))) f16 gf16;
typedef __attribute__((address_space(1))) f16x8 gf16x8;
typedef __attribute__((address_space(1))) f32x4 gf32x4;

__device__ __forceinline__ void wg_barrier() {
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct K5P {
    const f16* t;        // [M][ldt] rows
    f16* out;            // [M][ldo]
    const char* wq;      // [head][5 units][8192 B]: q0 .. q4 (a unit: 64 weight rows x 64 k as 8 tiles: tile 4*kk + j)
    const char* wo;      // [cg A: head x 2 units][cg B: head x 2 units][cg C: head x 1 unit]
    const char* kv;      // [item][head][3 units]: K fragments [kt 5][j 4][lane][8 B], V fragments [kt 5][dt 4][lane][8 B], pad
    const float* bq;     // [inner]  c . W_q . beta
    const float* bo2;    // [inner]  b_o
    int ldt, ldo;
    int B, S;            // batch items, rows per item
    int tps;             // tiles per batch item = ceil(S / 192): a tile never straddles two batch items
    int ntiles;          // B * tps
    int kv_len;          // text tokens (<= 80): key slots kv_len .. 79 are masked
    float eps;
};

static __device__ __attribute__((aligned(16))) u32x4 g_dump_page5[64 + 64];

__device__ __forceinline__ float dpp_add8(float v) {        // sum over the 8 lanes that share a row (lane & ~7)
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    return v;
}
// value of lanes l, l^16, l^32, l^48 combined (the four lane quads that hold one query's keys): see tattn2.hip
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

template <int INNER>
struct K5 {
    static constexpr int KS = INNER / 32;                 // MFMA k steps over the model width
    static constexpr int HEADS = INNER / 64;
    static constexpr int KM = KS / 2;                     // K-64 steps over the model width
    static constexpr int ROWS = 192;
    static constexpr int RB = INNER * 2;                  // bytes of one row of the image
    static constexpr int XB = ROWS * RB;
    static constexpr int UB = 8192, NU = 5;               // ring: NU units of UB bytes
    static constexpr int KVU = 3;                         // units of one head's keys + values
    static constexpr int KT = 5;                          // key tiles of 16 slots
    static constexpr int HSTEPS = KM + 1;                 // steps of one head: KM of q, one of attention
    static constexpr int P1S = HEADS * HSTEPS;
    static constexpr int NCGF = INNER / 128;              // full 128-column groups of the output projection
    static constexpr int NCG = (INNER + 127) / 128;
    static constexpr int NSTEP = P1S + NCG * HEADS;       // the output projection contracts head by head (K = 64)
    static constexpr int UPH = KM + KVU;                  // units of one head in the stream
    static constexpr int NPS = 6;                         // P0 passes of 8 rows
    static constexpr int PPP = 8 * RB / 1024;             // DMA pieces per pass
    static_assert(KS % 2 == 0 && (INNER % 128 == 0 || INNER % 128 == 64), "geometry");
    static_assert(XB + NU * UB <= 160 * 1024, "LDS budget");
    static constexpr int NCB = INNER / 64;                // column blocks of 64 channels (= PPP: one DMA piece each)
    static constexpr int RBB = NCB * 1024;                // bytes of one row block (8 rows)
    static_assert(PPP == NCB && XB == 24 * RBB, "a DMA piece is one (row block, column block): 8 rows x 128 bytes");
    static_assert(NCG == 3 && NPS == 6 && HEADS == 5, "the row prefetch schedule below is written for three column groups of five steps");
    static_assert(KVU <= NU && 2 * KT * 4 * 512 <= KVU * UB, "a head's keys and values must fit its units and the ring");

    // ---- the static schedule of a tile (tattn2.hip): step s consumes units [ub(s), ub(s+1)) of the weight stream
    static constexpr int kind(int s) { return s < P1S ? ((s % HSTEPS) < KM ? 0 : 1) : 2; }        // 0 q, 1 attention, 2 out
    static constexpr int ub1(int s) {
        if (s <= P1S) return UPH * (s / HSTEPS) + (s % HSTEPS);                 // q steps: one unit each; the attention step: KVU
        const int v = s - P1S, c = v / HEADS, m = v % HEADS;
        return UPH * HEADS + (c < NCGF ? 2 * HEADS * c + 2 * m : 2 * HEADS * NCGF + (v - NCGF * HEADS));
    }
    static constexpr int NUNITS = ub1(NSTEP);
    static constexpr int ub(int s) { return s <= NSTEP ? ub1(s) : NUNITS + ub1(s - NSTEP); }
    static constexpr int hm(int s) { return ub(s + 1) + NU; }
    static_assert(NUNITS % NU == 0, "the ring position of a unit must not depend on the tile");
    static_assert(ub1(HSTEPS) == UPH && ub1(KM + 1) - ub1(KM) == KVU, "the attention step owns the head's KVU units");

    // ---- every vector-memory instruction a wave issues, in order (tattn2.hip): the step's s_waitcnt vmcnt(N) is exact
    static constexpr int RS0 = P1S + HEADS;
    static constexpr int xp(int s) { return s == RS0 || s == RS0 + 1 ? 3 * PPP : 0; }               // row pieces issued in step s
    static constexpr int p0_mask_of(int s) {
        return s == RS0 + 3 ? 0x03 : s == RS0 + 5 ? 0x04 : s == RS0 + 6 ? 0x08 : s == RS0 + 7 ? 0x10 : s == RS0 + 8 ? 0x20 : 0;
    }
    static constexpr int nt_of(int c) { return c < NCGF ? 8 : 4; }
    static constexpr int first_of(int c) { return P1S + c * HEADS; }
    // q bias of the next head: loaded at the top of the head's attention step (its q accumulators are dead by then)
    static constexpr int n_bq(int s) { return kind(s) == 1 && s + 1 < P1S ? 4 : (s == NSTEP - 1 ? 4 : 0); }
    static constexpr int n_bias(int s) {
        for (int c = 0; c < NCG; ++c) if (s == first_of(c) - 1) return nt_of(c);
        return 0;
    }
    static constexpr int n_res(int s) {
        for (int c = 0; c < NCG; ++c) if (s == first_of(c) + 1) return 3 * nt_of(c) / 2;
        return 0;
    }
    static constexpr int n_st(int s) {
        for (int c = 0; c < NCG; ++c) if (s == first_of(c) + HEADS - 1) return 3 * nt_of(c) / 2;
        return 0;
    }
    static constexpr int prev(int s) { return s == 0 ? NSTEP - 1 : s - 1; }
    static constexpr int younger(int s) { return xp(prev(s)) + n_st(prev(s)) + n_bq(s) + n_bias(s) + n_res(s); }
    static constexpr int inflight(int s) { return 2 * (hm(s - 1) - ub(s + 2)) + younger(s); }

    struct Frag {
        f16x8 w[8], x[3];
    };
    struct State {
        Frag fa, fb;
        f32x4 aq[3][4];                          // q^T: [d][row]
        f32x4 acc[3][8];                         // output projection: [col][row]
        f16x8 oh[HEADS][3][2];                   // the heads' outputs as B operands: [time slot][row tile][k step of the head]
        f32x4 bqv[4];
        f16x4 qh[3][4];                          // q of the current head as fp16 MFMA operands
        f16x4 kf[KT][4], vh[KT][4];              // the head's K and V fragments (read once, before the step's barrier frees their units)
        f16x4 pt[3][KT];                         // P^T of the current head: [query tile][key tile]
        f16x8 rv[3][4];                          // residual rows of the current column group
        f32x4 bv[8];                             // its output bias, the projection's initial accumulator
        const gf16* resp[3];
        gf16* outp[3];
    };

    const K5P& p;
    char* smem;
    lchar* lds;
    int lane, n16, q4, wave;
    int rot, rotn;                               // head walking order of this / the next tile (tattn2.hip: a function of the tile's position in its item)
    int woffb, xb[2];
    int kvoff;                                   // LDS byte offset of this lane inside a fragment block: lane * 8
    int tb, tg, tbn, tgn;                        // this wave's (batch item, 48-row group inside it) in this tile / the next tile

    __device__ __forceinline__ K5(const K5P& p_, char* s) : p(p_), smem(s), lds((lchar*)s) {}

    __device__ static __forceinline__ int opaque(int v) {
        asm volatile("" : "+v"(v));
        return v;
    }

    // ---- weight stream
    template <int U>
    __device__ __forceinline__ const char* unit_src() const {
        constexpr int u = U % NUNITS;
        const bool nx = U >= NUNITS;                      // (the stream runs on into the next tile)
        const int r = nx ? rotn : rot;
        if constexpr (u < UPH * HEADS) {
            constexpr int hs = u / UPH, w = u % UPH;
            int h = hs + r;
            if (h >= HEADS) h -= HEADS;
            if constexpr (w < KM) return p.wq + (size_t)(h * KM + w) * UB;
            else {
                int item = nx ? tbn : tb;                 // (past the last tile: any valid item; the data is never used)
                if (item >= p.B) item = 0;
                return p.kv + ((size_t)(item * HEADS + h) * KVU + (w - KM)) * UB;
            }
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

    // global row index of local row r (0..47) of row group g4 of batch item b and whether it exists
    __device__ __forceinline__ bool grow_of(int b, int g4, int r, long long& gr) const {
        const int pix = g4 * 48 + r;
        const bool ok = b < p.B && pix < p.S;
        gr = ok ? (long long)b * p.S + pix : 0ll;
        return ok;
    }

    // ---- the row image (tattn2.hip's layout)
    template <int PS>
    __device__ __forceinline__ void issue_rows(int b, int g4) {
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
    template <int PS>
    __device__ __forceinline__ void p0_pass() {
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

    __device__ __forceinline__ f16x8 xfrag(int i, int ks) const {
        return *(const lf16x8*)(lds + xb[ks & 1] + (2 * i * RBB + 1024 * (ks >> 1)));
    }
    __device__ __forceinline__ f16x8 wfrag(int unit, int tile) const {
        return *(const lf16x8*)(lds + woffb + ((unit % NU) * UB + tile * 1024));
    }
    // fragment block BLK (512 bytes: 64 lanes x 8) of the head whose first K|V unit is U0: byte BLK * 512 of the head's stream
    template <int U0, int BLK>
    __device__ __forceinline__ f16x4 kvfrag() const {
        constexpr int byte = BLK * 512, u = U0 + byte / UB, o = byte % UB;
        return *(const lf16x4*)(lds + kvoff + (XB + (u % NU) * UB + o));
    }

    // fragments of half KK (one MFMA k step) of step S (S may be NSTEP: step 0 of the next tile)
    template <int S_, int KK>
    __device__ __forceinline__ void read_half(Frag& f) const {
        constexpr int S = S_ % NSTEP;
        constexpr int kd = kind(S), u0 = ub(S);
        if constexpr (kd == 0) {
            constexpr int m = S % HSTEPS;
#pragma unroll
            for (int j = 0; j < 4; ++j) f.w[j] = wfrag(u0, 4 * KK + j);
#pragma unroll
            for (int i = 0; i < 3; ++i) f.x[i] = xfrag(i, 2 * m + KK);
        } else if constexpr (kd == 2) {
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

    // the MFMAs of half KK of step S (q steps and output steps; the attention step has its own body)
    template <int S, int KK, int NDS, int NVM>
    __device__ __forceinline__ void mma_half(State& st, const Frag& f) {
        constexpr int kd = kind(S);
        if constexpr (kd == 0) {
            constexpr bool Z = (S % HSTEPS) == 0 && KK == 0;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    st.aq[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.w[j], f.x[i], Z ? st.bqv[j] : st.aq[i][j], 0, 0, 0);
        } else if constexpr (kd == 2) {
            constexpr int v = S - P1S, c = v / HEADS, hs = v % HEADS;
            constexpr bool Z = hs == 0 && KK == 0;
            constexpr int NT = nt_of(c);
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    st.acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.w[j], st.oh[hs][i][KK], Z ? st.bv[j] : st.acc[i][j], 0, 0, 0);
        }
        if constexpr (kd != 1) {
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
    }
    static constexpr int nds(int s_) {
        const int s = s_ % NSTEP;
        return kind(s) == 0 ? 7 : kind(s) == 1 ? 0 : nt_of((s - P1S) / HEADS);
    }

    // ---- attention of the head in time slot HS on the wave's 48 rows, all in registers.
    // First half of the attention step (before the step's barrier lets the ring overwrite the head's units): q as fp16
    // operands, the V fragments into registers, the scores against the K fragments in LDS, the softmax -> P^T as fp16.
    template <int S>
    __device__ __forceinline__ void attn_first(State& st) {
        constexpr int u0 = ub(S);
#ifdef K5_ABL_NOATT        /* lab build (timing only, wrong results): what the attention step's register work costs */
        for (int qt = 0; qt < 3; ++qt) for (int kk = 0; kk < 2; ++kk) for (int e = 0; e < 8; ++e) st.oh[S / HSTEPS][qt][kk][e] = (f16)st.aq[qt][2 * kk + (e >> 2)][e & 3];
        return;
#endif
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) st.qh[i][j][e] = (f16)st.aq[i][j][e];
        read_kv<u0>(st, std::make_integer_sequence<int, KT * 4>{});      // 40 ds_read_b64, all in flight together
        scores<0>(st);
        scores<1>(st);
        scores<2>(st);
    }
    template <int U0, int... B>
    __device__ __forceinline__ void read_kv(State& st, std::integer_sequence<int, B...>) {
        ((st.kf[B / 4][B % 4] = kvfrag<U0, B>()), ...);
        ((st.vh[B / 4][B % 4] = kvfrag<U0, KT * 4 + B>()), ...);
    }
    // scores of query tile QT: S^T = K q^T (query on the lane, the lane's keys of tile kt: 16*kt + 4*q4 + e)
    template <int QT>
    __device__ __forceinline__ void scores(State& st) {
        f32x4 sc[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            sc[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) sc[kt] = __builtin_amdgcn_mfma_f32_16x16x16f16(st.kf[kt][j], st.qh[QT][j], sc[kt], 0, 0, 0);
        }
        const int kl = p.kv_len - 4 * q4;                 // key slot 16*kt + 4*q4 + e exists iff 16*kt + e < kl
        float mx = -1.0e30f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            if (16 * kt + 16 > p.kv_len) {                // (wave-uniform: only the tiles that hold padded slots pay for the mask)
#pragma unroll
                for (int e = 0; e < 4; ++e) sc[kt][e] = 16 * kt + e < kl ? sc[kt][e] : -1.0e30f;      // exp2 below gives exactly 0
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) mx = fmaxf(mx, sc[kt][e]);
        }
        mx = quad_max(mx);
        float rs = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sc[kt][e] = __builtin_amdgcn_exp2f(sc[kt][e] - mx);         // (the scale is in W_q)
                rs += sc[kt][e];
            }
        rs = quad_sum(rs);
        const float inv = 1.0f / rs;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int e = 0; e < 4; ++e) st.pt[QT][kt][e] = (f16)(sc[kt][e] * inv);
    }
    // second half: O^T[d][query] = V^T P^T: lane = query row, registers e = d 16*dt + 4*q4 + e; two d tiles side by side
    // are one B operand of the output projection (k index of W_o permuted to match: packing.pack_k5)
    template <int HS>
    __device__ __forceinline__ void attn_pv(State& st) {
#ifdef K5_ABL_NOATT
        return;
#endif
#pragma unroll
        for (int qt = 0; qt < 3; ++qt)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                f32x4 o0 = (f32x4){0.f, 0.f, 0.f, 0.f}, o1 = o0;
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    o0 = __builtin_amdgcn_mfma_f32_16x16x16f16(st.vh[kt][2 * kk], st.pt[qt][kt], o0, 0, 0, 0);
                    o1 = __builtin_amdgcn_mfma_f32_16x16x16f16(st.vh[kt][2 * kk + 1], st.pt[qt][kt], o1, 0, 0, 0);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    st.oh[HS][qt][kk][e] = (f16)o0[e];
                    st.oh[HS][qt][kk][4 + e] = (f16)o1[e];
                }
            }
    }

    template <int HS, bool NEXT_TILE = false>
    __device__ __forceinline__ void load_bq(State& st) {
        int h = HS + (NEXT_TILE ? rotn : rot);
        if (h >= HEADS) h -= HEADS;
        const int o = opaque(h * 64 + 4 * q4);
#pragma unroll
        for (int j = 0; j < 4; ++j) st.bqv[j] = *(const gf32x4*)((const __attribute__((address_space(1))) float*)p.bq + o + 16 * j);
    }
    template <int C>
    __device__ __forceinline__ void load_bias(State& st) {
        const int o = opaque(C * 128 + 8 * q4);
#pragma unroll
        for (int j = 0; j < nt_of(C); ++j) st.bv[j] = *(const gf32x4*)((const __attribute__((address_space(1))) float*)p.bo2 + o + 32 * (j / 2) + 4 * (j % 2));
    }
    template <int C>
    __device__ __forceinline__ void load_residual(State& st) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const gf16* src = st.resp[i] + opaque(0);
#pragma unroll
            for (int a = 0; a < nt_of(C) / 2; ++a) st.rv[i][a] = *(const gf16x8*)(src + C * 128 + 32 * a);
        }
    }
    template <int C>
    __device__ __forceinline__ void epilogue(State& st) {
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
        if constexpr (n_bq(S) > 0) load_bq<(S == NSTEP - 1 ? 0 : S / HSTEPS + 1), S == NSTEP - 1>(st);
        if constexpr (n_bias(S) > 0) load_bias<(S + 1 - P1S) / HEADS>(st);
        if constexpr (n_res(S) > 0) load_residual<(S - P1S) / HEADS>(st);
        p0_passes<p0_mask_of(S)>(std::make_integer_sequence<int, NPS>{});
        if constexpr (kd == 1) {
            attn_first<S>(st);
        } else {
            read_half<S, 1>(st.fb);
            mma_half<S, 0, nds(S), 0>(st, st.fa);
        }
        // the units of step S+1 have landed for everyone, and nobody reads the units of step S any more
        __builtin_amdgcn_sched_barrier(0);
        wait_vm<inflight(S)>();
        __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): my reads of step S's units are done
        asm volatile("" ::: "memory");
        wg_barrier();
        issue_range<hm(S - 1), hm(S)>();
        if constexpr (xp(S) > 0) {
            issue_rows<3 * (S - RS0)>(tbn, tgn);
            issue_rows<3 * (S - RS0) + 1>(tbn, tgn);
            issue_rows<3 * (S - RS0) + 2>(tbn, tgn);
        }
        read_half<S + 1, 0>(st.fa);
        if constexpr (kd == 1) {
            attn_pv<S / HSTEPS>(st);
        } else {
            mma_half<S, 1, nds(S + 1), 2 * (hm(S) - hm(S - 1)) + xp(S)>(st, st.fb);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (n_st(S) > 0) epilogue<(S - P1S) / HEADS>(st);
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
        kvoff = lane * 8;
    }

    // tile -> (batch item, this wave's 48-row group inside it, head rotation); tiles are aligned to batch items
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
        int tile = blockIdx.x;
        set_tile(tile, tb, tg, rot);
        rotn = rot;
        tbn = tb;
        tgn = tg;
        State st;
        issue_range<0, NU>();
        load_bq<0>(st);
        first_rows(std::make_integer_sequence<int, NPS>{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wg_barrier();
        read_half<0, 0>(st.fa);

        for (;;) {
            asm volatile("" : "+v"(lane));
            asm volatile("" : "+s"(wave));
            set_lane_constants();
            const int next = tile + gridDim.x;
            set_tile(next, tbn, tgn, rotn);      // (past the last tile: every row reads the zero page)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                long long gr;
                const bool ok = grow_of(tb, tg, 16 * i + n16, gr);
                gf16* dump = (gf16*)g_dump_page5 + lane * 8;
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
        wait_vm<0>();        // (the stream ran on into a tile that does not exist: let its copies land before the LDS is released)
    }
};

template <int INNER>
__global__ __launch_bounds__(256, 1) void xattn_kernel(const K5P p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    K5<INNER> k(p, smem);
    k.run();
}

}  // namespace

extern "C" int vdx_cross_attn_block_supported(int inner, int kv_len) {
    return inner == 320 && kv_len >= 1 && kv_len <= 16 * K5<320>::KT;
}
// bytes of the static blob (vdx/packing.py pack_k5): q units, output-projection units, fp32 q bias, fp32 output bias
extern "C" size_t vdx_cross_attn_block_pack_bytes(int inner) {
    if (inner != 320) return 0;
    typedef K5<320> T;
    return (size_t)(T::NUNITS - T::HEADS * T::KVU) * T::UB + 2 * 320 * sizeof(float);
}
// bytes of the per-prompt key / value blob of ONE batch item (vdx/packing.py pack_k5_kv)
extern "C" size_t vdx_cross_attn_block_kv_bytes(int inner) {
    if (inner != 320) return 0;
    return (size_t)K5<320>::HEADS * K5<320>::KVU * K5<320>::UB;
}

extern "C" int vdx_cross_attn_block_f16(const void* t, int ldt, const void* packed, const void* kv_packed, int kv_len, float eps,
                                        void* out, int ldo, int n_items, int rows_per_item, int inner, vdx_stream_t stream) {
    VDX_CHECK(t && packed && kv_packed && out, "cross_attn_block: null pointer");
    VDX_CHECK(n_items > 0 && rows_per_item > 0, "cross_attn_block: empty problem");
    VDX_CHECK(vdx_cross_attn_block_supported(inner, kv_len), "cross_attn_block: inner=%d kv_len=%d not supported (inner 320, 1..80 keys)", inner, kv_len);
    VDX_CHECK(ldt % 8 == 0 && ldo % 8 == 0 && ldt >= inner && ldo >= inner, "cross_attn_block: bad leading dims");
    VDX_CHECK((long long)n_items * rows_per_item < (1ll << 31), "cross_attn_block: too many rows");
    VDX_CHECK(((uintptr_t)t % 16 == 0) && ((uintptr_t)out % 16 == 0) && ((uintptr_t)packed % 16 == 0) && ((uintptr_t)kv_packed % 16 == 0),
              "cross_attn_block: pointers must be 16-byte aligned");
    VDX_CHECK(t != out, "cross_attn_block: out may not alias t (a tile's residual is read after other tiles were written)");
    typedef K5<320> T;
    K5P p;
    p.t = (const f16*)t; p.out = (f16*)out;
    p.wq = (const char*)packed;
    p.wo = p.wq + (size_t)T::KM * T::HEADS * T::UB;
    p.bq = (const float*)(p.wq + (size_t)(T::NUNITS - T::HEADS * T::KVU) * T::UB);
    p.bo2 = p.bq + 320;
    p.kv = (const char*)kv_packed;
    p.ldt = ldt; p.ldo = ldo; p.B = n_items; p.S = rows_per_item;
    p.tps = (rows_per_item + T::ROWS - 1) / T::ROWS;
    p.ntiles = n_items * p.tps;
    p.kv_len = kv_len;
    p.eps = eps;
    constexpr int lds = T::XB + T::NU * T::UB;
    static const hipError_t attr_rc = hipFuncSetAttribute((const void*)xattn_kernel<320>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (attr_rc != hipSuccess) return vdx_fail("cross_attn_block: cannot reserve %d bytes of LDS", lds);
    const int ncu = vdx_grid_cus();
    const int rounds = (p.ntiles + ncu - 1) / ncu;
    const int grid = (p.ntiles + rounds - 1) / rounds;
    hipLaunchKernelGGL(xattn_kernel<320>, dim3(grid), dim3(256), lds, (hipStream_t)stream, p);
    return vdx_launch_status("vdx_cross_attn_block_f16");
}

// Lab variants (timing only, wrong results) are compiled in only under the macro below; vdx_build_flags() reports them.
extern "C" int vdx_lab_xattn(void) {
#if defined(K5_ABL_NOATT)
    return 128;
#else
    return 0;
#endif
}
