// ff_fused.hip — K8: the feed-forward sub-block of diffusers' BasicTransformerBlock as ONE kernel (SURVEY.md Appendix
// A.5 / A.6: `t = t + ff(norm3(t))`, ff = GEGLU(320 -> 2 x 1280, erf GELU) -> Linear(1280 -> 320); reached from
// fsdp_chunked_coherent.py:140 through Transformer2DModel and TransformerTemporalModel):
//
//     t' = t + W2 . ( val * gelu(gate) ) + b2,      [val | gate] = W1 . LayerNorm(t) + b1
//
// Un-fused this is LayerNorm, a GEGLU GEMM that writes the [rows][1280] intermediate (1.13 GB per call at level 0 of the
// XL step) and a GEMM that reads it back: 1.39 ms for 1.09 TFLOP.  Here the intermediate never leaves the CU.  The
// kernel is tattn2.hip's machinery (K7, second design) with the attention taken out:
//   * a workgroup of 4 waves holds 192 rows (48 per wave, private to it) as the LayerNorm-ed fp16 MFMA-operand image in
//     LDS (120 KB); LayerNorm's affine is folded into W1 / b1 on the host (packing.pack_k8);
//   * the hidden width is walked in 20 chunks of 64: five K-64 steps compute val^T and gate^T of the chunk as [hidden][row]
//     accumulators (96 registers; their bias is the initial accumulator, loaded straight into them while they are dead),
//     val * gelu(gate) is formed in registers and IS the B operand of the second product (host-permuted k index, as K7's
//     W_o), three steps add the chunk's contribution to the [column][row] output accumulators (240 registers: the
//     accumulator file), which live across the whole tile;
//   * weights stream through the same 5-unit ring of 8 KB units (15 units per chunk, 300 per tile, the stream runs on
//     across chunks and tiles), exact-or-conservative counted vmcnt waits, one barrier per step;
//   * persistent: a workgroup walks tiles blockIdx.x, + gridDim.x, ...
// Built for inner 320 (level 0 of the UNet: 10 feed-forwards per forward).
#include "vdx_common.h"
#include <utility>

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((address_space(3))) char lchar;
typedef __attribute__((address_space(3))) f16x8 lf16x8;
typedef __attribute__((address_space(1))) f16 gf16;
typedef __attribute__((address_space(1))) f16x8 gf16x8;
typedef __attribute__((address_space(1))) f32x4 gf32x4;
typedef __attribute__((address_space(1))) float gf32;

__device__ __forceinline__ void wg_barrier() {
#ifdef K8_ABL_NOBAR
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return;
#endif
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct K8P {
    const f16* t;        // [M][ldt]
    f16* out;            // [M][ldo]
    const char* w;       // [chunk 20][15 units][8192 B]: (val, gate) x 5 K-64 steps, then W2's k slice: 2 + 2 + 1 units
    const float* b1;     // [chunk 20][val 64 | gate 64]   b1 + W1 . beta
    const float* b2;     // [320]
    int ldt, ldo, M, ntiles;
    float eps;
    // PO (round 5): the transformer's proj_out + its residual fused behind the feed-forward — out = x + W_p . (t + ff(t)) + b_p
    const f16* x;        // [xrows][ldx] the transformer's input rows (row r of t pairs with row r % xrows of x)
    const char* wp;      // [25 units]: W_p in tattn2's output-projection unit format, natural k order
    const float* bp;     // [320]
    int ldx, xrows;
};

static __device__ __attribute__((aligned(16))) u32x4 g_dump_page8[64 + 64];

__device__ __forceinline__ float dpp_add8(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    return v;
}

// Phi(x) = 0.5 (1 + erf(x / sqrt 2)) as 0.5 + xc P(xc^2), xc = clamp(x, -4.5, 4.5), P of degree 8 (fitted; |error| <=
// 2e-5 on the interval, the clamped tail 4.5 (1 - Phi(4.5)) = 1.5e-5 in the product): 12 full-rate instructions, no
// transcendental — the chunk's 48 products per lane sit between two MFMA phases, so their issue time is exposed.
#define K8_C0 3.988662064e-01f
#define K8_C1 -6.624013931e-02f
#define K8_C2 9.729332291e-03f
#define K8_C3 -1.076692832e-03f
#define K8_C4 8.726890519e-05f
#define K8_C5 -4.958988029e-06f
#define K8_C6 1.845751427e-07f
#define K8_C7 -4.001098564e-09f
#define K8_C8 3.804222562e-11f
__device__ __forceinline__ float gelu_poly(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, -4.5f, 4.5f);
    const float u = xc * xc;
    float p = K8_C8;
    p = fmaf(p, u, K8_C7);
    p = fmaf(p, u, K8_C6);
    p = fmaf(p, u, K8_C5);
    p = fmaf(p, u, K8_C4);
    p = fmaf(p, u, K8_C3);
    p = fmaf(p, u, K8_C2);
    p = fmaf(p, u, K8_C1);
    p = fmaf(p, u, K8_C0);
    return x * fmaf(xc, p, 0.5f);
}

template <int INNER, bool PO = false>
struct K8 {
    static constexpr int KS = INNER / 32, KM = KS / 2;    // MFMA k steps / K-64 steps over the model width
    static constexpr int HID = 4 * INNER, CHUNKS = HID / 64;
    static constexpr int ROWS = 192, RB = INNER * 2, XB = ROWS * RB;
    static constexpr int UB = 8192, NU = 5;
    static constexpr int NCGF = INNER / 128, NCG = (INNER + 127) / 128;
    static constexpr int CSTEPS = KM + NCG;               // steps of one chunk: KM of val|gate, NCG of the output product
    static constexpr int UPC = 2 * KM + 2 * NCGF + (NCG - NCGF);    // units of one chunk
    static constexpr int NCB = INNER / 64, RBB = NCB * 1024, NPS = 6;
    static constexpr int NT = INNER / 16;                 // 16-column tiles of the output
    static_assert(INNER == 320, "geometry: written for inner 320");
    static_assert(XB + NU * UB <= 160 * 1024, "LDS budget");
    static_assert(UPC % NU == 0, "the ring position of a unit must not depend on the chunk");

    // ---- the static schedule of a chunk (the stream runs on into the next chunk: s >= CSTEPS)
    static constexpr int kind(int s) { return (s % CSTEPS) < KM ? 0 : 2; }          // 0 val|gate, 2 output product
    static constexpr int ubl(int r) { return r <= KM ? 2 * r : 2 * KM + (r - KM <= NCGF ? 2 * (r - KM) : 2 * NCGF + (r - KM - NCGF)); }
    static constexpr int ub(int s) { return UPC * (s / CSTEPS) + ubl(s % CSTEPS); }
    static constexpr int hm(int s) { return s < 0 ? NU : ub(s + 1) + NU; }
    static constexpr int nt_of(int g) { return g < NCGF ? 8 : 4; }
    // vector-memory instructions other than weight pieces, issued at the top of step s: the next chunk's bias (8 loads;
    // tools/k8_check_waits.py compares these counts and every wait with the emitted ISA)
    static constexpr int n_b1(int s) { return s % CSTEPS == KM ? 8 : 0; }
    static constexpr int inflight(int s) { return 2 * (hm(s - 1) - ub(s + 2)) + n_b1(s); }

    // ---- PO: the tail of a tile — 15 steps of tattn2's output-projection schedule (3 column groups x 5 K-64 steps, 25 units of
    // W_p), contracted over y = t + ff(t) held in registers as B operands; the next tile's rows are fetched and normalised
    // behind it (tattn2's row-prefetch schedule with RS0 = TRS0), so the tile seam of the plain kernel is gone.  The unit
    // stream runs on from the last chunk into the tail and from the tail into the next tile's chunk 0.
    static constexpr int TSTEPS = NCG * KM, TUNITS = 2 * KM * NCGF + KM * (NCG - NCGF), TRS0 = KM;
    static_assert(!PO || TUNITS % NU == 0, "the ring position of the next tile's first unit must not depend on the tail");
    static constexpr int tub(int t) {
        if (t >= TSTEPS) return TUNITS + ubl(t - TSTEPS);                      // the next tile's chunk 0
        const int c = t / KM, m = t % KM;
        return c < NCGF ? 2 * KM * c + 2 * m : 2 * KM * NCGF + (t - NCGF * KM);
    }
    static constexpr int thm(int t) { return t < 0 ? NU : tub(t + 1) + NU; }
    static constexpr int txp(int t) { return t == TRS0 || t == TRS0 + 1 ? 3 * (INNER / 64) : 0; }          // row pieces issued in tail step t
    static constexpr int tp0_mask(int t) {
        return t == TRS0 + 3 ? 0x03 : t == TRS0 + 5 ? 0x04 : t == TRS0 + 6 ? 0x08 : t == TRS0 + 7 ? 0x10 : t == TRS0 + 8 ? 0x20 : 0;
    }
    static constexpr int tfirst(int c) { return c * KM; }
    static constexpr int tn_bias(int t) {
        for (int c = 1; c < NCG; ++c) if (t == tfirst(c) - 1) return nt_of(c);        // (group 0's bias is loaded before the tail)
        return 0;
    }
    static constexpr int tn_res(int t) {
        for (int c = 0; c < NCG; ++c) if (t == tfirst(c) + 1) return 3 * nt_of(c) / 2;
        return 0;
    }
    static constexpr int tn_st(int t) {
        for (int c = 0; c < NCG; ++c) if (t == tfirst(c) + KM - 1) return 3 * nt_of(c) / 2;
        return 0;
    }
    // vector-memory instructions younger than the last unit piece tail step t needs (tattn2.hip `younger`); at t = 0 the loads
    // issued before the tail are left out: counting fewer only waits for more
    static constexpr int tyounger(int t) { return t == 0 ? 0 : txp(t - 1) + tn_st(t - 1) + tn_bias(t) + tn_res(t); }
    static constexpr int tinflight(int t) { return 2 * (thm(t - 1) - tub(t + 2)) + tyounger(t); }

    struct Frag {
        f16x8 w[8], x[3];
    };
    struct State {
        Frag fa, fb;
        f32x4 av[3][4], ag[3][4];        // val^T, gate^T of the chunk: [hidden][row]
        f32x4 b1v[8];                    // the next chunk's bias (val 0-3, gate 4-7): the initial accumulator of its first MFMAs
        f32x4 acc[3][NT];                // the output: [column][row]
        f16x8 hh[3][2];                  // val * gelu(gate) as B operands: [row tile][k step of the chunk]
        const gf16* resp[3];
        gf16* outp[3];
        // PO
        f16x8 yb[3][NT / 2];             // y = t + ff(t) as B operands of the tail: [row tile][k-32 step]
        f32x4 acc2[3][8];                // the tail's column group: [column][row]
        f16x8 rv2[3][4];                 // residual rows (x) of the current column group
        f32x4 bv[8];                     // its bias, the initial accumulator
        const gf16* xpr[3];              // this lane's rows of x (+ 8*q4), or the dump page
    };

    const K8P& p;
    char* smem;
    lchar* lds;
    int lane, n16, q4, wave;
    int woffb, xb[2];
    const char* wc0;                     // weights of the current chunk / the next chunk
    const char* wc1;
    const gf32* b1n;                     // bias of the next chunk

    __device__ __forceinline__ K8(const K8P& p_, char* s) : p(p_), smem(s), lds((lchar*)s) {}

    __device__ static __forceinline__ int opaque(int v) {
        asm volatile("" : "+v"(v));
        return v;
    }

    // ---- weight stream: unit U of the running chunk (U >= UPC: of the next chunk)
    template <int U>
    __device__ __forceinline__ void issue_unit() {
#ifdef K8_ABL_NOWDMA       /* diagnostic builds (timing only, wrong results): what each part of the tile costs */
        return;
#endif
        const char* src = (U >= UPC ? wc1 + (size_t)(U - UPC) * UB : wc0 + (size_t)U * UB) + (2 * wave) * 1024 + lane * 16;
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
    // tail unit U (U >= TUNITS: unit U - TUNITS of chunk 0 — of the next tile; the weights are the same for every tile)
    template <int U>
    __device__ __forceinline__ void issue_tunit() {
        const char* src = (U >= TUNITS ? p.w + (size_t)(U - TUNITS) * UB : p.wp + (size_t)U * UB) + (2 * wave) * 1024 + lane * 16;
        char* dst = smem + XB + (U % NU) * UB + (2 * wave) * 1024;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(src + 1024), (lptr_t)(dst + 1024), 16, 0, 0);
    }
    template <int U0, int U1>
    __device__ __forceinline__ void issue_trange() {
        if constexpr (U0 < U1) {
            issue_tunit<U0>();
            issue_trange<U0 + 1, U1>();
        }
    }

    // ---- the row image (tattn2.hip's layout: [row block of 8][column block of 64 channels][8 rows][128 B], chunk c of a
    // row at position c ^ (row & 7)); rows past M read the zero page
    template <int PS>
    __device__ __forceinline__ void issue_rows(int tile) {
        const long long gr = (long long)tile * ROWS + wave * 48 + 8 * PS + (lane >> 3);
        const bool ok = gr < p.M;
        const char* rowp = (const char*)(p.t + (ok ? gr : 0ll) * p.ldt) + (((lane & 7) ^ (lane >> 3)) << 4);
        const char* src = ok ? rowp : (const char*)g_zero_page;
        const int cstep = ok ? 128 : 0;
        char* dst = smem + (wave * 6 + PS) * RBB;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + cb * cstep), (lptr_t)(dst + cb * 1024), 16, 0, 0);
    }
    // centre and scale the 8 rows of pass PS in place (tattn2.hip p0_pass: fp32 statistics, fp16-exact differences)
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
#ifdef K8_ABL_NOLDSX
        f16x8 v;
        for (int e = 0; e < 8; ++e) v[e] = (f16)(float)(i + ks + e + lane);
        asm volatile("" : "+v"(v));
        return v;
#endif
        return *(const lf16x8*)(lds + xb[ks & 1] + (2 * i * RBB + 1024 * (ks >> 1)));
    }
    __device__ __forceinline__ f16x8 wfrag(int unit, int tile) const {
#ifdef K8_ABL_NOLDSW      /* timing only: weight fragments from a register pattern instead of LDS */
        f16x8 v;
        for (int e = 0; e < 8; ++e) v[e] = (f16)(float)(unit + tile + e + lane);
        asm volatile("" : "+v"(v));
        return v;
#endif
        return *(const lf16x8*)(lds + woffb + ((unit % NU) * UB + tile * 1024));
    }

    // fragments of half KK (one MFMA k step) of step S (S may be CSTEPS: step 0 of the next chunk)
    template <int S_, int KK>
    __device__ __forceinline__ void read_half(Frag& f) const {
        constexpr int S = S_ % CSTEPS;
        constexpr int u0 = ubl(S);
        if constexpr (kind(S) == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f.w[j] = wfrag(u0, 4 * KK + j);
                f.w[4 + j] = wfrag(u0 + 1, 4 * KK + j);
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) f.x[i] = xfrag(i, 2 * S + KK);
        } else {
            constexpr int g = S - KM;
            if constexpr (g < NCGF) {
#pragma unroll
                for (int j = 0; j < 8; ++j) f.w[j] = wfrag(u0 + KK, j);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) f.w[j] = wfrag(u0, 4 * KK + j);
            }
        }
    }

    template <int S, int KK, int NDS, int NVM>
    __device__ __forceinline__ void mma_half(State& st, const Frag& f) {
        if constexpr (kind(S) == 0) {
            constexpr bool Z = S == 0 && KK == 0;          // the chunk's first MFMAs start from its bias
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    st.av[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.w[j], f.x[i], Z ? st.b1v[j] : st.av[i][j], 0, 0, 0);
                    st.ag[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.w[4 + j], f.x[i], Z ? st.b1v[4 + j] : st.ag[i][j], 0, 0, 0);
                }
        } else {
            constexpr int g = S - KM;
#pragma unroll
            for (int j = 0; j < nt_of(g); ++j)
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    st.acc[i][8 * g + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.w[j], st.hh[i][KK], st.acc[i][8 * g + j], 0, 0, 0);
        }
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
        const int s = s_ % CSTEPS;
        return kind(s) == 0 ? 11 : nt_of(s - KM);
    }

    // val * gelu(gate) of the chunk -> the B operand of its output product.  Registers e of tile j: hidden 16*j + 4*q4 + e;
    // two tiles side by side are one operand (W2's k index is permuted to match: packing.pack_k8).
    __device__ __forceinline__ void geglu(State& st) {
#ifdef K8_ABL_NOGEGLU
        for (int i = 0; i < 3; ++i)
            for (int kk = 0; kk < 2; ++kk)
                for (int e = 0; e < 4; ++e) {
                    st.hh[i][kk][e] = (f16)(st.av[i][2 * kk][e] + st.ag[i][2 * kk][e]);
                    st.hh[i][kk][4 + e] = (f16)(st.av[i][2 * kk + 1][e] + st.ag[i][2 * kk + 1][e]);
                }
        return;
#endif
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    st.hh[i][kk][e] = (f16)(st.av[i][2 * kk][e] * gelu_poly(st.ag[i][2 * kk][e]));
                    st.hh[i][kk][4 + e] = (f16)(st.av[i][2 * kk + 1][e] * gelu_poly(st.ag[i][2 * kk + 1][e]));
                }
    }
    // bias of a chunk = the initial accumulators of its val / gate products (all three row tiles alike): loaded one chunk
    // ahead into 8 registers that the chunk's first MFMAs take as their C operand
    __device__ __forceinline__ void load_b1(State& st, const gf32* b) {
        const int o = opaque(4 * q4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            st.b1v[j] = *(const gf32x4*)(b + o + 16 * j);
            st.b1v[4 + j] = *(const gf32x4*)(b + 64 + o + 16 * j);
        }
    }

    // ================= PO: the tail (proj_out + residual) =================
    // y = t + ff(t): what the plain kernel's epilogue stores, kept as the tail's B operands — tile pair (2a, 2a+1) of a row
    // tile holds this lane's 8 consecutive columns 32a + 8*q4 .. +7 of row n16, which IS the B operand of k-32 step a
    // (k = 8*q4 + e, natural channel order).  Also the bias of the tail's first column group.
    __device__ __forceinline__ void make_y(State& st) {
        f16x8 rv[3][NT / 2];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int a = 0; a < NT / 2; ++a) rv[i][a] = *(const gf16x8*)(st.resp[i] + 32 * a);
        load_bias2<0>(st);
#pragma unroll
        for (int a = 0; a < NT / 2; ++a)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                f16x8 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = (f16)st.acc[i][2 * a][e];
                    o[4 + e] = (f16)st.acc[i][2 * a + 1][e];
                }
                st.yb[i][a] = o + rv[i][a];
            }
    }
    template <int C>
    __device__ __forceinline__ void load_bias2(State& st) {
        const int o = opaque(C * 128 + 8 * q4);
#pragma unroll
        for (int j = 0; j < nt_of(C); ++j) st.bv[j] = *(const gf32x4*)((const gf32*)p.bp + o + 32 * (j / 2) + 4 * (j % 2));
    }
    template <int C>
    __device__ __forceinline__ void load_residual2(State& st) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const gf16* src = st.xpr[i] + opaque(0);
#pragma unroll
            for (int a = 0; a < nt_of(C) / 2; ++a) st.rv2[i][a] = *(const gf16x8*)(src + C * 128 + 32 * a);
        }
    }
    // column group C: the projection (bias included) rounded to fp16, the residual added in fp16 (the reference's order)
    template <int C>
    __device__ __forceinline__ void epilogue2(State& st) {
#pragma unroll
        for (int a = 0; a < nt_of(C) / 2; ++a)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                f16x8 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = (f16)st.acc2[i][2 * a][e];
                    o[4 + e] = (f16)st.acc2[i][2 * a + 1][e];
                }
                o = o + st.rv2[i][a];
                *(gf16x8*)(st.outp[i] + C * 128 + 32 * a) = o;
            }
    }
    // weight fragments of half KK of tail step T (T >= TSTEPS: the next tile's chunk step T - TSTEPS)
    template <int T, int KK>
    __device__ __forceinline__ void read_thalf(Frag& f) const {
        if constexpr (T >= TSTEPS) {
            read_half<T - TSTEPS, KK>(f);              // (its unit indices are chunk-relative: the ring position is the same)
        } else {
            constexpr int c = T / KM, u0 = tub(T);
            if constexpr (c < NCGF) {
#pragma unroll
                for (int j = 0; j < 8; ++j) f.w[j] = wfrag(u0 + KK, j);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) f.w[j] = wfrag(u0, 4 * KK + j);
            }
        }
    }
    static constexpr int tnds(int t) { return t >= TSTEPS ? nds(t - TSTEPS) : nt_of(t / KM); }
    template <int T, int KK, int NDS, int NVM>
    __device__ __forceinline__ void tmma_half(State& st, const Frag& f) {
        constexpr int c = T / KM, m = T % KM;
        constexpr bool Z = m == 0 && KK == 0;
#pragma unroll
        for (int j = 0; j < nt_of(c); ++j)
#pragma unroll
            for (int i = 0; i < 3; ++i)
                st.acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.w[j], st.yb[i][2 * m + KK], Z ? st.bv[j] : st.acc2[i][j], 0, 0, 0);
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
    template <int T>
    __device__ __forceinline__ void tstep(State& st, int next_tile) {
        if constexpr (tn_bias(T) > 0) load_bias2<(T + 1) / KM>(st);
        if constexpr (tn_res(T) > 0) load_residual2<T / KM>(st);
        tp0_passes<tp0_mask(T)>(std::make_integer_sequence<int, NPS>{});
        read_thalf<T, 1>(st.fb);
        tmma_half<T, 0, tnds(T), 0>(st, st.fa);
        __builtin_amdgcn_sched_barrier(0);
        wait_vm<tinflight(T)>();
        __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0)
        asm volatile("" ::: "memory");
        wg_barrier();
        issue_trange<thm(T - 1), thm(T)>();
        if constexpr (txp(T) > 0) {
            issue_rows<3 * (T - TRS0)>(next_tile);
            issue_rows<3 * (T - TRS0) + 1>(next_tile);
            issue_rows<3 * (T - TRS0) + 2>(next_tile);
        }
        read_thalf<T + 1, 0>(st.fa);
        tmma_half<T, 1, tnds(T + 1), 2 * (thm(T) - thm(T - 1)) + txp(T)>(st, st.fb);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (tn_st(T) > 0) epilogue2<T / KM>(st);
    }
    template <int... T>
    __device__ __forceinline__ void tsteps(State& st, int next_tile, std::integer_sequence<int, T...>) {
        (tstep<T>(st, next_tile), ...);
    }
    template <int MASK, int... PS>
    __device__ __forceinline__ void tp0_passes(std::integer_sequence<int, PS...>) {
        ((MASK >> PS & 1 ? p0_pass<PS>() : void()), ...);
    }

    template <int S>
    __device__ __forceinline__ void step(State& st) {
        if constexpr (n_b1(S) > 0) load_b1(st, b1n);
        read_half<S, 1>(st.fb);
        mma_half<S, 0, nds(S), 0>(st, st.fa);
        __builtin_amdgcn_sched_barrier(0);
        wait_vm<inflight(S)>();
        __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0)
        asm volatile("" ::: "memory");
        wg_barrier();
        issue_range<hm(S - 1), hm(S)>();
        read_half<S + 1, 0>(st.fa);
        mma_half<S, 1, nds(S + 1), 2 * (hm(S) - hm(S - 1))>(st, st.fb);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (S == KM - 1) geglu(st);
    }
    template <int... S>
    __device__ __forceinline__ void steps(State& st, std::integer_sequence<int, S...>) {
        (step<S>(st), ...);
    }
    template <int... PS>
    __device__ __forceinline__ void rows_in(int tile, std::integer_sequence<int, PS...>) {
        (issue_rows<PS>(tile), ...);
    }
    template <int... PS>
    __device__ __forceinline__ void rows_norm(std::integer_sequence<int, PS...>) {
        (p0_pass<PS>(), ...);
    }

    __device__ __forceinline__ void set_lane_constants() {
        n16 = lane & 15;
        q4 = lane >> 4;
        const int g = (0x1320 >> (4 * (n16 >> 2))) & 3;
        woffb = XB + n16 * 64 + ((q4 ^ g) << 4);
        const int rr = n16 & 7, xrow = (wave * 6 + (n16 >> 3)) * RBB + rr * 128;
        xb[0] = xrow + ((q4 ^ rr) << 4);
        xb[1] = xrow + (((4 + q4) ^ rr) << 4);
    }

    // output bias = the initial value of the output accumulators (tile 2a + jj, register e: column 32a + 8*q4 + 4*jj + e)
    __device__ __forceinline__ void init_acc(State& st) {
        const int o = opaque(8 * q4);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const f32x4 b = *(const gf32x4*)((const gf32*)p.b2 + o + 32 * (j / 2) + 4 * (j % 2));
#pragma unroll
            for (int i = 0; i < 3; ++i) st.acc[i][j] = b;
        }
    }
    // rows of this tile as this lane reads the residual and writes the result (rows past M: the dump page)
    __device__ __forceinline__ void set_row_pointers(State& st, int tile) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const long long gr = (long long)tile * ROWS + wave * 48 + 16 * i + n16;
            const bool ok = gr < p.M;
            gf16* dump = (gf16*)g_dump_page8 + lane * 8;
            st.resp[i] = ok ? (const gf16*)p.t + gr * p.ldt + 8 * q4 : dump;
            st.outp[i] = ok ? (gf16*)p.out + gr * p.ldo + 8 * q4 : dump;
            if constexpr (PO) {
                const long long gx = gr >= p.xrows ? gr - p.xrows : gr;       // (xrows = M or M / 2: a shared-prefix batch pairs both items with the same rows of x)
                st.xpr[i] = ok ? (const gf16*)p.x + gx * p.ldx + 8 * q4 : dump;
            }
        }
    }
    // tile pair (2a, 2a+1) gives this lane 8 consecutive columns 32a + 8*q4 .. +7 of row n16 (+16i); the projection (bias
    // included) is rounded to fp16 and the residual added in fp16 — the reference's order
    __device__ __forceinline__ void epilogue(State& st) {
        f16x8 rv[3][NT / 2];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int a = 0; a < NT / 2; ++a) rv[i][a] = *(const gf16x8*)(st.resp[i] + 32 * a);
#pragma unroll
        for (int a = 0; a < NT / 2; ++a)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                f16x8 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = (f16)st.acc[i][2 * a][e];
                    o[4 + e] = (f16)st.acc[i][2 * a + 1][e];
                }
                o = o + rv[i][a];
                *(gf16x8*)(st.outp[i] + 32 * a) = o;
            }
    }

    __device__ __forceinline__ void run() {
        const int tid = threadIdx.x;
        lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        set_lane_constants();
        int tile = blockIdx.x;
        State st;
        wc0 = p.w;
        wc1 = p.w + (size_t)UPC * UB;
        issue_range<0, NU>();
        load_b1(st, (const gf32*)p.b1);
        rows_in(tile, std::make_integer_sequence<int, NPS>{});
        for (;;) {
            asm volatile("" : "+v"(lane));
            asm volatile("" : "+s"(wave));
            set_lane_constants();
            set_row_pointers(st, tile);
            init_acc(st);
            if (!PO || tile == (int)blockIdx.x) {          // (PO: a later tile's rows were fetched and normalised behind the previous
                wait_vm<0>();                              //  tile's tail, whose last step also read the first fragments)
                rows_norm(std::make_integer_sequence<int, NPS>{});
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                wg_barrier();                        // (first tile: the first units have landed for everyone)
                read_half<0, 0>(st.fa);
            }
            for (int c = 0; c < CHUNKS; ++c) {
                asm volatile("" : "+v"(lane));
                set_lane_constants();
                const int cn = c + 1 == CHUNKS ? 0 : c + 1;
                wc0 = p.w + (size_t)c * (UPC * UB);
                wc1 = PO && c + 1 == CHUNKS ? p.wp : p.w + (size_t)cn * (UPC * UB);      // (PO: the stream runs on into the tail)
                b1n = (const gf32*)p.b1 + cn * 128;
                steps(st, std::make_integer_sequence<int, CSTEPS>{});
            }
            const int next = tile + gridDim.x;
            if constexpr (PO) {
                // the last chunk's last step left the fragments of "chunk step 0" in st.fa: the tail's first units sit in those
                // ring slots, so the registers hold tail weights read as val / gate tiles — replaced here
                asm volatile("" : "+v"(lane));
                set_lane_constants();
                make_y(st);
                read_thalf<0, 0>(st.fa);
                tsteps(st, next, std::make_integer_sequence<int, TSTEPS>{});
                if (next >= p.ntiles) break;
                tile = next;
                continue;
            }
            rows_in(next, std::make_integer_sequence<int, NPS>{});     // (the image is dead; past the last tile: the zero page)
            epilogue(st);
            if (next >= p.ntiles) break;
            tile = next;
        }
        wait_vm<0>();
    }
};

template <int INNER, bool PO>
__global__ __launch_bounds__(256, 1) void ff_fused_kernel(const K8P p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    K8<INNER, PO> k(p, smem);
    k.run();
}

}  // namespace

extern "C" int vdx_ff_block_supported(int inner) { return inner == 320; }
// bytes of the packed blob (vdx/packing.py pack_k8): weight units, fp32 b1' per chunk, fp32 b2
extern "C" size_t vdx_ff_block_pack_bytes(int inner) {
    if (inner != 320) return 0;
    typedef K8<320> T;
    return (size_t)T::CHUNKS * T::UPC * T::UB + (size_t)(2 * T::HID + 320) * sizeof(float);
}

static int ff_block_launch(const void* t, int ldt, const void* packed, float eps, void* out, int ldo, int M, int inner,
                           const void* x, int ldx, int xrows, const void* proj_packed, vdx_stream_t stream) {
    VDX_CHECK(t && packed && out, "ff_block: null pointer");
    VDX_CHECK(M > 0, "ff_block: empty problem");
    VDX_CHECK(vdx_ff_block_supported(inner), "ff_block: inner=%d not supported (320)", inner);
    VDX_CHECK(ldt % 8 == 0 && ldo % 8 == 0 && ldt >= inner && ldo >= inner, "ff_block: bad leading dims");
    VDX_CHECK(((uintptr_t)t % 16 == 0) && ((uintptr_t)out % 16 == 0) && ((uintptr_t)packed % 16 == 0), "ff_block: pointers must be 16-byte aligned");
    VDX_CHECK(t != out, "ff_block: out may not alias t (a tile's residual is read after other tiles were written)");
    typedef K8<320> T;
    K8P p;
    p.t = (const f16*)t; p.out = (f16*)out;
    p.w = (const char*)packed;
    p.b1 = (const float*)(p.w + (size_t)T::CHUNKS * T::UPC * T::UB);
    p.b2 = p.b1 + 2 * T::HID;
    p.ldt = ldt; p.ldo = ldo; p.M = M;
    p.ntiles = (M + T::ROWS - 1) / T::ROWS;
    p.eps = eps;
    p.x = (const f16*)x; p.ldx = ldx; p.xrows = xrows;
    p.wp = (const char*)proj_packed;
    p.bp = (const float*)(p.wp + (size_t)T::TUNITS * T::UB);
    constexpr int lds = T::XB + T::NU * T::UB;
    auto kern = proj_packed ? ff_fused_kernel<320, true> : ff_fused_kernel<320, false>;
    static const hipError_t attr_rc = [] {
        const hipError_t a = hipFuncSetAttribute((const void*)ff_fused_kernel<320, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        const hipError_t b = hipFuncSetAttribute((const void*)ff_fused_kernel<320, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        return a != hipSuccess ? a : b;
    }();
    if (attr_rc != hipSuccess) return vdx_fail("ff_block: cannot reserve %d bytes of LDS", lds);
    const int ncu = vdx_grid_cus();
    const int rounds = (p.ntiles + ncu - 1) / ncu;
    const int grid = (p.ntiles + rounds - 1) / rounds;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, (hipStream_t)stream, p);
    return vdx_launch_status(proj_packed ? "vdx_ff_block_proj_f16" : "vdx_ff_block_f16");
}

extern "C" int vdx_ff_block_f16(const void* t, int ldt, const void* packed, float eps, void* out, int ldo, int M, int inner,
                                vdx_stream_t stream) {
    return ff_block_launch(t, ldt, packed, eps, out, ldo, M, inner, nullptr, 0, 0, nullptr, stream);
}

// bytes of the tail blob (vdx/packing.py pack_k8_proj): W_p units, fp32 b_p
extern "C" size_t vdx_ff_block_proj_pack_bytes(int inner) {
    if (inner != 320) return 0;
    return (size_t)K8<320>::TUNITS * K8<320>::UB + 320 * sizeof(float);
}
// K8 with the transformer's proj_out + residual behind it: out = x[r % xrows] + W_p . (t[r] + ff(LayerNorm(t[r]))) + b_p
extern "C" int vdx_ff_block_proj_f16(const void* t, int ldt, const void* packed, float eps, const void* x, int ldx, int xrows,
                                     const void* proj_packed, void* out, int ldo, int M, int inner, vdx_stream_t stream) {
    VDX_CHECK(x && proj_packed, "ff_block_proj: null pointer");
    VDX_CHECK(ldx % 8 == 0 && ldx >= inner && ((uintptr_t)x % 16 == 0) && ((uintptr_t)proj_packed % 16 == 0), "ff_block_proj: x / blob alignment or leading dim");
    VDX_CHECK(xrows == M || 2 * xrows == M, "ff_block_proj: xrows = %d must be M or M / 2 (M = %d)", xrows, M);
    VDX_CHECK(x != out, "ff_block_proj: out may not alias x");
    return ff_block_launch(t, ldt, packed, eps, out, ldo, M, inner, x, ldx, xrows, proj_packed, stream);
}

// Lab variants of this translation unit (phase stamps, ablations: timing only, some give WRONG results) are compiled in only
// under the macros below; a library that carries one says so through vdx_build_flags() and vdx/_lib.py refuses to load it
// as the product.
extern "C" int vdx_lab_ff_fused(void) {
#if defined(K8_ABL_NOWDMA) || defined(K8_ABL_NOLDSX) || defined(K8_ABL_NOLDSW) || defined(K8_ABL_NOGEGLU) || defined(K8_ABL_NOBAR)
    return 32;
#else
    return 0;
#endif
}
