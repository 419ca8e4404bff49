// flash.hip — flash-style attention on v_mfma_f32_32x32x16_f16: spatial
// self-attention (seq up to 9216, never materialised) and text cross-attention (77 keys).
//
// A wave owns QB sub-blocks of 32 queries (QB = 2: 64 queries per wave); K / V tiles of 64 keys go through LDS
// (LDS-DMA staged) and every fragment read from LDS feeds QB MFMAs.  Per sub-block:
//   * scores are computed TRANSPOSED, S^T = K.Q^T, so a lane owns ONE query column: the row max is
//     lane-local plus one exchange with lane^32, and the exponentiated tile is already the B
//     operand of O^T = V^T.P^T (accumulator-as-operand, no LDS round trip for P).  K rows are fed
//     in the bit-swapped order that makes that operand's permuted k-index the natural key order,
//     so V^T fragments are plain 16-byte LDS reads when V arrives pre-transposed ([d][key], VROW = false: the
//     text K/V of cross-attention and the CLIP tower, projected once by a swapped GEMM).  VROW = true: V arrives
//     as ROWS ([key][d], the third column block of one q|k|v projection), is staged like K and read through
//     ds_read_b64_tr_b16 — the transposition happens in the LDS read, two 8-byte reads per operand.
//   * Q is pre-scaled by scale*log2(e); the running maximum is folded INTO the score contraction by
//     one extra k-step ([1,0,..] row of "K" times [-m,0,..] column of "Q"), so S' = S - m costs one
//     MFMA per 32 keys and no per-score VALU op, and in the common tile p = exp2(S') directly.
//     m only has to be the same for a row's p and its row sum, so its fp16 rounding is harmless.
//   * row sums by v_dot2_f32_f16 against ones, row maxima by v_max3_f32 — both as FOUR independent chains per
//     sub-block: a dependent chain of either costs 8.3 cycles per instruction, independent ones 5.3
//     (tools/micro/coissue.hip, profiles/r04_flash.md).
// Blocks are 256 threads, two per CU: a SIMD holds one wave of each.  profiles/r04_flash.md has the accounting of a tile
// (timing-only builds, phase stamps, instruction costs) and the measured reason why an explicit pairing of the two waves of
// a SIMD in alternating matrix / vector segments (a 512-thread "ping-pong" kernel, built in round 4, bit-identical, commit
// "flash: ping-pong kernel experiment") is not faster: beside a wave that streams MFMAs, the softmax's vector mix is all
// but starved, whichever wave is older or has the priority.
#include "attn_common.h"

// Diagnostic switches (tools/flash_abl.sh builds them into csrc/build/abl/libflash_<tag>.so; the product defines none):
//   FL_ABL_NOEXP / NOSUM / NOMAX / NODMA / NOKREAD / NOVREAD / NOBAR   timing-only builds with one part of the tile
//       removed (WRONG RESULTS by construction) — profiles/r04_flash.md's ablation table;
//   FL_STAMPS   per-wave s_memtime sums of the tile's phases into g_fl_stamps (read back with vdx_flash_stamps_read).
//   FL_ABL_MFMA16   timing-only: every 32x32x16 MFMA issued as two 16x16x32 on quarters of its accumulator (the same flops; WRONG
//       RESULTS) — what the tile costs on the shape the chip clocks higher on (profiles/r04_flash.md §11).
#ifdef FL_ABL_MFMA16
template <int SEL>
__device__ __forceinline__ f32x16 fl_mfma(const f16x8 a, const f16x8 b, f32x16 c) {
    f32x4 q0 = __builtin_shufflevector(c, c, 8 * SEL, 8 * SEL + 1, 8 * SEL + 2, 8 * SEL + 3);
    f32x4 q1 = __builtin_shufflevector(c, c, 8 * SEL + 4, 8 * SEL + 5, 8 * SEL + 6, 8 * SEL + 7);
    q0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, q0, 0, 0, 0);
    q1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, q1, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        c[8 * SEL + e] = q0[e];
        c[8 * SEL + 4 + e] = q1[e];
    }
    return c;
}
#else
template <int SEL>
__device__ __forceinline__ f32x16 fl_mfma(const f16x8 a, const f16x8 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
#endif
#ifdef FL_STAMPS
static __device__ unsigned long long g_fl_stamps[8 * 32768];     // [block * waves + wave][8]: four phase sums, total, tiles
#define FL_T(i)                                                     \
    do {                                                            \
        __builtin_amdgcn_sched_barrier(0);                          \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        fl_sum[i] += now_ - fl_last;                                \
        fl_last = now_;                                             \
        __builtin_amdgcn_sched_barrier(0);                          \
    } while (0)
#else
#define FL_T(i)
#endif

struct FlashP {
    const f16 *q, *k, *vt;   // vt: V^T [heads*64][ldvt]  (VROW: V rows [n_kv*skv_pad][ldvt])
    f16* out;
    int ldq, ldk, ldvt, ldo;
    int sq, skv, skv_pad, seq_per_kv, heads;
    int xcd;           // XCD-aware block order in use (npairs % 8 == 0)
    int nqb, npairs;   // query blocks per (sequence, head) pair; pairs = n_seq * heads  (1-D grid of nqb * npairs blocks)
    int causal;   // 1: key j attends only to queries >= j (CLIP text tower); K/V tiles above the block's last query are skipped
    float c;  // scale * log2(e)
};

typedef const __attribute__((address_space(1))) void* fl_gptr_t;
typedef __attribute__((address_space(3))) void* fl_lptr_t;

// exchange with lane ^ 32 without LDS: v_permlane32_swap of (x, x) leaves the low half's value in a, the high half's in b
__device__ __forceinline__ float max_xor32(float x) {
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}

// 1-D grid, XCD-aware: blocks L and L+8 share an XCD (and its L2).  All query blocks of one (sequence, head)
// pair read the same K / V (2.4 MB at 9216 keys), so a pair's blocks are given to ONE XCD (pair = xcd mod 8)
// when the pair count divides by 8: its K/V then come from HBM once instead of once per XCD.
__device__ __forceinline__ void fl_block_map(const FlashP& p, int& pair, int& qblk) {
    if (p.xcd) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        pair = (slot / p.nqb) * 8 + xcd;
        qblk = slot % p.nqb;
    } else {
        pair = blockIdx.x / p.nqb;
        qblk = blockIdx.x % p.nqb;
    }
}

// Q fragments (B operand of S^T = K.Q^T): lane = query column, pre-scaled
template <int QB>
__device__ __forceinline__ void fl_load_q(const FlashP& p, int seq, int head, int q0, int r32, int h, f16x8 (&qf)[QB][4]) {
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qr = min(q0 + qb * 32 + r32, p.sq - 1);
        const f16* src = p.q + ((size_t)seq * p.sq + qr) * p.ldq + head * 64 + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const f16x8 raw = *(const f16x8*)(src + 16 * ks);
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[qb][ks][j] = (f16)((float)raw[j] * p.c);
        }
    }
}

// The softmax state of a wave's QB sub-blocks and one tile's vector work.
// The offset m is LAZY and MONOTONE.  r_run is the row's running maximum over all tiles so far, measured against the
// offset in use.  The offset stays put (and, while it is 0, its MFMA is skipped) as long as r_run lies in (-4, 10]:
// p = exp2(S') <= 2^10 is exact-enough fp16 with fp32 sums, and a running maximum >= 2^-4 keeps the terms that matter out
// of fp16 subnormals.  When r_run leaves the window the offset is re-centred ON THE RUNNING maximum (rounded to fp16; only
// consistency between p and l matters), never on the current tile's: after the first unmasked tile r_run >= -4 always
// holds, so from then on the offset only RISES (alpha <= 2^-10) and a row moves at most (score range / 10) times.  The one
// downward move possible is the first one (first tile all below -4): nothing has been accumulated yet, so it rescales
// nothing (alpha = 1) — exp2(-d) would overflow there for scores below -128 (0 * inf = NaN).  Tiles far below the running
// maximum simply underflow to p = 0, as they do in an fp32 softmax.
template <int QB>
struct FlSoft {
    float r_run[QB];              // running row maximum relative to the offset in use
    float l_run[QB][2];           // my half of the row sum, as two partial sums (element pair e of a tile goes to sum e & 1):
                                  // with the QB sub-blocks interleaved that is four independent dot2c chains
    f16x8 negm[QB];               // [-m, 0, ...]: the offset's column of "Q" (m: the offset in use, exp2 units, an fp16 number);
                                  // EVERY lane holds -m in element 0 — the k-step's "K" row e0 is zero for lane half 1, so
                                  // what half 1 of this operand holds does not matter, and m is read back from here
    bool offset_on;               // wave-uniform: some row of this wave has a non-zero offset
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            r_run[qb] = NEG_BIG;
            l_run[qb][0] = l_run[qb][1] = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) negm[qb][j] = (f16)0.0f;
        }
        offset_on = false;
    }
    static __device__ __forceinline__ f16x8 e0(int h) {        // [1, 0, ...]: the offset's row of "K"
        f16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (f16)((j == 0 && h == 0) ? 1.0f : 0.0f);
        return v;
    }
    // s_acc (scores minus the offset in use, masked by the caller) -> pf (fp16 probabilities, the B operand of P.V)
    __device__ __forceinline__ void tile(f32x16 (&s_acc)[QB][2], f32x16 (&o_acc)[QB][2], f16x8 (&pf)[QB][4], int h) {
        bool move[QB], any_move = false;
#ifndef FL_ABL_NOMAX
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            float mc[4] = {NEG_BIG, NEG_BIG, NEG_BIG, NEG_BIG};      // four independent v_max3 chains
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int j = 0; j < 16; j += 2) mc[(j >> 1) & 3] = fmaxf(fmaxf(mc[(j >> 1) & 3], s_acc[qb][kb][j]), s_acc[qb][kb][j + 1]);
            const float m = fmaxf(fmaxf(mc[0], mc[1]), fmaxf(mc[2], mc[3]));
            r_run[qb] = fmaxf(r_run[qb], max_xor32(m));
            // outside (-4, 10] and not "every tile so far masked" (nothing to centre on): bitwise, no short-circuit branches
            move[qb] = (fabsf(r_run[qb] - 3.0f) > 7.0f) & (r_run[qb] > -1.0e29f);
            any_move |= move[qb];
        }
#else
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) move[qb] = false;
#endif
        if (__builtin_amdgcn_ballot_w64(any_move) != 0) {            // rare
            bool nonzero = false;
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                // the fp16 offset enters the contraction as an MFMA operand: keep it finite (an infinite one
                // times the zero rows of its k-step would poison every score with NaN)
                const float m_old = -(float)negm[qb][0];
                const float tgt = fminf(fmaxf(m_old + (move[qb] ? r_run[qb] : 0.f), -60000.0f), 60000.0f);
                const float m_new = (float)(f16)tgt;
                const float d = m_new - m_old;                       // shift actually applied
                negm[qb][0] = (f16)(-m_new);
                r_run[qb] -= d;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int j = 0; j < 16; ++j) s_acc[qb][kb][j] -= d;
                const float alpha = d > 0.f ? __builtin_amdgcn_exp2f(-d) : 1.0f;   // d < 0: first move, O = l = 0
                l_run[qb][0] *= alpha;
                l_run[qb][1] *= alpha;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    o_acc[qb][0][j] *= alpha;
                    o_acc[qb][1][j] *= alpha;
                }
                nonzero |= m_new != 0.f;
            }
            offset_on = __builtin_amdgcn_ballot_w64(nonzero) != 0;
        }
        // p = exp2(S') packed to fp16 pairs, row sums by dot2
        f16x2 one2;
        one2[0] = one2[1] = (f16)1.0f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int j = 0; j < 16; j += 2)
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    f16x2 pp;
#ifndef FL_ABL_NOEXP
                    pp[0] = (f16)__builtin_amdgcn_exp2f(s_acc[qb][kb][j]);
                    pp[1] = (f16)__builtin_amdgcn_exp2f(s_acc[qb][kb][j + 1]);
#else
                    pp[0] = (f16)s_acc[qb][kb][j];
                    pp[1] = (f16)s_acc[qb][kb][j + 1];
#endif
#ifndef FL_ABL_NOSUM
                    l_run[qb][(j >> 1) & 1] = __builtin_amdgcn_fdot2(pp, one2, l_run[qb][(j >> 1) & 1], false);
#endif
                    pf[qb][kb * 2 + (j >> 3)][j & 7] = pp[0];
                    pf[qb][kb * 2 + (j >> 3)][(j & 7) + 1] = pp[1];
                }
    }
    __device__ __forceinline__ float row_sum(int qb) const {
        const float l = l_run[qb][0] + l_run[qb][1];
        return l + __shfl_xor(l, 32, 64);
    }
};

// epilogue: O[query][d], lane = query; pair lanes (l, l^32) to emit 16-byte stores
template <int QB>
__device__ __forceinline__ void fl_store(const FlashP& p, const FlSoft<QB>& st, const f32x16 (&o_acc)[QB][2], int seq, int head, int q0, int r32, int h) {
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const float inv = 1.0f / st.row_sum(qb);
        const int qrow = q0 + qb * 32 + r32;
        f16* dst = p.out + ((size_t)seq * p.sq + qrow) * p.ldo + head * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                // register group g holds d = 32*db + 8*g + 4*h + (0..3)
                f16x4 mine[2];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) mine[u][j] = (f16)(o_acc[qb][db][4 * (g + u) + j] * inv);
                const f16x4 send = h ? mine[0] : mine[1];
                typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                u32x2 sb = __builtin_bit_cast(u32x2, send), rb;
                rb[0] = __shfl_xor(sb[0], 32, 64);
                rb[1] = __shfl_xor(sb[1], 32, 64);
                const f16x4 recv = __builtin_bit_cast(f16x4, rb);
                f16x8 o;
                const f16x4 lo = h ? recv : mine[0], hi = h ? mine[1] : recv;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o[j] = lo[j];
                    o[4 + j] = hi[j];
                }
                if (qrow < p.sq) *(f16x8*)(dst + 32 * db + 8 * (g + h)) = o;
            }
        }
    }
}

// transposed-read address of a lane inside a V tile (VROW), for d block 0; d block 1 = ^ 64.  Lane 4q+p of a 16-lane group
// supplies row q, d columns 4p..4p+3 of a block of 4 keys x 16 d and receives column (lane & 15) of the 4 rows:
// group (lane >> 4) & 1 takes d 16..31 of the d block, lane half h keys 8h.. of the operand's 16.
__device__ __forceinline__ int fl_tr_addr(int lane) {
    const int h = lane >> 5, tq = (lane >> 2) & 3, tp = lane & 3, tg = (lane >> 4) & 1;
    return (8 * h + tq) * 128 + (((2 * tg + (tp >> 1)) ^ ((tq >> 1) << 2)) << 4) + 8 * (tp & 1);
}
__device__ __forceinline__ f16x8 fl_read_vtr(const char* vb) {      // keys 16kk + 8h + (0..3 | 4..7) of d = 32db + (lane & 31)
    typedef short s16x4v __attribute__((__vector_size__(8)));
    typedef __attribute__((address_space(3))) s16x4v* ltr_t;
    struct TrPair { s16x4v lo, hi; } pr;
    pr.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ltr_t)vb);
    pr.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ltr_t)(vb + 512));
    return __builtin_bit_cast(f16x8, pr);
}

// CAUSAL is a template parameter: the spatial / cross-attention instantiations carry no trace of the mask
template <int QB, bool CAUSAL, bool VROW>
__global__ __launch_bounds__(256) void flash_attn_kernel(const FlashP p) {
    // LDS: 2 stages x { K tile [64 key][64 d], V^T tile [64 d][64 key] }, 128-B rows,
    // 16-B chunk c of row r stored at chunk c ^ ((r >> 1) & 7)  (conflict-free for both reads).
    // VROW: the second half of a stage is the V tile [64 key][64 d], chunk c of row r at c ^ (4 * ((r >> 1) & 1)):
    // a transposed read takes 4 key rows x 32 d (64 B) per 32-lane half; rows r, r+1 are 128 B apart (other half of
    // the 64 banks) and the XOR moves rows r+2, r+3 to the other 64 B of their lines: 32 lanes x 8 B on 64 banks once.
    __shared__ __attribute__((aligned(16))) char smem[2 * 16384];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r32 = lane & 31, h = lane >> 5;
    int pair, qblk;
    fl_block_map(p, pair, qblk);
    const int head = pair % p.heads, seq = pair / p.heads;
    const int kvb = seq / p.seq_per_kv;
    const int q0 = (qblk * 4 + wave) * (32 * QB);

    f16x8 qf[QB][4];
    fl_load_q<QB>(p, seq, head, q0, r32, h, qf);

    // ---- staging by LDS-DMA (global_load_lds_dwordx4): 2 K pieces + 2 V^T pieces per wave per tile.
    // Wave-instruction i fills LDS rows (i*256 + wave*64)/8 .. +7 of the tile; lane l lands at row
    // +(l>>3), 16-byte slot l&7, and therefore fetches data chunk (l&7) ^ ((row>>1)&7).  Source
    // pointers walk forward one tile per iteration; bounds are only checked in tiles that cross
    // skv_pad (rows / chunks past it come from the zero page).
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const f16* zp = (const f16*)g_zero_page;
    const int st_row0 = tid >> 3, st_row1 = st_row0 + 32;             // K key / V^T d row of my piece 0 / 1
    const int ch0 = (tid & 7) ^ ((st_row0 >> 1) & 7), ch1 = (tid & 7) ^ ((st_row1 >> 1) & 7);
    const int chv = (tid & 7) ^ (((st_row0 >> 1) & 1) << 2);           // VROW (st_row1 = st_row0 + 32: same swizzle)
    const f16* kptr0 = p.k + ((size_t)kvb * p.skv_pad + st_row0) * p.ldk + head * 64 + ch0 * 8;
    const f16* kptr1 = p.k + ((size_t)kvb * p.skv_pad + st_row1) * p.ldk + head * 64 + ch1 * 8;
    const f16* vptr0 = VROW ? p.vt + ((size_t)kvb * p.skv_pad + st_row0) * p.ldvt + head * 64 + chv * 8
                            : p.vt + ((size_t)head * 64 + st_row0) * p.ldvt + (size_t)kvb * p.skv_pad + ch0 * 8;
    const f16* vptr1 = VROW ? p.vt + ((size_t)kvb * p.skv_pad + st_row1) * p.ldvt + head * 64 + chv * 8
                            : p.vt + ((size_t)head * 64 + st_row1) * p.ldvt + (size_t)kvb * p.skv_pad + ch1 * 8;
    const size_t kstep = (size_t)64 * p.ldk, vstep = VROW ? (size_t)64 * p.ldvt : 64;
    auto issue = [&](int t, int buf) {
        char* sk = smem + buf * 16384 + wv * 1024;
        const int k0 = t * 64;
        const f16 *k0p = kptr0, *k1p = kptr1, *v0p = vptr0, *v1p = vptr1;
        if (k0 + 64 > p.skv_pad) {                                    // tile crosses skv_pad (wave-uniform)
            if (k0 + st_row0 >= p.skv_pad) k0p = zp;
            if (k0 + st_row1 >= p.skv_pad) k1p = zp;
            if (VROW) {
                if (k0 + st_row0 >= p.skv_pad) v0p = zp;
                if (k0 + st_row1 >= p.skv_pad) v1p = zp;
            } else {
                if (k0 + ch0 * 8 >= p.skv_pad) v0p = zp;              // chunks never straddle skv_pad
                if (k0 + ch1 * 8 >= p.skv_pad) v1p = zp;
            }
        }
        __builtin_amdgcn_global_load_lds((fl_gptr_t)k0p, (fl_lptr_t)sk, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((fl_gptr_t)k1p, (fl_lptr_t)(sk + 4096), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((fl_gptr_t)v0p, (fl_lptr_t)(sk + 8192), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((fl_gptr_t)v1p, (fl_lptr_t)(sk + 8192 + 4096), 16, 0, 0);
        kptr0 += kstep;
        kptr1 += kstep;
        vptr0 += vstep;
        vptr1 += vstep;
    };

    f32x16 o_acc[QB][2];
    FlSoft<QB> st;
    st.init();
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int j = 0; j < 16; ++j) o_acc[qb][0][j] = o_acc[qb][1][j] = 0.f;

    const f16x8 e0 = FlSoft<QB>::e0(h);
    const int krow = pi_row(r32);
    const int tr0 = fl_tr_addr(lane), tr1 = tr0 ^ 64;
    int ntiles = (p.skv + 63) >> 6;
    if (CAUSAL) ntiles = min(ntiles, ((qblk + 1) * 4 * 32 * QB + 63) >> 6);   // keys beyond the block's last query: all masked
    issue(0, 0);
    __syncthreads();                       // LDS-DMA in flight: the barrier's fence waits vmcnt(0)
#ifdef FL_STAMPS
    unsigned long long fl_sum[4] = {0, 0, 0, 0}, fl_last = __builtin_amdgcn_s_memtime();
    const unsigned long long fl_first = fl_last;
#endif
    for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
#ifndef FL_ABL_NODMA
        if (t + 1 < ntiles) issue(t + 1, cur ^ 1);   // buffer cur^1 was last read before the previous barrier
#endif
        FL_T(0);
        const char* Ks = smem + cur * 16384;
        const char* Vs = Ks + 8192;
        const int k0 = t * 64;

        // ---- S' = K . Q^T - m : two 32-key blocks, each K fragment feeds QB sub-blocks ------------
        f32x16 s_acc[QB][2], zero16;
#pragma unroll
        for (int j = 0; j < 16; ++j) zero16[j] = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int row = kb * 32 + krow;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int c = 2 * ks + h;
#if defined(FL_ABL_NOKREAD)
                const f16x8 kf = qf[0][(ks + kb) & 3];
#else
                const f16x8 kf = *(const f16x8*)(Ks + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
#endif
#pragma unroll
                for (int qb = 0; qb < QB; ++qb)
                    s_acc[qb][kb] = (ks & 1) ? fl_mfma<1>(kf, qf[qb][ks], s_acc[qb][kb]) : fl_mfma<0>(kf, qf[qb][ks], ks == 0 ? zero16 : s_acc[qb][kb]);
            }
            if (st.offset_on) {
#pragma unroll
                for (int qb = 0; qb < QB; ++qb)
                    s_acc[qb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0, st.negm[qb], s_acc[qb][kb], 0, 0, 0);   // - m
            }
        }
        if (k0 + 64 > p.skv) {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int j = 0; j < 16; ++j)
                        if (k0 + kb * 32 + acc_key(j, h) >= p.skv) s_acc[qb][kb][j] = NEG_BIG;
        }
        if (CAUSAL && k0 + 63 > q0) {                              // tile reaches past this wave's first query
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int j = 0; j < 16; ++j)
                        if (k0 + kb * 32 + acc_key(j, h) > q0 + qb * 32 + r32) s_acc[qb][kb][j] = NEG_BIG;
        }
        FL_T(1);
        // ---- row maxima, p = exp2(S') packed to fp16 pairs, row sums; O^T += V^T . P^T ------------------
        f16x8 pf[QB][4];
        st.tile(s_acc, o_acc, pf, h);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {  // kk = 2*kb + s : keys 16*kk + 8*h .. +7
            const int c = 2 * kk + h;
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                f16x8 vf;
#ifdef FL_ABL_NOVREAD
                vf = qf[0][(kk + db) & 3];
#else
                if (VROW) {
                    vf = fl_read_vtr(Vs + (db ? tr1 : tr0) + kk * 2048);
                } else {
                    const int row = db * 32 + r32;
                    vf = *(const f16x8*)(Vs + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
                }
#endif
#pragma unroll
                for (int qb = 0; qb < QB; ++qb)
                    o_acc[qb][db] = (kk & 1) ? fl_mfma<1>(vf, pf[qb][kk], o_acc[qb][db]) : fl_mfma<0>(vf, pf[qb][kk], o_acc[qb][db]);
            }
        }
        FL_T(2);
#ifndef FL_ABL_NOBAR
        __syncthreads();                   // next tile landed (vmcnt(0)) and this one is fully read
#endif
        FL_T(3);
    }
#ifdef FL_STAMPS
    {
        const unsigned long long tot = __builtin_amdgcn_s_memtime() - fl_first;
        if (lane == 0 && blockIdx.x < 8192) {
            unsigned long long* dst = g_fl_stamps + ((size_t)blockIdx.x * 4 + wave) * 8;
            dst[0] = fl_sum[0]; dst[1] = fl_sum[1]; dst[2] = fl_sum[2]; dst[3] = fl_sum[3];
            dst[4] = tot; dst[5] = (unsigned long long)ntiles;
        }
    }
#endif
    fl_store<QB>(p, st, o_acc, seq, head, q0, r32, h);
}

template <bool VROW>
static int flash_launch(const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt,
                        void* out, int ldo, int n_seq, int sq, int skv, int skv_pad, int heads,
                        int seq_per_kv, float scale, int causal, vdx_stream_t stream) {
    VDX_CHECK(q && k && vt && out, "flash_attn: null pointer");
    VDX_CHECK(n_seq > 0 && sq > 0 && skv > 0 && heads > 0 && seq_per_kv > 0, "flash_attn: empty problem");
    VDX_CHECK(skv_pad >= skv && (VROW || skv_pad % 8 == 0), "flash_attn: skv_pad=%d must be >= skv=%d%s", skv_pad, skv,
              VROW ? "" : " and a multiple of 8");
    VDX_CHECK(ldq % 8 == 0 && ldk % 8 == 0 && ldvt % 8 == 0 && ldo % 8 == 0, "flash_attn: leading dims must be multiples of 8");
    VDX_CHECK(n_seq % seq_per_kv == 0, "flash_attn: n_seq=%d not a multiple of seq_per_kv=%d", n_seq, seq_per_kv);
    VDX_CHECK(heads <= 65535 && n_seq <= 65535, "flash_attn: grid too large");
    FlashP p;
    p.q = (const f16*)q; p.k = (const f16*)k; p.vt = (const f16*)vt; p.out = (f16*)out;
    p.ldq = ldq; p.ldk = ldk; p.ldvt = ldvt; p.ldo = ldo;
    p.sq = sq; p.skv = skv; p.skv_pad = skv_pad; p.seq_per_kv = seq_per_kv; p.causal = causal ? 1 : 0;
    VDX_CHECK(!causal || seq_per_kv == 1, "flash_attn: causal masking is for self-attention (seq_per_kv == 1)");
    p.c = scale * 1.44269504088896341f;
    // 64 queries per wave when the sequence is long enough to fill the chip with 256-query blocks
    const bool two = sq >= 512 && skv >= 256;
    p.heads = heads;
    p.npairs = n_seq * heads;
    p.xcd = p.npairs % 8 == 0 ? 1 : 0;
    p.nqb = two ? (sq + 255) / 256 : (sq + 127) / 128;
    VDX_CHECK((long long)p.nqb * p.npairs < (1ll << 31), "flash_attn: grid too large");
    if (two) {
        dim3 grid(p.nqb * p.npairs);
        if (causal) hipLaunchKernelGGL((flash_attn_kernel<2, true, VROW>), grid, dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((flash_attn_kernel<2, false, VROW>), grid, dim3(256), 0, (hipStream_t)stream, p);
    } else {
        dim3 grid(p.nqb * p.npairs);
        if (causal) hipLaunchKernelGGL((flash_attn_kernel<1, true, VROW>), grid, dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((flash_attn_kernel<1, false, VROW>), grid, dim3(256), 0, (hipStream_t)stream, p);
    }
    return vdx_launch_status(VROW ? "vdx_flash_attn_rows_f16" : "vdx_flash_attn_f16");
}

extern "C" int vdx_flash_attn_f16(const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt,
                                  void* out, int ldo, int n_seq, int sq, int skv, int skv_pad, int heads,
                                  int seq_per_kv, float scale, int causal, vdx_stream_t stream) {
    return flash_launch<false>(q, ldq, k, ldk, vt, ldvt, out, ldo, n_seq, sq, skv, skv_pad, heads, seq_per_kv, scale, causal, stream);
}

extern "C" int vdx_flash_attn_rows_f16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv,
                                       void* out, int ldo, int n_seq, int sq, int skv, int skv_pad, int heads,
                                       int seq_per_kv, float scale, int causal, vdx_stream_t stream) {
    return flash_launch<true>(q, ldq, k, ldk, v, ldv, out, ldo, n_seq, sq, skv, skv_pad, heads, seq_per_kv, scale, causal, stream);
}

#ifdef FL_STAMPS
// diagnostic builds only: the per-wave phase sums of the last launch ([block * waves + wave][8] x u64)
extern "C" int vdx_flash_stamps_read(void* host, size_t bytes) {
    VDX_CHECK(host && bytes <= sizeof(g_fl_stamps), "flash_stamps_read: bad buffer");
    const hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(g_fl_stamps), bytes, 0, hipMemcpyDeviceToHost);
    return e == hipSuccess ? 0 : vdx_fail("flash_stamps_read: %s", hipGetErrorString(e));
}
#endif

// Lab variants of this translation unit (phase stamps, ablations: timing only, some give WRONG results) are compiled in only
// under the macros below; a library that carries one says so through vdx_build_flags() and vdx/_lib.py refuses to load it
// as the product.
extern "C" int vdx_lab_flash(void) {
#if defined(FL_STAMPS) || defined(FL_ABL_NOEXP) || defined(FL_ABL_MFMA16) || defined(FL_ABL_NOVREAD) || defined(FL_ABL_NOSUM) || defined(FL_ABL_NOMAX) || defined(FL_ABL_NOKREAD) || defined(FL_ABL_NODMA) || defined(FL_ABL_NOBAR)
    return 16;
#else
    return 0;
#endif
}
