// attn.hip — attention cores on v_mfma_f32_32x32x16_f16 (SURVEY.md §2.3 K4/K5/K7 core).
//
// flash_attn_kernel: spatial self-attention (seq up to 9216, never materialised) and text
//   cross-attention (77 keys).  Per wave 32 queries, per block 4 waves = 128 queries; K/V tiles of
//   64 keys staged through LDS (double-buffered, one barrier per tile).
//   Scores are computed TRANSPOSED, S^T = K.Q^T, so a lane owns ONE query column: row max / row
//   sum are lane-local plus one exchange with lane^32, and the exponentiated tile is already the
//   B operand of O^T = V^T.P^T (accumulator-as-operand, no LDS round trip for P).  The K rows are
//   fed in the order that makes the permuted k-index of that operand the natural key order, so
//   V^T fragments are plain 16-byte LDS reads.  V arrives pre-transposed ([d][key]) — the V
//   projection GEMM is simply issued with swapped operands.
//
// temporal_attn_kernel: TransformerTemporalModel's attention over the F <= 32 frames of one
//   latent pixel: one wave per (pixel, head), operands straight from global memory, a single
//   32x32 score tile, complete softmax in registers, O = P.V.  HBM-bound by construction.
#include "vdx_common.h"



#define NEG_BIG (-1.0e30f)

// key (0..31) held by accumulator register `reg` of lane-half `h` when K rows are fed through pi()
__device__ __forceinline__ int acc_key(int reg, int h) {
    return 16 * (reg >> 3) + 8 * h + 4 * ((reg >> 2) & 1) + (reg & 3);
}
// A-operand row i must carry key pi(i) = i with bits 2 and 3 swapped
__device__ __forceinline__ int pi_row(int i) {
    return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1);
}

struct FlashP {
    const f16 *q, *k, *vt;
    f16* out;
    int ldq, ldk, ldvt, ldo;
    int sq, skv, skv_pad, seq_per_kv;
    float c;  // scale * log2(e)
};

__global__ __launch_bounds__(256) void flash_attn_kernel(const FlashP p) {
    // LDS: 2 stages x { K tile [64 key][64 d], V^T tile [64 d][64 key] }, 128-B rows,
    // 16-B chunk c of row r stored at chunk c ^ ((r >> 1) & 7)  (conflict-free for both reads)
    __shared__ __attribute__((aligned(16))) char smem[2 * 16384];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r32 = lane & 31, h = lane >> 5;
    const int head = blockIdx.y, seq = blockIdx.z;
    const int kvb = seq / p.seq_per_kv;
    const int q0 = blockIdx.x * 128 + wave * 32;

    // ---- Q fragments (B operand of S^T = K.Q^T): lane = query column.  Pre-scaled by
    // scale*log2(e) so the scores come out of the MFMA already in exp2 units.
    f16x8 qf[4];
    {
        const int qr = min(q0 + r32, p.sq - 1);
        const f16* src = p.q + ((size_t)seq * p.sq + qr) * p.ldq + head * 64 + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const f16x8 raw = *(const f16x8*)(src + 16 * ks);
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[ks][j] = (f16)((float)raw[j] * p.c);
        }
    }

    // ---- staging descriptors: 2 K chunks + 2 V^T chunks per thread per tile ---------------
    const int cch = tid & 7;
    const f16* zp = (const f16*)g_zero_page;
    int st_row[2], st_lds[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        st_row[i] = (i * 256 + tid) >> 3;
        st_lds[i] = st_row[i] * 128 + ((cch ^ ((st_row[i] >> 1) & 7)) << 4);
    }
    // per-thread source pointers walk forward one tile per iteration (no per-tile 64-bit address
    // arithmetic); bounds are only checked in tiles that cross skv_pad (wave-uniform branch)
    const f16* kptr0 = p.k + ((size_t)kvb * p.skv_pad + st_row[0]) * p.ldk + head * 64 + cch * 8;
    const f16* kptr1 = kptr0 + (size_t)32 * p.ldk;                       // st_row[1] = st_row[0] + 32
    const f16* vptr0 = p.vt + ((size_t)head * 64 + st_row[0]) * p.ldvt + (size_t)kvb * p.skv_pad + cch * 8;
    const f16* vptr1 = vptr0 + (size_t)32 * p.ldvt;
    const size_t kstep = (size_t)64 * p.ldk;
    u32x4 rk[2], rv[2];
    auto gload = [&](int t) {
        const int k0 = t * 64;
        if (k0 + 64 <= p.skv_pad) {
            rk[0] = *(const u32x4*)kptr0;
            rk[1] = *(const u32x4*)kptr1;
            rv[0] = *(const u32x4*)vptr0;
            rv[1] = *(const u32x4*)vptr1;
        } else {                                                       // tile crosses skv_pad: zero fill
            const u32x4 z = {0u, 0u, 0u, 0u};
            const bool vok = k0 + cch * 8 < p.skv_pad;                   // chunks never straddle skv_pad
            rk[0] = k0 + st_row[0] < p.skv_pad ? *(const u32x4*)kptr0 : z;
            rk[1] = k0 + st_row[1] < p.skv_pad ? *(const u32x4*)kptr1 : z;
            rv[0] = vok ? *(const u32x4*)vptr0 : z;
            rv[1] = vok ? *(const u32x4*)vptr1 : z;
        }
        kptr0 += kstep;
        kptr1 += kstep;
        vptr0 += 64;
        vptr1 += 64;
    };
    auto lstore = [&](int buf) {
        char* s = smem + buf * 16384;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *(u32x4*)(s + st_lds[i]) = rk[i];
            *(u32x4*)(s + 8192 + st_lds[i]) = rv[i];
        }
    };

    // Online softmax with the running maximum folded INTO the score contraction: one extra k-step
    // multiplies a constant [1,0,..] row of "K" with a [-m,0,..] column of "Q", so S' = S - m costs
    // one MFMA per 32 keys on the matrix pipe (which has slack here) and no per-score VALU op; in the
    // common tile (no new maximum) p = exp2(S') directly.  m only has to be the SAME for a row's p and
    // its row sum, so its fp16 rounding is harmless.  The VALU keeps one v_exp, half a v_max3, half a
    // v_cvt_pk and half a v_dot2 (row sum) per score.
    f32x16 o_acc[2];
#pragma unroll
    for (int j = 0; j < 16; ++j) o_acc[0][j] = o_acc[1][j] = 0.f;
    float m_run = 0.f;                     // offset in use (exp2 units, fp16-representable)
    float l_run = 0.f;                     // this lane's half of the row sum
    f16x2 one2;
    one2[0] = one2[1] = (f16)1.0f;
    f16x8 e0, negm;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        e0[j] = (f16)((j == 0 && h == 0) ? 1.0f : 0.0f);
        negm[j] = (f16)0.0f;
    }

    const int krow = pi_row(r32);
    const int ntiles = (p.skv + 63) >> 6;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        if (t + 1 < ntiles) gload(t + 1);
        const char* Ks = smem + cur * 16384;
        const char* Vs = Ks + 8192;

        // ---- S' = K . Q^T - m : two 32-key blocks.  All 8 K fragments are requested up front so
        // the MFMA chain never waits for an LDS round trip per k-step.
        f16x8 kf[2][4];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int row = kb * 32 + krow;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int c = 2 * ks + h;
                kf[kb][ks] = *(const f16x8*)(Ks + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
            }
        }
        f32x16 s_acc[2], zero16;
#pragma unroll
        for (int j = 0; j < 16; ++j) zero16[j] = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                s_acc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kb][ks], qf[ks], ks == 0 ? zero16 : s_acc[kb], 0, 0, 0);
            s_acc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0, negm, s_acc[kb], 0, 0, 0);   // - m
        }
        // V^T fragments of the first 32 keys: requested now, they land during the softmax VALU work
        f16x8 vfa[2][2], vfb[2][2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const int row = db * 32 + r32, c = 2 * kk + h;
                vfa[kk][db] = *(const f16x8*)(Vs + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
            }
        const int k0 = t * 64;
        if (k0 + 64 > p.skv) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (k0 + kb * 32 + acc_key(j, h) >= p.skv) s_acc[kb][j] = NEG_BIG;
        }
        float mx = NEG_BIG;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int j = 0; j < 16; ++j) mx = fmaxf(mx, s_acc[kb][j]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        // tile 0 establishes the offset (may be negative); later it moves only when the row maximum
        // grows by more than 1/16 (p then stays <= 2^(1/16): no overflow, no perpetual re-trigger)
        const bool move = t == 0 || mx > 0.0625f;
        if (__builtin_amdgcn_ballot_w64(move) != 0) {              // rare after the first tiles
            const float m_new = (float)(f16)(m_run + (move ? mx : 0.f));
            const float d = m_new - m_run;                         // shift actually applied
            m_run = m_new;
            negm[0] = (f16)(h == 0 ? -m_new : 0.f);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int j = 0; j < 16; ++j) s_acc[kb][j] -= d;
            if (t > 0) {
                const float alpha = __builtin_amdgcn_exp2f(-d);
                l_run *= alpha;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    o_acc[0][j] *= alpha;
                    o_acc[1][j] *= alpha;
                }
            }
        }
        // p = exp2(S'), packed to fp16 pairs; row sum by v_dot2_f32_f16 against ones (half the adds)
        f16x8 pf[4];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int j = 0; j < 16; j += 2) {
                f16x2 pp;
                pp[0] = (f16)__builtin_amdgcn_exp2f(s_acc[kb][j]);
                pp[1] = (f16)__builtin_amdgcn_exp2f(s_acc[kb][j + 1]);
                l_run = __builtin_amdgcn_fdot2(pp, one2, l_run, false);
                pf[kb * 2 + (j >> 3)][j & 7] = pp[0];
                pf[kb * 2 + (j >> 3)][(j & 7) + 1] = pp[1];
            }
        // ---- O^T += V^T . P^T : second half of V^T requested before the first half is consumed ----
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const int row = db * 32 + r32, c = 2 * (kk + 2) + h;
                vfb[kk][db] = *(const f16x8*)(Vs + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int db = 0; db < 2; ++db)
                o_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfa[kk][db], pf[kk], o_acc[db], 0, 0, 0);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int db = 0; db < 2; ++db)
                o_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfb[kk][db], pf[kk + 2], o_acc[db], 0, 0, 0);
        if (t + 1 < ntiles) lstore(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: O[query][d], lane = query; pair lanes (l, l^32) to emit 16-byte stores ----
    const float inv = 1.0f / (l_run + __shfl_xor(l_run, 32, 64));
    const int qrow = q0 + r32;
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
                for (int j = 0; j < 4; ++j) mine[u][j] = (f16)(o_acc[db][4 * (g + u) + j] * inv);
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

extern "C" int vdx_flash_attn_f16(const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt,
                                  void* out, int ldo, int n_seq, int sq, int skv, int skv_pad, int heads,
                                  int seq_per_kv, float scale, vdx_stream_t stream) {
    VDX_CHECK(q && k && vt && out, "flash_attn: null pointer");
    VDX_CHECK(n_seq > 0 && sq > 0 && skv > 0 && heads > 0 && seq_per_kv > 0, "flash_attn: empty problem");
    VDX_CHECK(skv_pad >= skv && skv_pad % 8 == 0, "flash_attn: skv_pad=%d must be >= skv=%d and a multiple of 8", skv_pad, skv);
    VDX_CHECK(ldq % 8 == 0 && ldk % 8 == 0 && ldvt % 8 == 0 && ldo % 8 == 0, "flash_attn: leading dims must be multiples of 8");
    VDX_CHECK(n_seq % seq_per_kv == 0, "flash_attn: n_seq=%d not a multiple of seq_per_kv=%d", n_seq, seq_per_kv);
    VDX_CHECK(heads <= 65535 && n_seq <= 65535, "flash_attn: grid too large");
    FlashP p;
    p.q = (const f16*)q; p.k = (const f16*)k; p.vt = (const f16*)vt; p.out = (f16*)out;
    p.ldq = ldq; p.ldk = ldk; p.ldvt = ldvt; p.ldo = ldo;
    p.sq = sq; p.skv = skv; p.skv_pad = skv_pad; p.seq_per_kv = seq_per_kv;
    p.c = scale * 1.44269504088896341f;
    dim3 grid((sq + 127) / 128, heads, n_seq);
    hipLaunchKernelGGL(flash_attn_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    return vdx_launch_status("vdx_flash_attn_f16");
}

// =============================================================================================
struct TempP {
    const f16* qkv;
    f16* out;
    int ldqkv, ldo, B, F, HW, heads;
    float c;
    long long items;
};

__global__ __launch_bounds__(256) void temporal_attn_kernel(const TempP p) {
    const int lane = threadIdx.x & 63, r32 = lane & 31, h = lane >> 5;
    const long long item = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= p.items) return;
    const int head = (int)(item % p.heads);
    const long long bp = item / p.heads;
    const int pix = (int)(bp % p.HW), b = (int)(bp / p.HW);
    const int inner = p.heads * 64;
    const size_t row0 = (size_t)b * p.F * p.HW + pix;  // frame f lives at row0 + f*HW
    const f16* zp = (const f16*)g_zero_page;

    // S^T = K.Q^T : A = K rows (fed through pi so the k-order of the next product is natural),
    // B = Q^T (lane = query frame)
    f32x16 s_acc;
#pragma unroll
    for (int j = 0; j < 16; ++j) s_acc[j] = 0.f;
    {
        const int kr = pi_row(r32);
        const f16* ksrc = p.qkv + (row0 + (size_t)kr * p.HW) * p.ldqkv + inner + head * 64 + 8 * h;
        const f16* qsrc = p.qkv + (row0 + (size_t)r32 * p.HW) * p.ldqkv + head * 64 + 8 * h;
        f16x8 kf[4], qf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            kf[ks] = *(const f16x8*)(kr < p.F ? ksrc + 16 * ks : zp);
            qf[ks] = *(const f16x8*)(r32 < p.F ? qsrc + 16 * ks : zp);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            s_acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks], qf[ks], s_acc, 0, 0, 0);
    }
    // V as B operand of O = P.V : lane = column d, element j of k-step s = V[16s + 8h + j][d]
    f16x8 vf[2][2];
    {
        const f16* vsrc = p.qkv + row0 * p.ldqkv + 2 * inner + head * 64 + r32;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int key = 16 * s + 8 * h + j;
                const f16* src = vsrc + (size_t)key * p.HW * p.ldqkv;
                const bool ok = key < p.F;
                vf[0][s][j] = *(ok ? src : zp);
                vf[1][s][j] = *(ok ? src + 32 : zp);
            }
    }
    // complete softmax (all keys are in this one tile); normalise P before the second product
    float mx = NEG_BIG;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (acc_key(j, h) >= p.F) s_acc[j] = NEG_BIG;
        mx = fmaxf(mx, s_acc[j]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mc = mx * p.c;
    float e[16], rs = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        e[j] = __builtin_amdgcn_exp2f(s_acc[j] * p.c - mc);
        rs += e[j];
    }
    const float inv = 1.0f / (rs + __shfl_xor(rs, 32, 64));
    f16x8 pf[2];
#pragma unroll
    for (int j = 0; j < 16; ++j) pf[j >> 3][j & 7] = (f16)(e[j] * inv);
    // O = P.V : A = P (accumulator as operand: X^T.B form), rows = query frames
#pragma unroll
    for (int db = 0; db < 2; ++db) {
        f32x16 o;
#pragma unroll
        for (int j = 0; j < 16; ++j) o[j] = 0.f;
#pragma unroll
        for (int s = 0; s < 2; ++s) o = __builtin_amdgcn_mfma_f32_32x32x16_f16(pf[s], vf[db][s], o, 0, 0, 0);
        // C layout: col = lane&31 = d, row(reg) = (reg&3) + 8*(reg>>2) + 4*h = query frame
        f16* dst = p.out + row0 * p.ldo + head * 64 + 32 * db + r32;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int f = (j & 3) + 8 * (j >> 2) + 4 * h;
            if (f < p.F) dst[(size_t)f * p.HW * p.ldo] = (f16)o[j];
        }
    }
}

extern "C" int vdx_temporal_attn_f16(const void* qkv, int ldqkv, void* out, int ldo, int B, int F, int HW,
                                     int heads, float scale, vdx_stream_t stream) {
    VDX_CHECK(qkv && out, "temporal_attn: null pointer");
    VDX_CHECK(B > 0 && F > 0 && HW > 0 && heads > 0, "temporal_attn: empty problem");
    VDX_CHECK(F <= 32, "temporal_attn: F=%d frames per chunk exceeds 32", F);
    VDX_CHECK(ldqkv % 8 == 0 && ldqkv >= 3 * heads * 64 && ldo >= heads * 64, "temporal_attn: bad leading dims");
    TempP p;
    p.qkv = (const f16*)qkv; p.out = (f16*)out; p.ldqkv = ldqkv; p.ldo = ldo;
    p.B = B; p.F = F; p.HW = HW; p.heads = heads;
    p.c = scale * 1.44269504088896341f;
    p.items = (long long)B * HW * heads;
    const long long blocks = (p.items + 3) / 4;
    VDX_CHECK(blocks < (1ll << 31), "temporal_attn: too many items");
    hipLaunchKernelGGL(temporal_attn_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    return vdx_launch_status("vdx_temporal_attn_f16");
}
