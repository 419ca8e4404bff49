// attn.hip — attention cores on v_mfma_f32_32x32x16_f16 (SURVEY.md §2.3 K4/K5/K7 core).
//
// (the flash-style spatial / cross attention kernel lives in flash.hip)
//
// temporal_attn_kernel: TransformerTemporalModel's attention over the F <= 128 frames of one
//   latent pixel: one wave per (pixel, head, 32 query frames), operands straight from global memory,
//   ceil(F/32) 32x32 score tiles, complete softmax in registers, O = P.V.  HBM-bound by construction.
#include "attn_common.h"



// =============================================================================================
struct TempP {
    const f16* qkv;
    f16* out;
    int ldqkv, ldo, B, F, HW, heads, nqb;
    float c;
    long long items;
};

// NKB = key blocks of 32 frames (F <= 32*NKB); one wave per (pixel, head, block of 32 query frames).
// Chunks of the BASELINE configurations have F <= 24 (NKB = 1: a single score tile); NKB 2..4 serve
// `--mode fsdp` clips of up to 128 frames (fsdp_chunked_coherent.py:292,303-305 — no chunking there) and
// the monolithic same-T memory baseline.
template <int NKB>
__global__ __launch_bounds__(256) void temporal_attn_kernel(const TempP p) {
    const int lane = threadIdx.x & 63, r32 = lane & 31, h = lane >> 5;
    const long long item = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= p.items) return;
    const int qblk = (int)(item % p.nqb);
    const long long ph = item / p.nqb;
    const int head = (int)(ph % p.heads);
    const long long bp = ph / p.heads;
    const int pix = (int)(bp % p.HW), b = (int)(bp / p.HW);
    const int inner = p.heads * 64;
    const size_t row0 = (size_t)b * p.F * p.HW + pix;  // frame f lives at row0 + f*HW
    const f16* zp = (const f16*)g_zero_page;
    const int qf0 = qblk * 32;                          // first query frame of this wave

    // S^T = K.Q^T : A = K rows (fed through pi so the k-order of the next product is natural),
    // B = Q^T (lane = query frame)
    f32x16 s_acc[NKB];
    {
        const int qr = qf0 + r32;
        const f16* qsrc = p.qkv + (row0 + (size_t)qr * p.HW) * p.ldqkv + head * 64 + 8 * h;
        f16x8 qf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const f16x8*)(qr < p.F ? qsrc + 16 * ks : zp);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const int kr = 32 * kb + pi_row(r32);
            const f16* ksrc = p.qkv + (row0 + (size_t)kr * p.HW) * p.ldqkv + inner + head * 64 + 8 * h;
            f16x8 kf[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) kf[ks] = *(const f16x8*)(kr < p.F ? ksrc + 16 * ks : zp);
#pragma unroll
            for (int j = 0; j < 16; ++j) s_acc[kb][j] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                s_acc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks], qf[ks], s_acc[kb], 0, 0, 0);
        }
    }
    // V as B operand of O = P.V : lane = column d, element j of k-step s = V[16s + 8h + j][d]
    f16x8 vf[2][2 * NKB];
    {
        const f16* vsrc = p.qkv + row0 * p.ldqkv + 2 * inner + head * 64 + r32;
#pragma unroll
        for (int s = 0; s < 2 * NKB; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int key = 16 * s + 8 * h + j;
                const f16* src = vsrc + (size_t)key * p.HW * p.ldqkv;
                const bool ok = key < p.F;
                vf[0][s][j] = *(ok ? src : zp);
                vf[1][s][j] = *(ok ? src + 32 : zp);
            }
    }
    // complete softmax (all keys are in registers); normalise P before the second product
    float mx = NEG_BIG;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (32 * kb + acc_key(j, h) >= p.F) s_acc[kb][j] = NEG_BIG;
            mx = fmaxf(mx, s_acc[kb][j]);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mc = mx * p.c;
    float rs = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            s_acc[kb][j] = __builtin_amdgcn_exp2f(s_acc[kb][j] * p.c - mc);
            rs += s_acc[kb][j];
        }
    const float inv = 1.0f / (rs + __shfl_xor(rs, 32, 64));
    f16x8 pf[2 * NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int j = 0; j < 16; ++j) pf[2 * kb + (j >> 3)][j & 7] = (f16)(s_acc[kb][j] * inv);
    // O = P.V : A = P (accumulator as operand: X^T.B form), rows = query frames
#pragma unroll
    for (int db = 0; db < 2; ++db) {
        f32x16 o;
#pragma unroll
        for (int j = 0; j < 16; ++j) o[j] = 0.f;
#pragma unroll
        for (int s = 0; s < 2 * NKB; ++s) o = __builtin_amdgcn_mfma_f32_32x32x16_f16(pf[s], vf[db][s], o, 0, 0, 0);
        // C layout: col = lane&31 = d, row(reg) = (reg&3) + 8*(reg>>2) + 4*h = query frame
        f16* dst = p.out + row0 * p.ldo + head * 64 + 32 * db + r32;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int f = qf0 + (j & 3) + 8 * (j >> 2) + 4 * h;
            if (f < p.F) dst[(size_t)f * p.HW * p.ldo] = (f16)o[j];
        }
    }
}

extern "C" int vdx_temporal_attn_f16(const void* qkv, int ldqkv, void* out, int ldo, int B, int F, int HW,
                                     int heads, float scale, vdx_stream_t stream) {
    VDX_CHECK(qkv && out, "temporal_attn: null pointer");
    VDX_CHECK(B > 0 && F > 0 && HW > 0 && heads > 0, "temporal_attn: empty problem");
    VDX_CHECK(F <= 128, "temporal_attn: F=%d frames per chunk exceeds 128", F);
    VDX_CHECK(ldqkv % 8 == 0 && ldqkv >= 3 * heads * 64 && ldo >= heads * 64, "temporal_attn: bad leading dims");
    TempP p;
    p.qkv = (const f16*)qkv; p.out = (f16*)out; p.ldqkv = ldqkv; p.ldo = ldo;
    p.B = B; p.F = F; p.HW = HW; p.heads = heads;
    p.nqb = (F + 31) / 32;
    p.c = scale * 1.44269504088896341f;
    p.items = (long long)B * HW * heads * p.nqb;
    const long long blocks = (p.items + 3) / 4;
    VDX_CHECK(blocks < (1ll << 31), "temporal_attn: too many items");
    const dim3 grid((unsigned)blocks), blk(256);
    hipStream_t st = (hipStream_t)stream;
    switch (p.nqb) {
        case 1: hipLaunchKernelGGL(temporal_attn_kernel<1>, grid, blk, 0, st, p); break;
        case 2: hipLaunchKernelGGL(temporal_attn_kernel<2>, grid, blk, 0, st, p); break;
        case 3: hipLaunchKernelGGL(temporal_attn_kernel<3>, grid, blk, 0, st, p); break;
        default: hipLaunchKernelGGL(temporal_attn_kernel<4>, grid, blk, 0, st, p); break;
    }
    return vdx_launch_status("vdx_temporal_attn_f16");
}
