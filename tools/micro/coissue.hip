// Dev tool: what two waves of ONE SIMD cost each other on gfx950 — the model behind profiles/r04_flash.md.
// 512-thread blocks (waves w and w + 4 share a SIMD), one block per CU; waves 0-3 run role A, waves 4-7 role B; every wave
// stamps s_memtime around its loop.  Printed: cycles per instruction of each role alone, and of both when paired.
// Roles: 0 idle  1 v_mfma_f32_32x32x16_f16 (8 accumulators)  2 v_exp_f32  3 v_fma_f32  4 v_cvt_pk_f16_f32
//        5 v_dot2c_f32_f16 one chain  6 v_dot2c four chains  7 v_max3_f32 one chain  8 v_max3 four chains
//        9 v_mfma_f32_16x16x32_f16 (8 accumulators)  10 ds_read_b128 (conflict-free, 8 in flight)
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/coissue.hip -o tools/micro/coissue ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int role>
__device__ __forceinline__ void role_body(int iters, float* sink) {
    const int lane = threadIdx.x & 63;
    if (role == 30) {              // the softmax mix at raised priority
        asm volatile("s_setprio 3");
        role_body<24>(iters, sink);
        asm volatile("s_setprio 0");
        return;
    }
    if (role == 28 || role == 29 || role == 31) {      // a PACED MFMA stream: the wave does not present its next MFMA while the pipe is busy
        f16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(lane * 0.01f + e); b[e] = (_Float16)(e * 0.25f - lane * 0.003f); }
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
                    if (role == 28) asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
                    else if (role == 29) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
                    else asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += acc[i][lane & 15];
        *sink = s;
        return;
    }
    if (role == 1) {
        f16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(lane * 0.01f + e); b[e] = (_Float16)(e * 0.25f - lane * 0.003f); }
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += acc[i][lane & 15];
        *sink = s;
    } else if (role == 9) {
        f16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(lane * 0.01f + e); b[e] = (_Float16)(e * 0.25f - lane * 0.003f); }
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += acc[i][lane & 3];
        *sink = s;
    } else if (role == 27) {
        // the same 164 instructions per tile as role 24, ordered so that no instruction reads a result younger than 8
        // instructions: 36 v_max3 in 8 chains, then 64 v_exp, then 32 v_cvt_pk, then 32 v_dot2c in 8 chains
        float sc[64];
        for (int i = 0; i < 64; ++i) sc[i] = lane * 0.001f + i * 0.01f;
        float l[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        unsigned pk[32];
        for (int it = 0; it < iters; ++it) {
            float m[8] = {-1e30f, -1e30f, -1e30f, -1e30f, -1e30f, -1e30f, -1e30f, -1e30f};
#pragma unroll
            for (int i = 0; i < 64; i += 2) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m[(i >> 1) & 7]) : "v"(sc[i]), "v"(sc[i + 1]));
            asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m[0]) : "v"(m[1]), "v"(m[2]));
            asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m[3]) : "v"(m[4]), "v"(m[5]));
            asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m[6]) : "v"(m[7]), "v"(m[0]));
            asm volatile("v_max_f32 %0, %0, %1" : "+v"(m[6]) : "v"(m[3]));
            float e[64];
#pragma unroll
            for (int i = 0; i < 64; ++i) asm volatile("v_exp_f32 %0, %1" : "=v"(e[i]) : "v"(sc[i]));
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk[i]) : "v"(e[2 * i]), "v"(e[2 * i + 1]));
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(l[i & 7]) : "v"(pk[i]), "v"(0x3c003c00u));
            sc[0] += m[6] * 1e-30f;
        }
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += l[i];
        for (int i = 0; i < 32; ++i) s += (float)pk[i];
        *sink = s;
    } else if (role == 26) {       // the same MFMA stream with its accumulators in AccVGPRs
        f16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(lane * 0.01f + e); b[e] = (_Float16)(e * 0.25f - lane * 0.003f); }
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
        }
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += acc[i][lane & 15];
        *sink = s;
    } else if (role == 32) {
        // ONE wave carrying both: the tile's 164 vector instructions with its 32 MFMAs spliced in, one MFMA per ~5 vector
        // instructions (inline asm on both sides, so the order below is the issue order): does a wave's own vector work run
        // in the shadow of its own MFMAs?  (per tile: MFMA 1024 cycles, vector mix ~1240 alone)
        float sc[64];
        for (int i = 0; i < 64; ++i) sc[i] = lane * 0.001f + i * 0.01f;
        float l0 = 0.f, l1 = 0.f, l2 = 0.f, l3 = 0.f;
        unsigned pk[32];
        f16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(lane * 0.01f + e); b[e] = (_Float16)(e * 0.25f - lane * 0.003f); }
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
#define MF_(k) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[(k) & 7]) : "v"(a), "v"(b))
        for (int it = 0; it < iters; ++it) {
            float m0 = -1e30f, m1 = -1e30f, m2 = -1e30f, m3 = -1e30f;
#pragma unroll
            for (int i = 0; i < 64; i += 8) {
                if (i < 48) MF_(i >> 3);                 // 6 MFMAs among the 34 max instructions
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m0) : "v"(sc[i]), "v"(sc[i + 1]));
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m1) : "v"(sc[i + 2]), "v"(sc[i + 3]));
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m2) : "v"(sc[i + 4]), "v"(sc[i + 5]));
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m3) : "v"(sc[i + 6]), "v"(sc[i + 7]));
            }
            asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m0) : "v"(m1), "v"(m2));
            asm volatile("v_max_f32 %0, %0, %1" : "+v"(m0) : "v"(m3));
#pragma unroll
            for (int i = 0; i < 64; i += 2) {
                float x, y;
                if ((i >> 1) % 5 != 4 || i == 58) MF_(i >> 1);     // 26 MFMAs among the 128 exp / cvt / dot2c
                asm volatile("v_exp_f32 %0, %1" : "=v"(x) : "v"(sc[i]));
                asm volatile("v_exp_f32 %0, %1" : "=v"(y) : "v"(sc[i + 1]));
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk[i >> 1]) : "v"(x), "v"(y));
                if ((i & 6) == 0) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(l0) : "v"(pk[i >> 1]), "v"(0x3c003c00u));
                else if ((i & 6) == 2) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(l1) : "v"(pk[i >> 1]), "v"(0x3c003c00u));
                else if ((i & 6) == 4) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(l2) : "v"(pk[i >> 1]), "v"(0x3c003c00u));
                else asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(l3) : "v"(pk[i >> 1]), "v"(0x3c003c00u));
            }
            sc[0] += m0 * 1e-30f;
        }
#undef MF_
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
        float s = l0 + l1 + l2 + l3;
        for (int i = 0; i < 32; ++i) s += (float)pk[i];
        for (int i = 0; i < 8; ++i) s += acc[i][lane & 15];
        *sink = s;
    } else if (role == 33) {
        // (role 33: the same with 64 v_mfma_f32_16x16x32_f16 — the same flops, a quarter of the accumulator registers per instruction)
        // ONE wave carrying both: the tile's 164 vector instructions with its 32 MFMAs spliced in, one MFMA per ~5 vector
        // instructions (inline asm on both sides, so the order below is the issue order): does a wave's own vector work run
        // in the shadow of its own MFMAs?  (per tile: MFMA 1024 cycles, vector mix ~1240 alone)
        float sc[64];
        for (int i = 0; i < 64; ++i) sc[i] = lane * 0.001f + i * 0.01f;
        float l0 = 0.f, l1 = 0.f, l2 = 0.f, l3 = 0.f;
        unsigned pk[32];
        f16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(lane * 0.01f + e); b[e] = (_Float16)(e * 0.25f - lane * 0.003f); }
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#define MF1_(k) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[(k) & 15]) : "v"(a), "v"(b))
#define MF_(k) do { MF1_(2 * (k)); MF1_(2 * (k) + 1); } while (0)
        for (int it = 0; it < iters; ++it) {
            float m0 = -1e30f, m1 = -1e30f, m2 = -1e30f, m3 = -1e30f;
#pragma unroll
            for (int i = 0; i < 64; i += 8) {
                if (i < 48) MF_(i >> 3);                 // 6 MFMAs among the 34 max instructions
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m0) : "v"(sc[i]), "v"(sc[i + 1]));
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m1) : "v"(sc[i + 2]), "v"(sc[i + 3]));
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m2) : "v"(sc[i + 4]), "v"(sc[i + 5]));
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m3) : "v"(sc[i + 6]), "v"(sc[i + 7]));
            }
            asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m0) : "v"(m1), "v"(m2));
            asm volatile("v_max_f32 %0, %0, %1" : "+v"(m0) : "v"(m3));
#pragma unroll
            for (int i = 0; i < 64; i += 2) {
                float x, y;
                if ((i >> 1) % 5 != 4 || i == 58) MF_(i >> 1);     // 26 MFMAs among the 128 exp / cvt / dot2c
                asm volatile("v_exp_f32 %0, %1" : "=v"(x) : "v"(sc[i]));
                asm volatile("v_exp_f32 %0, %1" : "=v"(y) : "v"(sc[i + 1]));
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk[i >> 1]) : "v"(x), "v"(y));
                if ((i & 6) == 0) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(l0) : "v"(pk[i >> 1]), "v"(0x3c003c00u));
                else if ((i & 6) == 2) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(l1) : "v"(pk[i >> 1]), "v"(0x3c003c00u));
                else if ((i & 6) == 4) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(l2) : "v"(pk[i >> 1]), "v"(0x3c003c00u));
                else asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(l3) : "v"(pk[i >> 1]), "v"(0x3c003c00u));
            }
            sc[0] += m0 * 1e-30f;
        }
#undef MF_
#undef MF1_
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
        float s = l0 + l1 + l2 + l3;
        for (int i = 0; i < 32; ++i) s += (float)pk[i];
        for (int i = 0; i < 16; ++i) s += acc[i][lane & 3];
        *sink = s;
    } else if (role == 24) {
        // the vector segment's instruction mix per 64-key tile: 36 v_max3 (4 chains) + 64 v_exp + 32 v_cvt_pk + 32 v_dot2c (4 chains),
        // on 64 distinct "score" registers as in flash.hip
        float sc[64];
        for (int i = 0; i < 64; ++i) sc[i] = lane * 0.001f + i * 0.01f;
        float l0 = 0.f, l1 = 0.f, l2 = 0.f, l3 = 0.f;
        unsigned pk[32];
        for (int it = 0; it < iters; ++it) {
            float m0 = -1e30f, m1 = -1e30f, m2 = -1e30f, m3 = -1e30f;
#pragma unroll
            for (int i = 0; i < 64; i += 8) {
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m0) : "v"(sc[i]), "v"(sc[i + 1]));
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m1) : "v"(sc[i + 2]), "v"(sc[i + 3]));
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m2) : "v"(sc[i + 4]), "v"(sc[i + 5]));
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m3) : "v"(sc[i + 6]), "v"(sc[i + 7]));
            }
            asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m0) : "v"(m1), "v"(m2));
            asm volatile("v_max_f32 %0, %0, %1" : "+v"(m0) : "v"(m3));
#pragma unroll
            for (int i = 0; i < 64; i += 2) {
                float a, b;
                asm volatile("v_exp_f32 %0, %1" : "=v"(a) : "v"(sc[i]));
                asm volatile("v_exp_f32 %0, %1" : "=v"(b) : "v"(sc[i + 1]));
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk[i >> 1]) : "v"(a), "v"(b));
                if ((i & 6) == 0) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(l0) : "v"(pk[i >> 1]), "v"(0x3c003c00u));
                else if ((i & 6) == 2) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(l1) : "v"(pk[i >> 1]), "v"(0x3c003c00u));
                else if ((i & 6) == 4) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(l2) : "v"(pk[i >> 1]), "v"(0x3c003c00u));
                else asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(l3) : "v"(pk[i >> 1]), "v"(0x3c003c00u));
            }
            sc[0] += m0 * 1e-30f;
        }
        float s = l0 + l1 + l2 + l3;
        for (int i = 0; i < 32; ++i) s += (float)pk[i];
        *sink = s;
    } else if (role == 10) {
        extern __shared__ char lds[];
        f32x4 v[8];
        const unsigned addr = (unsigned)(lane * 16);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[i]) : "v"(addr), "n"(1024 * i));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += v[i][0];
        *sink = s;
    } else {
        float x[16];
        for (int i = 0; i < 16; ++i) x[i] = lane * 0.001f + i * 0.01f;
        float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
        unsigned pk[16];
        for (int i = 0; i < 16; ++i) pk[i] = 0x3c003800u + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (role == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
                    else if (role == 3) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[i]));
                    else if (role == 4) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk[i]) : "v"(x[i]), "v"(x[(i + 1) & 15]));
                    else if (role == 5) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(c0) : "v"(pk[i]), "v"(pk[(i + 1) & 15]));
                    else if (role == 6) {
                        if ((i & 3) == 0) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(c0) : "v"(pk[i]), "v"(pk[(i + 1) & 15]));
                        else if ((i & 3) == 1) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(c1) : "v"(pk[i]), "v"(pk[(i + 1) & 15]));
                        else if ((i & 3) == 2) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(c2) : "v"(pk[i]), "v"(pk[(i + 1) & 15]));
                        else asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(c3) : "v"(pk[i]), "v"(pk[(i + 1) & 15]));
                    } else if (role == 11) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(pk[i]) : "v"(x[i]), "v"(x[(i + 1) & 15]));
                    else if (role == 12) asm volatile("v_exp_f16 %0, %0" : "+v"(pk[i]));
                    else if (role == 13) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(pk[i]) : "v"(pk[(i + 1) & 15]));
                    else if (role == 14) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double*)&x[2 * (i & 7)]) : "v"(*(double*)&x[2 * ((i + 1) & 7)]));
                    else if (role == 15) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(x[(i + 1) & 15]));
                    else if (role == 16) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(pk[i]) : "v"(pk[(i + 1) & 15]));
                    else if (role == 17) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(x[i]) : "v"(pk[i]), "v"(pk[(i + 1) & 15]));
                    else if (role == 18) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(pk[i]) : "v"(pk[(i + 1) & 15]), "v"(pk[(i + 2) & 15]));
                    else if (role == 19) asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(pk[i]) : "v"(x[i]));
                    else if (role == 20) asm volatile("v_mov_b32 %0, %1" : "=v"(pk[i]) : "v"(x[i]));
                    else if (role == 21) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(x[(i + 1) & 15]));
                    else if (role == 22) asm volatile("v_pk_fma_f16 %0, %0, %1, %1" : "+v"(pk[i]) : "v"(pk[(i + 1) & 15]));
                    else if (role == 23) asm volatile("v_exp_f32 %0, %1" : "=v"(x[i]) : "v"(x[(i + 5) & 15]));
                    else if (role == 7) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(c0) : "v"(x[i]), "v"(x[(i + 1) & 15]));
                    else if (role == 8) {
                        if ((i & 3) == 0) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(c0) : "v"(x[i]), "v"(x[(i + 1) & 15]));
                        else if ((i & 3) == 1) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(c1) : "v"(x[i]), "v"(x[(i + 1) & 15]));
                        else if ((i & 3) == 2) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(c2) : "v"(x[i]), "v"(x[(i + 1) & 15]));
                        else asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(c3) : "v"(x[i]), "v"(x[(i + 1) & 15]));
                    }
                }
            }
        }
        float s = c0 + c1 + c2 + c3;
        for (int i = 0; i < 16; ++i) s += x[i] + (float)pk[i];
        *sink = s;
    }
}

template <int roleA, int roleB>
__global__ __launch_bounds__(512) void k(int itA, int itB, unsigned long long* cyc, float* out) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float sink = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        if (roleA != 0) role_body<roleA>(itA, &sink);
    } else {
        if (roleB != 0) role_body<roleB>(itB, &sink);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    out[blockIdx.x * 512 + threadIdx.x] = sink;
}

static const char* NAMES[] = {"idle", "mfma 32x32x16", "v_exp_f32", "v_fma_f32", "v_cvt_pk_f16_f32", "v_dot2c 1 chain", "v_dot2c 4 chains",
                              "v_max3 1 chain", "v_max3 4 chains", "mfma 16x16x32", "ds_read_b128 x8", "v_cvt_pkrtz_f16_f32", "v_exp_f16", "v_pk_add_f16", "v_pk_mul_f32", "v_max_f32",
                              "v_pk_max_f16", "v_dot2_f32_f16 16ch", "v_perm_b32", "v_cvt_f16_f32", "v_mov_b32", "v_add_f32", "v_pk_fma_f16", "v_exp_f32 indep", "softmax mix x164", "?", "mfma 32x32x16 agpr", "softmax batched x164", "mfma + 16 nop", "mfma + 24 nop", "mix @prio 3", "mfma + 28 nop", "mix + own 32 mfma", "mix + own 64 mfma16"};

template <int roleA, int roleB>
static void run(int itA, int itB) {
    const int blocks = 256;
    unsigned long long* cyc;
    float* out;
    hipMalloc(&cyc, blocks * 8 * 8);
    hipMalloc(&out, blocks * 512 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<roleA, roleB>), dim3(blocks), dim3(512), 65536, 0, itA, itB, cyc, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned long long> h(blocks * 8);
    hipMemcpy(h.data(), cyc, blocks * 8 * 8, hipMemcpyDeviceToHost);
    std::vector<double> a, b;
    for (int i = 0; i < blocks; ++i) for (int w = 0; w < 8; ++w) (w < 4 ? a : b).push_back((double)h[i * 8 + w]);
    std::sort(a.begin(), a.end());
    std::sort(b.begin(), b.end());
    const double ma = a[a.size() / 2], mb = b[b.size() / 2];
    const double nA = roleA ? (roleA == 24 || roleA == 27 || roleA == 30 || roleA == 32 || roleA == 33 ? 164.0 : 64.0) * itA : 1, nB = roleB ? (roleB == 24 || roleB == 27 || roleB == 30 || roleB == 32 || roleB == 33 ? 164.0 : 64.0) * itB : 1;
    printf("A %-18s B %-18s : A %7.2f cyc/instr   B %7.2f cyc/instr   (%.3f ms; clock ~%.2f GHz)\n", NAMES[roleA], NAMES[roleB],
           roleA ? ma / nA : 0.0, roleB ? mb / nB : 0.0, ms, std::max(ma, mb) / ms / 1e6);
    hipFree(cyc); hipFree(out);
}

int main(int argc, char** argv) {
    const int N = 4000;
#define ALONE(r) run<r, 0>(N, 0)
#define SAME(r) run<r, r>(N, N)
#define MF(r) run<1, r>(N, N)
    if (argc > 5) {           // a wave's own vector work in the shadow of its own MFMAs (per tile = 164 x the printed figure)
        run<24, 0>(N, 0); run<32, 0>(N, 0); run<32, 32>(N, N); run<1, 0>(N / 2, 0);
        run<33, 0>(N, 0); run<33, 33>(N, N); run<9, 0>(N, 0);
        return 0;
    }
    if (argc > 4) {           // paced MFMA streams (s_nop behind every MFMA) and priorities beside the softmax mix
        run<28, 0>(N, 0); run<29, 0>(N, 0); run<31, 0>(N, 0);
        run<28, 24>(N, N); run<24, 28>(N, N);
        run<29, 24>(N, N); run<24, 29>(N, N);
        run<31, 24>(N, N); run<24, 31>(N, N);
        run<29, 2>(N, 4 * N); run<2, 29>(4 * N, N);
        run<1, 30>(N, N); run<30, 1>(N, N);
        run<29, 30>(N, N); run<30, 29>(N, N);
        return 0;
    }
    if (argc > 3) {           // each vector instruction as the OLDER wave beside a longer MFMA stream
        run<2, 1>(2 * N, N); run<3, 1>(2 * N, N); run<4, 1>(2 * N, N); run<6, 1>(2 * N, N); run<8, 1>(2 * N, N); run<15, 1>(2 * N, N);
        run<21, 1>(2 * N, N); run<13, 1>(2 * N, N); run<17, 1>(2 * N, N); run<11, 1>(2 * N, N); run<23, 1>(2 * N, N);
        return 0;
    }
    if (argc > 2) {           // the flash tile's vector mix beside an MFMA stream, either age order; 164 instr per iteration, MFMA 64
        run<24, 0>(N, 0);
        run<0, 24>(0, N);
        run<24, 24>(N, N);
        run<24, 1>(N, N * 164 / 64 / 4);     // vector wave older; MFMA wave with ~equal duration (32 cyc x 64 x it = 6.4 x 164 x N)
        run<1, 24>(N * 164 / 64 / 4, N);     // MFMA wave older
        run<24, 1>(N, N);                    // MFMA wave runs much longer
        run<1, 24>(N, N);
        run<27, 0>(N, 0);
        run<27, 27>(N, N);
        run<27, 1>(N, N);
        run<1, 27>(N, N);
        run<26, 0>(N, 0);
        run<24, 26>(N, N);                   // accumulators in AccVGPRs
        run<26, 24>(N, N);
        run<2, 26>(4 * N, N);
        run<2, 1>(4 * N, N);
        return 0;
    }
    if (argc > 1) {           // instruction costs, one wave per SIMD alone
        ALONE(2); ALONE(23); ALONE(3); ALONE(4); ALONE(11); ALONE(19); ALONE(12); ALONE(13); ALONE(22); ALONE(14); ALONE(15); ALONE(16); ALONE(17); ALONE(18);
        ALONE(20); ALONE(21); ALONE(5); ALONE(6); ALONE(7); ALONE(8);
        SAME(11); SAME(15); SAME(21);
        return 0;
    }
    ALONE(1); ALONE(2); ALONE(3); ALONE(4); ALONE(5); ALONE(6); ALONE(7); ALONE(8); ALONE(9); ALONE(10);
    SAME(1); SAME(2); SAME(3); SAME(4); SAME(5); SAME(6); SAME(7); SAME(8); SAME(9); SAME(10);
    MF(2); MF(3); MF(4); MF(5); MF(6); MF(7); MF(8); MF(9); MF(10);
    run<9, 2>(N, N);
    run<1, 2>(N, 4 * N);      // MFMA wave beside a 4x longer exp wave (the vector wave never runs out)
    run<1, 3>(N, 8 * N);
    run<2, 1>(4 * N, N);      // the vector wave is the OLDER one (waves 0-3)
    run<3, 1>(8 * N, N);
    return 0;
}
