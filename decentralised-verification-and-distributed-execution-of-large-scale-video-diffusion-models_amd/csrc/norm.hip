// norm.hip — GroupNorm (4-D and 5-D, optional 2-source concat, optional SiLU) and LayerNorm
// over channels-last fp16 rows.  HBM-bound streaming kernels:
//   GroupNorm = stats pass (1 read) + tiny finalize + apply pass (1 read + 1 write)
//   LayerNorm = 1 read + 1 write, one wave per row, row cached in registers.
// Thread mapping: a block owns a slab of rows of ONE sample; the block size is a multiple of
// the number of 8-channel vectors per row, so every thread keeps a fixed channel vector
// (coalesced 16-byte accesses, per-channel scale/shift held in registers).
#include "vdx_common.h"

// Rows of one sample per block: 128 for the big levels, halved until the grid has >= min_blocks blocks so the deep
// levels (6912 rows in all at level 3) still spread over the 256 CUs.  The statistics pass stops at 256 blocks (one
// per CU: its per-block reduction epilogue costs as much as ~64 rows of streaming — 21 -> 15 us at level 2), the
// apply pass, which has no epilogue, at 1024 (28 us at level 2 against 35 with the coarser split).
static int gn_rows_for(int n_samples, int rows_per_sample, int min_blocks) {
    int rows = 128;
    while (rows > 8 && (long long)n_samples * ((rows_per_sample + rows - 1) / rows) < min_blocks) rows >>= 1;
    return rows;
}
static int gn_stat_rows(int n_samples, int rows_per_sample) { return gn_rows_for(n_samples, rows_per_sample, 256); }

struct GnP {
    const f16 *x, *x2;
    int c1, c2, ldx, ldx2;
    int C, G, cpg, nvec, krows;      // nvec = C/8, krows = rows processed in parallel per block
    int n_samples, rps, nslabs, slab_rows;   // rps = rows per sample; slabs = row partition of the STATISTICS pass
    int apply_rows;                          // rows of a sample per block of the apply pass
    float* partial;                  // [n_samples][nslabs][G][3]  (count, mean, M2)
    float* ab;                       // [n_samples][C][2]          (scale, shift)
};

__device__ __forceinline__ f16x8 gn_load(const GnP& p, size_t row, int cv) {
    const int c = cv * 8;
    const f16* src = c < p.c1 ? p.x + row * p.ldx + c : p.x2 + row * p.ldx2 + (c - p.c1);
    return *(const f16x8*)src;
}

// Statistics are (count, mean, M2 = sum of squared deviations) triples merged pairwise (Chan et al.):
//   n = na + nb,  d = mb - ma,  mean = ma + d * nb / n,  M2 = M2a + M2b + d^2 * na * nb / n.
// A thread sums its rows SHIFTED by the first value it sees per channel (sum (x - k), sum (x - k)^2), so nothing of
// the size of the mean is ever squared: E[x^2] - mean^2 from plain fp32 sums loses the variance of a group whose mean
// is a few hundred standard deviations out (large-mean channels of trained checkpoints); this form does not.
struct Moments {
    float n, mean, m2;
};
__device__ __forceinline__ Moments merge(const Moments a, const Moments b) {
    if (b.n == 0.f) return a;
    if (a.n == 0.f) return b;
    const float n = a.n + b.n, d = b.mean - a.mean;
    return Moments{n, a.mean + d * (b.n / n), a.m2 + b.m2 + d * d * (a.n * b.n / n)};
}

__global__ void gn_partial_kernel(const GnP p) {
    // [2][krows][C] per-thread channel (mean, M2) + [krows] row counts; reduced to groups in a FIXED order (no float
    // atomics: the statistics, and therefore every output of the network, are bitwise reproducible)
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    const int sample = blockIdx.y, slab = blockIdx.x;
    const int cv = tid % p.nvec, rsub = tid / p.nvec;
    float k[8], s[8], ss[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) k[j] = s[j] = ss[j] = 0.f;
    const int r_end = min(p.rps, (slab + 1) * p.slab_rows);
    const size_t base = (size_t)sample * p.rps;
    int cnt = 0;
    int r = slab * p.slab_rows + rsub;
    auto acc = [&](const f16x8 v) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float f = (float)v[j] - k[j];
            s[j] += f;
            ss[j] += f * f;
        }
    };
    if (r < r_end) {                                        // the first row sets the shift
        const f16x8 v = gn_load(p, base + r, cv);
#pragma unroll
        for (int j = 0; j < 8; ++j) k[j] = (float)v[j];
        acc(v);
        r += p.krows;
        cnt = 1;
    }
    // four rows per trip, loads first: one 16-byte load in flight per thread leaves the kernel latency-bound
    // (3.5 TB/s at level 0); the accumulation order per thread is unchanged (row after row)
    for (; r + 3 * p.krows < r_end; r += 4 * p.krows, cnt += 4) {
        const f16x8 v0 = gn_load(p, base + r, cv), v1 = gn_load(p, base + r + p.krows, cv);
        const f16x8 v2 = gn_load(p, base + r + 2 * p.krows, cv), v3 = gn_load(p, base + r + 3 * p.krows, cv);
        acc(v0); acc(v1); acc(v2); acc(v3);
    }
    for (; r < r_end; r += p.krows, ++cnt) acc(gn_load(p, base + r, cv));
    const int plane = p.krows * p.C;
    const float inv = cnt ? 1.0f / (float)cnt : 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        lds[rsub * p.C + cv * 8 + j] = k[j] + s[j] * inv;                     // mean of this thread's rows
        lds[plane + rsub * p.C + cv * 8 + j] = ss[j] - s[j] * s[j] * inv;     // their M2 (shifted: no cancellation)
    }
    if (cv == 0) lds[2 * plane + rsub] = (float)cnt;
    __syncthreads();
    // (group, row subset) pairs first: the cpg channel triples of one row subset all have the same count, so their
    // merge needs no division per step: mean = average of the means, M2 = sum M2 + n * sum (mean_c - mean)^2
    float* tmp = lds + 2 * plane + p.krows;                 // [G * krows][2]
    for (int i = tid; i < p.G * p.krows; i += blockDim.x) {
        const int g = i / p.krows, r = i - g * p.krows;
        const float* mp = lds + r * p.C + g * p.cpg;
        const float* qp = lds + plane + r * p.C + g * p.cpg;
        float ms = 0.f, qs = 0.f;
        for (int c = 0; c < p.cpg; ++c) {
            ms += mp[c];
            qs += qp[c];
        }
        const float mean = ms / (float)p.cpg;
        float dev = 0.f;
        for (int c = 0; c < p.cpg; ++c) {
            const float d = mp[c] - mean;
            dev += d * d;
        }
        tmp[2 * i] = mean;
        tmp[2 * i + 1] = qs + lds[2 * plane + r] * dev;
    }
    __syncthreads();
    float* dst = p.partial + ((size_t)sample * p.nslabs + slab) * 3 * p.G;
    for (int g = tid; g < p.G; g += blockDim.x) {
        Moments m{0.f, 0.f, 0.f};
        for (int r = 0; r < p.krows; ++r)
            m = merge(m, Moments{lds[2 * plane + r] * (float)p.cpg, tmp[2 * (g * p.krows + r)], tmp[2 * (g * p.krows + r) + 1]});
        dst[3 * g] = m.n;
        dst[3 * g + 1] = m.mean;
        dst[3 * g + 2] = m.m2;
    }
}

// one wave per (sample, group): merge the slab triples in double, emit per-channel scale/shift
__global__ void gn_finalize_kernel(const GnP p, const f16* gamma, const f16* beta, float eps) {
    const int lane = threadIdx.x & 63;
    const int sg = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (sg >= p.n_samples * p.G) return;
    const int sample = sg / p.G, g = sg % p.G;
    // Slab triples -> one (n, mean, M2) per (sample, group), in double and without a division per slab: with every
    // mean taken relative to K = the mean of slab 0,
    //     N = sum n_i,  S1 = sum n_i d_i,  S2 = sum n_i d_i^2,  Q = sum M2_i      (d_i = mean_i - K)
    //     mean = K + S1 / N,   M2 = Q + S2 - S1^2 / N
    // which is Chan's merge summed up; d_i is a few standard deviations at most, so nothing cancels at double width.
    // Fixed order (lane-strided, then a butterfly): bitwise reproducible.
    const float* src0 = p.partial + (size_t)sample * p.nslabs * 3 * p.G + 3 * g;
    const size_t stride = (size_t)3 * p.G;
    const double K = (double)src0[1];
    double N = 0.0, S1 = 0.0, S2 = 0.0, Q = 0.0;
    auto add = [&](float nf, float mf, float qf) {
        const double nn = (double)nf, d = (double)mf - K;
        N += nn;
        S1 += nn * d;
        S2 += nn * d * d;
        Q += (double)qf;
    };
    int i = lane;
    for (; i + 192 < p.nslabs; i += 256) {                  // four slabs per trip, their loads in flight together
        const float* a0 = src0 + (size_t)i * stride;
        const float* a1 = a0 + 64 * stride;
        const float* a2 = a1 + 64 * stride;
        const float* a3 = a2 + 64 * stride;
        const float n0 = a0[0], m0 = a0[1], q0 = a0[2], n1 = a1[0], m1 = a1[1], q1 = a1[2];
        const float n2 = a2[0], m2_ = a2[1], q2 = a2[2], n3 = a3[0], m3 = a3[1], q3 = a3[2];
        add(n0, m0, q0); add(n1, m1, q1); add(n2, m2_, q2); add(n3, m3, q3);
    }
    for (; i < p.nslabs; i += 64) {
        const float* a0 = src0 + (size_t)i * stride;
        add(a0[0], a0[1], a0[2]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {                      // both partners add (low, high) in the same order
        const double Nb = __shfl_xor(N, o, 64), S1b = __shfl_xor(S1, o, 64), S2b = __shfl_xor(S2, o, 64), Qb = __shfl_xor(Q, o, 64);
        const bool low = (lane & o) == 0;
        N = low ? N + Nb : Nb + N;
        S1 = low ? S1 + S1b : S1b + S1;
        S2 = low ? S2 + S2b : S2b + S2;
        Q = low ? Q + Qb : Qb + Q;
    }
    const double n = N, mean = N > 0.0 ? K + S1 / N : 0.0, m2 = N > 0.0 ? Q + S2 - S1 * S1 / N : 0.0;
    double var = n > 0.0 ? m2 / n : 0.0;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    for (int j = lane; j < p.cpg; j += 64) {
        const int c = g * p.cpg + j;
        const float a = rstd * (float)gamma[c];
        float* dst = p.ab + ((size_t)sample * p.C + c) * 2;
        dst[0] = a;
        dst[1] = (float)beta[c] - (float)mean * a;
    }
}

template <bool SILU>
__global__ void gn_apply_kernel(const GnP p, f16* y, int ldy) {
    const int tid = threadIdx.x;
    const int sample = blockIdx.y, slab = blockIdx.x;
    const int cv = tid % p.nvec, rsub = tid / p.nvec;
    float a[8], b[8];
    const float* ab = p.ab + ((size_t)sample * p.C + cv * 8) * 2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = ab[2 * j];
        b[j] = ab[2 * j + 1];
    }
    const int r_end = min(p.rps, (slab + 1) * p.apply_rows);
    const size_t base = (size_t)sample * p.rps;
    for (int r = slab * p.apply_rows + rsub; r < r_end; r += p.krows) {
        const size_t row = base + r;
        const f16x8 v = gn_load(p, row, cv);
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float f = (float)v[j] * a[j] + b[j];
            if (SILU) f = silu_f(f);
            o[j] = (f16)f;
        }
        *(f16x8*)(y + row * ldy + cv * 8) = o;
    }
}

static int gn_threads(int nvec, int* krows) {
    int k = nvec >= 256 ? 1 : (256 + nvec - 1) / nvec;
    *krows = k;
    return k * nvec;
}

// `partition_samples` (> 0) fixes the row-slab partition as if the call had that many samples: a sample's statistics
// are then bit-identical whether it is normalised alone, in a half batch or in the full batch (the slab size otherwise
// follows the batch size to fill the chip).  0 = use n_samples.
extern "C" size_t vdx_groupnorm_workspace_part(int n_samples, int rows_per_sample, int C, int G, int partition_samples) {
    const int rows = gn_stat_rows(partition_samples > 0 ? partition_samples : n_samples, rows_per_sample);
    const size_t nslabs = (rows_per_sample + rows - 1) / rows;
    return ((((size_t)n_samples * nslabs * G * 3 + 3) & ~(size_t)3) + (size_t)n_samples * C * 2) * sizeof(float);
}
extern "C" size_t vdx_groupnorm_workspace(int n_samples, int rows_per_sample, int C, int G) {
    return vdx_groupnorm_workspace_part(n_samples, rows_per_sample, C, G, 0);
}

extern "C" int vdx_groupnorm_part_f16(const void* x, int c1, int ldx, const void* x2, int c2, int ldx2,
                                      const void* gamma, const void* beta, float eps, int G,
                                      int n_samples, int rows_per_sample, int silu, void* y, int ldy,
                                      void* workspace, int partition_samples, vdx_stream_t stream);
extern "C" int vdx_groupnorm_f16(const void* x, int c1, int ldx, const void* x2, int c2, int ldx2,
                                 const void* gamma, const void* beta, float eps, int G,
                                 int n_samples, int rows_per_sample, int silu, void* y, int ldy,
                                 void* workspace, vdx_stream_t stream) {
    return vdx_groupnorm_part_f16(x, c1, ldx, x2, c2, ldx2, gamma, beta, eps, G, n_samples, rows_per_sample, silu, y, ldy,
                                  workspace, 0, stream);
}
// statistics (+ per-channel scale / shift) always; the apply pass unless `stats_only` (the fold into a following Linear)
static int gn_run(const void* x, int c1, int ldx, const void* x2, int c2, int ldx2,
                  const void* gamma, const void* beta, float eps, int G,
                  int n_samples, int rows_per_sample, int silu, void* y, int ldy,
                  void* workspace, int partition_samples, bool stats_only, GnP& p, vdx_stream_t stream) {
    VDX_CHECK(x && gamma && beta && (y || stats_only) && workspace, "groupnorm: null pointer");
    const int C = c1 + c2;
    VDX_CHECK(c1 > 0 && c1 % 8 == 0 && c2 % 8 == 0, "groupnorm: c1=%d c2=%d must be multiples of 8", c1, c2);
    VDX_CHECK((c2 == 0) == (x2 == nullptr), "groupnorm: x2/c2 mismatch");
    VDX_CHECK(G > 0 && C % G == 0, "groupnorm: C=%d not divisible by G=%d", C, G);
    VDX_CHECK(ldx % 8 == 0 && ldy % 8 == 0 && (c2 == 0 || ldx2 % 8 == 0), "groupnorm: leading dims must be multiples of 8");
    VDX_CHECK(n_samples > 0 && rows_per_sample > 0, "groupnorm: empty input");
    VDX_CHECK(C / 8 <= 1024, "groupnorm: C=%d too wide", C);
    p.x = (const f16*)x; p.x2 = (const f16*)x2; p.c1 = c1; p.c2 = c2; p.ldx = ldx; p.ldx2 = ldx2;
    p.C = C; p.G = G; p.cpg = C / G; p.nvec = C / 8;
    p.n_samples = n_samples; p.rps = rows_per_sample;
    p.slab_rows = gn_stat_rows(partition_samples > 0 ? partition_samples : n_samples, rows_per_sample);
    p.apply_rows = gn_rows_for(n_samples, rows_per_sample, 1024);
    p.nslabs = (rows_per_sample + p.slab_rows - 1) / p.slab_rows;
    p.partial = (float*)workspace;
    p.ab = p.partial + (((size_t)n_samples * p.nslabs * G * 3 + 3) & ~(size_t)3);     // 16-byte aligned: K1 / K3 read it with 16-byte loads
    const int nt = gn_threads(p.nvec, &p.krows);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(p.nslabs, n_samples);
    hipLaunchKernelGGL(gn_partial_kernel, grid, dim3(nt), (2 * (size_t)p.krows * C + p.krows + 2 * (size_t)G * p.krows) * sizeof(float), st, p);
    const int nsg = n_samples * G;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((nsg + 3) / 4), dim3(256), 0, st, p, (const f16*)gamma, (const f16*)beta, eps);
    if (stats_only) return vdx_launch_status("vdx_groupnorm_f16 (statistics)");
    dim3 agrid((rows_per_sample + p.apply_rows - 1) / p.apply_rows, n_samples);
    if (silu)
        hipLaunchKernelGGL(gn_apply_kernel<true>, agrid, dim3(nt), 0, st, p, (f16*)y, ldy);
    else
        hipLaunchKernelGGL(gn_apply_kernel<false>, agrid, dim3(nt), 0, st, p, (f16*)y, ldy);
    return vdx_launch_status("vdx_groupnorm_f16");
}
extern "C" int vdx_groupnorm_part_f16(const void* x, int c1, int ldx, const void* x2, int c2, int ldx2,
                                      const void* gamma, const void* beta, float eps, int G,
                                      int n_samples, int rows_per_sample, int silu, void* y, int ldy,
                                      void* workspace, int partition_samples, vdx_stream_t stream) {
    GnP p;
    return gn_run(x, c1, ldx, x2, c2, ldx2, gamma, beta, eps, G, n_samples, rows_per_sample, silu, y, ldy, workspace,
                  partition_samples, false, p, stream);
}

// statistics only: the per-(sample, channel) scale / shift pairs ([n_samples][C][2] fp32) are left in the workspace at
// *scale_shift_offset bytes — what a consumer that applies the normalisation itself reads (vdx_tconv_gn_f16)
extern "C" int vdx_groupnorm_stats_f16(const void* x, int c1, int ldx, const void* x2, int c2, int ldx2, const void* gamma,
                                       const void* beta, float eps, int G, int n_samples, int rows_per_sample, void* workspace,
                                       int partition_samples, size_t* scale_shift_offset, vdx_stream_t stream) {
    VDX_CHECK(scale_shift_offset, "groupnorm_stats: null pointer");
    GnP p;
    if (const int rc = gn_run(x, c1, ldx, x2, c2, ldx2, gamma, beta, eps, G, n_samples, rows_per_sample, 0, nullptr, 0, workspace,
                              partition_samples, true, p, stream))
        return rc;
    *scale_shift_offset = (size_t)((const char*)p.ab - (const char*)workspace);
    return 0;
}

// ---- GroupNorm folded into the Linear that follows it (Transformer2DModel / TransformerTemporalModel: norm -> proj_in,
// no activation between them).  With the per-(sample, channel) scale a and shift b of the statistics,
//     W . (a x + b) + bias  =  (W diag(a_s)) . x  +  (bias + W . b_s):
// one weight matrix and one bias vector PER SAMPLE (48 x 200 KB at level 0) and the GEMM reads the raw rows — the
// normalised tensor (one read + one write of the activation) never exists.  The mean term of b_s is formed with the
// ROUNDED fp16 weights the GEMM will use, W16 = fp16(W a):  bias_s = bias + W.beta - W16 . mean_s,  so that the GEMM
// result is exactly sum W16 (x - mean) + const: the rounding of W16 multiplies deviations, not a large mean.
// One wave per (sample, output row).
__global__ void gn_fold_kernel(const float* ab, const f16* beta, const f16* w, const f16* bias, int C, int N, int n_samples,
                               f16* w_out, float* bias_out) {
    const int lane = threadIdx.x & 63;
    const long long sn = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (sn >= (long long)n_samples * N) return;
    const int sample = (int)(sn / N), n = (int)(sn % N);
    const float* abs_ = ab + (size_t)sample * C * 2;
    const f16* wr = w + (size_t)n * C;
    f16* wo = w_out + ((size_t)sample * N + n) * C;
    float acc = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float a = abs_[2 * c], b = abs_[2 * c + 1], wv = (float)wr[c], be = (float)beta[c];
        const f16 w16 = (f16)(wv * a);
        wo[c] = w16;
        const float mean = a != 0.f ? (be - b) / a : 0.f;       // (a = rstd * gamma; gamma = 0: the channel contributes beta only)
        acc += wv * be - (float)w16 * mean;
    }
    acc = wave_sum(acc);
    if (lane == 0) bias_out[(size_t)sample * N + n] = acc + (bias ? (float)bias[n] : 0.f);
}

extern "C" int vdx_groupnorm_fold_linear_f16(const void* x, int C, int ldx, const void* gamma, const void* beta, float eps, int G,
                                             int n_samples, int rows_per_sample, void* workspace, int partition_samples,
                                             const void* w, const void* bias, int N, void* w_out, void* bias_out,
                                             vdx_stream_t stream) {
    VDX_CHECK(w && w_out && bias_out, "groupnorm_fold_linear: null pointer");
    VDX_CHECK(N > 0 && (long long)n_samples * N < (1ll << 31), "groupnorm_fold_linear: bad N");
    GnP p;
    if (const int rc = gn_run(x, C, ldx, nullptr, 0, 0, gamma, beta, eps, G, n_samples, rows_per_sample, 0, nullptr, 0, workspace,
                              partition_samples, true, p, stream))
        return rc;
    const long long waves = (long long)n_samples * N;
    hipLaunchKernelGGL(gn_fold_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p.ab, (const f16*)beta,
                       (const f16*)w, (const f16*)bias, C, N, n_samples, (f16*)w_out, (float*)bias_out);
    return vdx_launch_status("vdx_groupnorm_fold_linear_f16");
}

// ---- LayerNorm: one wave per row, row cached in registers ---------------------------------
template <int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const f16* x, int ldx, const f16* gamma, const f16* beta,
                                                         float eps, int M, int C, f16* y, int ldy) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nvec = C >> 3;
    const f16* xr = x + (size_t)row * ldx;
    f16x8 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int cv = lane + 64 * i;
        if (cv < nvec) {
            v[i] = *(const f16x8*)(xr + cv * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += (float)v[i][j];
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (lane + 64 * i < nvec) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = (float)v[i][j] - mean;
                q += d * d;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    f16* yr = y + (size_t)row * ldy;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int cv = lane + 64 * i;
        if (cv < nvec) {
            const f16x8 g = *(const f16x8*)(gamma + cv * 8);
            const f16x8 b = *(const f16x8*)(beta + cv * 8);
            f16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (f16)(((float)v[i][j] - mean) * rstd * (float)g[j] + (float)b[j]);
            *(f16x8*)(yr + cv * 8) = o;
        }
    }
}

// Row-group variant for widths C = LPR * NL * 8 (320 = 8 x 5, 640 = 16 x 5, 1280 = 32 x 5, 512 = 16 x 4, ...): LPR
// adjacent lanes share a row (lane sub-index s holds column vectors s, s + LPR, ...: every wave load covers 64 / LPR
// rows in pieces of LPR x 16 contiguous bytes), so ONE cross-lane reduction over LPR lanes (DPP adds up to 16 lanes)
// serves all the rows of the wave at once — the one-row-per-wave kernel spends 2 x 6 shuffles per row on 40 of its 64
// lanes at C = 320.  A wave walks row groups with a grid stride and loads the next group before it reduces the
// current one; gamma / beta stay in registers.  Two-pass variance (mean, then sum of squared deviations).
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
template <int LPR>
__device__ __forceinline__ float lanes_sum(float v) {
    v = dpp_add<0xB1>(v);                          // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);                          // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);                         // row_half_mirror: the other quad of the 8 (all four lanes hold its sum)
    if (LPR >= 16) v = dpp_add<0x140>(v);          // row_mirror: the other 8 of the 16
    if (LPR >= 32) v += __shfl_xor(v, 16, 64);
    if (LPR >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}

template <int LPR, int NL>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const f16* __restrict__ x, int ldx, const f16* __restrict__ gamma,
                                                              const f16* __restrict__ beta, float eps, int M,
                                                              f16* __restrict__ y, int ldy, int ngroups) {
    constexpr int RW = 64 / LPR;
    constexpr float inv_c = 1.0f / (float)(LPR * NL * 8);
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR, rw = lane / LPR;
    const int nwaves = gridDim.x * 4;
    int grp = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (grp >= ngroups) return;
    f16x8 g[NL], b[NL], v[NL], nv[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        g[j] = *(const f16x8*)(gamma + (sub + LPR * j) * 8);
        b[j] = *(const f16x8*)(beta + (sub + LPR * j) * 8);
    }
    auto fetch = [&](int gi, f16x8* dst) {
        const int row = gi * RW + rw;
        if (row < M) {
            const f16* xr = x + (size_t)row * ldx + sub * 8;
#pragma unroll
            for (int j = 0; j < NL; ++j) dst[j] = *(const f16x8*)(xr + LPR * 8 * j);
        } else {
#pragma unroll
            for (int j = 0; j < NL; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) dst[j][e] = (f16)0.f;
        }
    };
    fetch(grp, v);
    while (true) {
        const int nxt = grp + nwaves;
        if (nxt < ngroups) fetch(nxt, nv);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NL; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) s += (float)v[j][e];
        const float mean = lanes_sum<LPR>(s) * inv_c;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NL; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = (float)v[j][e] - mean;
                q += d * d;
            }
        const float rstd = rsqrtf(lanes_sum<LPR>(q) * inv_c + eps);
        const int row = grp * RW + rw;
        if (row < M) {
            f16* yr = y + (size_t)row * ldy + sub * 8;
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                // keep gamma / beta PACKED across the loop: without this the fp32 conversions are hoisted out of it
                // and the kernel needs 192 registers (two waves per SIMD)
                asm volatile("" : "+v"(g[j]), "+v"(b[j]));
                f16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (f16)(((float)v[j][e] - mean) * rstd * (float)g[j][e] + (float)b[j][e]);
                *(f16x8*)(yr + LPR * 8 * j) = o;
            }
        }
        if (nxt >= ngroups) break;
#pragma unroll
        for (int j = 0; j < NL; ++j) v[j] = nv[j];
        grp = nxt;
    }
}

template <int LPR>
static bool ln_rows_launch(int nl, dim3 grid, hipStream_t st, const f16* x, int ldx, const f16* gamma, const f16* beta, float eps,
                           int M, f16* y, int ldy, int ngroups) {
#define LNR(NL) hipLaunchKernelGGL((layernorm_rows_kernel<LPR, NL>), grid, dim3(256), 0, st, x, ldx, gamma, beta, eps, M, y, ldy, ngroups)
    switch (nl) {
        case 1: LNR(1); return true;
        case 2: LNR(2); return true;
        case 3: LNR(3); return true;
        case 4: LNR(4); return true;
        case 5: LNR(5); return true;
        default: return false;
    }
#undef LNR
}

extern "C" int vdx_layernorm_f16(const void* x, int ldx, const void* gamma, const void* beta, float eps,
                                 int M, int C, void* y, int ldy, vdx_stream_t stream) {
    VDX_CHECK(x && gamma && beta && y, "layernorm: null pointer");
    VDX_CHECK(M > 0 && C > 0 && C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0, "layernorm: bad shape M=%d C=%d", M, C);
    const int nv = (C / 8 + 63) / 64;
    VDX_CHECK(nv <= 4, "layernorm: C=%d too wide (max 2048)", C);
    hipStream_t st = (hipStream_t)stream;
    const int nvec = C / 8;
    for (int lpr = 8; lpr <= 64; lpr <<= 1) {             // C = lpr * nl * 8 with nl <= 5: the row-group kernel
        if (nvec % lpr || nvec / lpr > 5) continue;
        const int ngroups = (M + 64 / lpr - 1) / (64 / lpr);
        const int want = (ngroups + 3) / 4;
        dim3 rgrid(want < 2048 ? want : 2048);            // <= 8 blocks per CU; a wave then walks >= 1 row groups
        bool ok = false;
        const f16 *xx = (const f16*)x, *gg = (const f16*)gamma, *bb = (const f16*)beta;
        if (lpr == 8) ok = ln_rows_launch<8>(nvec / lpr, rgrid, st, xx, ldx, gg, bb, eps, M, (f16*)y, ldy, ngroups);
        else if (lpr == 16) ok = ln_rows_launch<16>(nvec / lpr, rgrid, st, xx, ldx, gg, bb, eps, M, (f16*)y, ldy, ngroups);
        else if (lpr == 32) ok = ln_rows_launch<32>(nvec / lpr, rgrid, st, xx, ldx, gg, bb, eps, M, (f16*)y, ldy, ngroups);
        else ok = ln_rows_launch<64>(nvec / lpr, rgrid, st, xx, ldx, gg, bb, eps, M, (f16*)y, ldy, ngroups);
        if (ok) return vdx_launch_status("vdx_layernorm_f16");
    }
    dim3 grid((M + 3) / 4), block(256);
#define LN_LAUNCH(NV) hipLaunchKernelGGL(layernorm_kernel<NV>, grid, block, 0, st, (const f16*)x, ldx, (const f16*)gamma, (const f16*)beta, eps, M, C, (f16*)y, ldy)
    switch (nv) {
        case 1: LN_LAUNCH(1); break;
        case 2: LN_LAUNCH(2); break;
        case 3: LN_LAUNCH(3); break;
        default: LN_LAUNCH(4); break;
    }
#undef LN_LAUNCH
    return vdx_launch_status("vdx_layernorm_f16");
}

// ---- row softmax, in place (AutoencoderKL mid-block attention: one head of 512 channels over h*w tokens; the
// [tokens][tokens] score matrix is a plain GEMM, this kernel, and a second GEMM).  One block per row, the row
// lives in registers (<= 8 vectors of 8 per thread = 16384 columns): 1 read + 1 write.  Scores are scaled in
// fp32 (diffusers: baddbmm alpha, upcast softmax), probabilities rounded to fp16 once.
template <int NV>
__global__ __launch_bounds__(256) void softmax_rows_kernel(f16* x, int ld, int cols, float c /* scale*log2(e) */) {
    __shared__ float red[8];
    const int tid = threadIdx.x, nvec = cols >> 3;
    f16* xr = x + (size_t)blockIdx.x * ld;
    f16x8 v[NV];
    float m = -3.0e38f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int cv = tid + 256 * i;
        if (cv < nvec) {
            v[i] = *(const f16x8*)(xr + cv * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) m = fmaxf(m, (float)v[i][j]);
        }
    }
    m = wave_max(m);
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    // scale > 0: max commutes with it
    float e[NV][8], sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (tid + 256 * i < nvec) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                e[i][j] = __builtin_amdgcn_exp2f(((float)v[i][j] - m) * c);
                sum += e[i][j];
            }
        }
    }
    sum = wave_sum(sum);
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int cv = tid + 256 * i;
        if (cv < nvec) {
            f16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (f16)(e[i][j] * inv);
            *(f16x8*)(xr + cv * 8) = o;
        }
    }
}

extern "C" int vdx_softmax_rows_f16(void* x, int ld, int rows, int cols, float scale, vdx_stream_t stream) {
    VDX_CHECK(x, "softmax_rows: null pointer");
    VDX_CHECK(rows > 0 && cols > 0 && cols % 8 == 0 && ld % 8 == 0 && ld >= cols, "softmax_rows: bad shape rows=%d cols=%d ld=%d", rows, cols, ld);
    VDX_CHECK(scale > 0.f, "softmax_rows: scale must be positive");
    const int nv = (cols / 8 + 255) / 256;
    VDX_CHECK(nv <= 8, "softmax_rows: %d columns exceed 16384", cols);
    hipStream_t st = (hipStream_t)stream;
    const float c = scale * 1.44269504088896341f;
#define SM_LAUNCH(NV) hipLaunchKernelGGL(softmax_rows_kernel<NV>, dim3(rows), dim3(256), 0, st, (f16*)x, ld, cols, c)
    if (nv <= 1) SM_LAUNCH(1);
    else if (nv <= 2) SM_LAUNCH(2);
    else if (nv <= 5) SM_LAUNCH(5);
    else SM_LAUNCH(8);
#undef SM_LAUNCH
    return vdx_launch_status("vdx_softmax_rows_f16");
}
