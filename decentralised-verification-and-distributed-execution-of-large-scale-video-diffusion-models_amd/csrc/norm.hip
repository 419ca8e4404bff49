// norm.hip — GroupNorm (4-D and 5-D, optional 2-source concat, optional SiLU) and LayerNorm
// over channels-last fp16 rows.  HBM-bound streaming kernels:
//   GroupNorm = stats pass (1 read) + tiny finalize + apply pass (1 read + 1 write)
//   LayerNorm = 1 read + 1 write, one wave per row, row cached in registers.
// Thread mapping: a block owns a slab of rows of ONE sample; the block size is a multiple of
// the number of 8-channel vectors per row, so every thread keeps a fixed channel vector
// (coalesced 16-byte accesses, per-channel scale/shift held in registers).
#include "vdx_common.h"

// Rows of one sample per block: 128 for the big levels; halved until the grid has >= 1024 blocks so the
// deep levels (6912 rows in all at level 3) still spread over the 256 CUs instead of ~50-100 blocks.
static int gn_slab_rows(int n_samples, int rows_per_sample) {
    int rows = 128;
    while (rows > 8 && (long long)n_samples * ((rows_per_sample + rows - 1) / rows) < 1024) rows >>= 1;
    return rows;
}

struct GnP {
    const f16 *x, *x2;
    int c1, c2, ldx, ldx2;
    int C, G, cpg, nvec, krows;      // nvec = C/8, krows = rows processed in parallel per block
    int n_samples, rps, nslabs, slab_rows;   // rps = rows per sample
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
    int cnt = 0;
    for (int r = slab * p.slab_rows + rsub; r < r_end; r += p.krows, ++cnt) {
        const f16x8 v = gn_load(p, (size_t)sample * p.rps + r, cv);
        if (cnt == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) k[j] = (float)v[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float f = (float)v[j] - k[j];
            s[j] += f;
            ss[j] += f * f;
        }
    }
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
    double n = 0.0, mean = 0.0, m2 = 0.0;
    auto merge_d = [&](double nb, double mb, double m2b) {
        if (nb == 0.0) return;
        const double nn = n + nb, d = mb - mean;
        m2 += m2b + d * d * (n * nb / nn);
        mean += d * (nb / nn);
        n = nn;
    };
    for (int i = lane; i < p.nslabs; i += 64) {
        const float* src = p.partial + ((size_t)sample * p.nslabs + i) * 3 * p.G + 3 * g;
        merge_d((double)src[0], (double)src[1], (double)src[2]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {                      // fixed butterfly order: reproducible
        const double nb = __shfl_xor(n, o, 64), mb = __shfl_xor(mean, o, 64), m2b = __shfl_xor(m2, o, 64);
        // both partners must end with the same value: merge symmetrically (lower lane's triple first)
        const bool low = (lane & o) == 0;
        const double na = low ? n : nb, ma = low ? mean : mb, m2a = low ? m2 : m2b;
        const double nc = low ? nb : n, mc = low ? mb : mean, m2c = low ? m2b : m2;
        n = na; mean = ma; m2 = m2a;
        merge_d(nc, mc, m2c);
    }
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
    const int r_end = min(p.rps, (slab + 1) * p.slab_rows);
    for (int r = slab * p.slab_rows + rsub; r < r_end; r += p.krows) {
        const size_t row = (size_t)sample * p.rps + r;
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
    const int rows = gn_slab_rows(partition_samples > 0 ? partition_samples : n_samples, rows_per_sample);
    const size_t nslabs = (rows_per_sample + rows - 1) / rows;
    return ((size_t)n_samples * nslabs * G * 3 + (size_t)n_samples * C * 2) * sizeof(float);
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
extern "C" int vdx_groupnorm_part_f16(const void* x, int c1, int ldx, const void* x2, int c2, int ldx2,
                                      const void* gamma, const void* beta, float eps, int G,
                                      int n_samples, int rows_per_sample, int silu, void* y, int ldy,
                                      void* workspace, int partition_samples, vdx_stream_t stream) {
    VDX_CHECK(x && gamma && beta && y && workspace, "groupnorm: null pointer");
    const int C = c1 + c2;
    VDX_CHECK(c1 > 0 && c1 % 8 == 0 && c2 % 8 == 0, "groupnorm: c1=%d c2=%d must be multiples of 8", c1, c2);
    VDX_CHECK((c2 == 0) == (x2 == nullptr), "groupnorm: x2/c2 mismatch");
    VDX_CHECK(G > 0 && C % G == 0, "groupnorm: C=%d not divisible by G=%d", C, G);
    VDX_CHECK(ldx % 8 == 0 && ldy % 8 == 0 && (c2 == 0 || ldx2 % 8 == 0), "groupnorm: leading dims must be multiples of 8");
    VDX_CHECK(n_samples > 0 && rows_per_sample > 0, "groupnorm: empty input");
    VDX_CHECK(C / 8 <= 1024, "groupnorm: C=%d too wide", C);
    GnP p;
    p.x = (const f16*)x; p.x2 = (const f16*)x2; p.c1 = c1; p.c2 = c2; p.ldx = ldx; p.ldx2 = ldx2;
    p.C = C; p.G = G; p.cpg = C / G; p.nvec = C / 8;
    p.n_samples = n_samples; p.rps = rows_per_sample;
    p.slab_rows = gn_slab_rows(partition_samples > 0 ? partition_samples : n_samples, rows_per_sample);
    p.nslabs = (rows_per_sample + p.slab_rows - 1) / p.slab_rows;
    p.partial = (float*)workspace;
    p.ab = p.partial + (size_t)n_samples * p.nslabs * G * 3;
    const int nt = gn_threads(p.nvec, &p.krows);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(p.nslabs, n_samples);
    hipLaunchKernelGGL(gn_partial_kernel, grid, dim3(nt), (2 * (size_t)p.krows * C + p.krows + 2 * (size_t)G * p.krows) * sizeof(float), st, p);
    const int nsg = n_samples * G;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((nsg + 3) / 4), dim3(256), 0, st, p, (const f16*)gamma, (const f16*)beta, eps);
    if (silu)
        hipLaunchKernelGGL(gn_apply_kernel<true>, grid, dim3(nt), 0, st, p, (f16*)y, ldy);
    else
        hipLaunchKernelGGL(gn_apply_kernel<false>, grid, dim3(nt), 0, st, p, (f16*)y, ldy);
    return vdx_launch_status("vdx_groupnorm_f16");
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

// Packed variant for the level-0/1 widths (C = 320 / 640: 40 / 80 vectors of 8 per row): a wave takes
// R = 8 / 4 rows = 320 vectors = five fully used 1 KB wave loads instead of one row on 40 of its 64 lanes
// (the one-wave-per-row kernel reaches 3.7 TB/s at C = 320).  Flat vector v = 64*j + lane belongs to row v / nvec;
// the per-row sums are R wave reductions of masked partials — the same number of reductions per row as before.
template <int R>
__global__ __launch_bounds__(256) void layernorm_packed_kernel(const f16* x, int ldx, const f16* gamma, const f16* beta,
                                                                float eps, int M, int C, f16* y, int ldy) {
    constexpr int NL = 5;
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
    if (row0 >= M) return;
    const int nvec = C >> 3;           // R * nvec == 320
    f16x8 v[NL];
    int rr[NL], cc[NL];
    float s[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        const int flat = 64 * j + lane;
        rr[j] = flat / nvec;
        cc[j] = flat - rr[j] * nvec;
        const bool ok = row0 + rr[j] < M;
        s[j] = 0.f;
        if (ok) {
            v[j] = *(const f16x8*)(x + (size_t)(row0 + rr[j]) * ldx + cc[j] * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[j] += (float)v[j][e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[j][e] = (f16)0.f;
        }
    }
    float mean[NL], rstd[NL];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < NL; ++j) t += rr[j] == r ? s[j] : 0.f;
        const float m = wave_sum(t) / (float)C;
#pragma unroll
        for (int j = 0; j < NL; ++j) mean[j] = rr[j] == r ? m : mean[j];
    }
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = (float)v[j][e] - mean[j];
            q += d * d;
        }
        s[j] = q;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < NL; ++j) t += rr[j] == r ? s[j] : 0.f;
        const float rs = rsqrtf(wave_sum(t) / (float)C + eps);
#pragma unroll
        for (int j = 0; j < NL; ++j) rstd[j] = rr[j] == r ? rs : rstd[j];
    }
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        if (row0 + rr[j] < M) {
            const f16x8 g = *(const f16x8*)(gamma + cc[j] * 8);
            const f16x8 b = *(const f16x8*)(beta + cc[j] * 8);
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (f16)(((float)v[j][e] - mean[j]) * rstd[j] * (float)g[e] + (float)b[e]);
            *(f16x8*)(y + (size_t)(row0 + rr[j]) * ldy + cc[j] * 8) = o;
        }
    }
}

extern "C" int vdx_layernorm_f16(const void* x, int ldx, const void* gamma, const void* beta, float eps,
                                 int M, int C, void* y, int ldy, vdx_stream_t stream) {
    VDX_CHECK(x && gamma && beta && y, "layernorm: null pointer");
    VDX_CHECK(M > 0 && C > 0 && C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0, "layernorm: bad shape M=%d C=%d", M, C);
    const int nv = (C / 8 + 63) / 64;
    VDX_CHECK(nv <= 4, "layernorm: C=%d too wide (max 2048)", C);
    hipStream_t st = (hipStream_t)stream;
    if (C == 320 || C == 640) {       // packed: R rows per wave, 4 waves per block (measured: 3.75 -> 4.0 TB/s at 320; no gain at 1280)
        const int R = 2560 / C;
        dim3 pgrid((M + 4 * R - 1) / (4 * R)), pblock(256);
#define LNP_LAUNCH(RR) hipLaunchKernelGGL(layernorm_packed_kernel<RR>, pgrid, pblock, 0, st, (const f16*)x, ldx, (const f16*)gamma, (const f16*)beta, eps, M, C, (f16*)y, ldy)
        if (R == 8) LNP_LAUNCH(8);
        else LNP_LAUNCH(4);
#undef LNP_LAUNCH
        return vdx_launch_status("vdx_layernorm_f16");
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
