// elementwise.hip — layout edges of the UNet (conv_in / output permute), SiLU, and the
// orchestration ops the reference owns: ctx injection (fsdp_chunked_coherent.py:133-137),
// CFG combine + DDIM step (:141-142), linear-ramp blend (:204-217).
// The orchestration kernels round to fp16 after every tensor op, in the order torch evaluates
// the reference's expressions ON A GPU (the reference's tensors live on `cuda`): a 0-d fp32
// coefficient stays fp32 on either side of `*`, and `/ cpu_scalar` is `* (1/scalar)`.  ctx
// injection and blend are bit-exact against the CPU oracle; for the DDIM step torch-CPU differs
// from torch-GPU in exactly those two rules (oracle/ddim_ref.py `step_gpu_rules` restates them).
#include "vdx_common.h"

// fp32 -> fp16 as its own rounding step: the empty asm keeps hipcc from fusing the preceding fp32
// op and this conversion into v_fma_mixlo_f16 (one rounding), so results match torch's
// "compute in fp32, then cast" for fp16 tensors bit for bit.
__device__ __forceinline__ f16 rn16(float x) {
    asm volatile("" : "+v"(x));
    return (f16)x;
}

// ---- conv_in gather: (B,Cin,F,H,W) latent -> im2col rows [B*F*H*W][Kpad], K = (ky*3+kx)*Cin + ci
// (zero padded to Kpad, a multiple of 64) so conv_in runs on the MFMA GEMM like every other conv.
__global__ void im2col_in_kernel(const f16* x, f16* out, int B, int Cin, int F, int H, int W, int Kpad) {
    const int nvec = Kpad >> 3;
    const long long total = (long long)B * F * H * W * nvec;
    const int HW = H * W, K = 9 * Cin;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int kv = (int)(idx % nvec);
        const long long pix = idx / nvec;
        const int xx = (int)(pix % W), yy = (int)((pix / W) % H);
        const int f = (int)((pix / HW) % F), b = (int)(pix / ((long long)HW * F));
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = kv * 8 + j;
            const int tap = k / Cin, ci = k - tap * Cin;
            const int y = yy + tap / 3 - 1, xq = xx + tap % 3 - 1;
            const bool ok = k < K && (unsigned)y < (unsigned)H && (unsigned)xq < (unsigned)W;
            o[j] = ok ? x[(((size_t)b * Cin + ci) * F + f) * HW + y * W + xq] : (f16)0.f;
        }
        *(f16x8*)(out + (size_t)pix * Kpad + kv * 8) = o;
    }
}

extern "C" int vdx_im2col_in_f16(const void* x, void* out, int B, int Cin, int F, int H, int W, int Kpad,
                                 vdx_stream_t stream) {
    VDX_CHECK(x && out, "im2col_in: null pointer");
    VDX_CHECK(B > 0 && Cin > 0 && F > 0 && H > 0 && W > 0, "im2col_in: bad shape");
    VDX_CHECK(Kpad % 64 == 0 && Kpad >= 9 * Cin, "im2col_in: Kpad=%d must be a multiple of 64 >= 9*Cin", Kpad);
    const long long total = (long long)B * F * H * W * (Kpad / 8);
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(im2col_in_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f16*)x, (f16*)out,
                       B, Cin, F, H, W, Kpad);
    return vdx_launch_status("vdx_im2col_in_f16");
}

// ---- rows [B*F*HW][ld] -> (B,C,F,H,W) -------------------------------------------------------
__global__ void rows_to_ncfhw_kernel(const f16* rows, int ld, f16* out, int B, int C, int F, int HW) {
    const long long total = (long long)B * C * F * HW;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int p = (int)(idx % HW);
        const int f = (int)((idx / HW) % F);
        const int c = (int)((idx / ((long long)HW * F)) % C);
        const int b = (int)(idx / ((long long)HW * F * C));
        out[idx] = rows[(((size_t)b * F + f) * HW + p) * ld + c];
    }
}

extern "C" int vdx_rows_to_ncfhw_f16(const void* rows, int ld, void* out, int B, int C, int F, int H, int W,
                                     vdx_stream_t stream) {
    VDX_CHECK(rows && out && B > 0 && C > 0 && F > 0 && H > 0 && W > 0 && ld >= C, "rows_to_ncfhw: bad arguments");
    const long long total = (long long)B * C * F * H * W;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(rows_to_ncfhw_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f16*)rows, ld,
                       (f16*)out, B, C, F, H * W);
    return vdx_launch_status("vdx_rows_to_ncfhw_f16");
}

// ---- SiLU (time-embedding MLP) --------------------------------------------------------------
__global__ void silu_kernel(const f16* x, f16* y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = (f16)silu_f((float)x[i]);
}
extern "C" int vdx_silu_f16(const void* x, void* y, size_t n, vdx_stream_t stream) {
    VDX_CHECK(x && y && n > 0, "silu: bad arguments");
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(silu_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f16*)x, (f16*)y, n);
    return vdx_launch_status("vdx_silu_f16");
}

// ---- sinusoidal timestep embedding (diffusers `Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)`,
// SURVEY.md Appendix A.2): emb[b] = [cos(t * f_0..f_{h-1}) | sin(t * f_0..f_{h-1})], f_j = exp(-ln(10000) * j / h),
// h = dim / 2, computed in fp32 and stored fp16 like the reference's `t_emb.to(dtype)`.  The timestep is read from
// DEVICE memory, so a forward needs no host value of it (no sync on device-tensor timesteps, nothing to copy per step).
__global__ void timestep_embedding_kernel(const float* t, f16* out, int B, int dim) {
    const int half = dim >> 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * dim) return;
    const int b = i / dim, c = i - b * dim;
    const int j = c < half ? c : c - half;
    const float f = expf(-9.210340371976184f * (float)j / (float)half);
    const float a = t[0] * f;
    out[i] = (f16)(c < half ? cosf(a) : sinf(a));
}
extern "C" int vdx_timestep_embedding_f16(const float* t_device, void* out, int B, int dim, vdx_stream_t stream) {
    VDX_CHECK(t_device && out && B > 0 && dim > 0 && dim % 2 == 0, "timestep_embedding: bad arguments");
    const int n = B * dim;
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, t_device,
                       (f16*)out, B, dim);
    return vdx_launch_status("vdx_timestep_embedding_f16");
}

// ---- exact GELU (CLIP text tower MLP) ---------------------------------------------------------
__global__ void gelu_kernel(const f16* x, f16* y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = (f16)gelu_erf_f((float)x[i]);
}
extern "C" int vdx_gelu_f16(const void* x, void* y, size_t n, vdx_stream_t stream) {
    VDX_CHECK(x && y && n > 0, "gelu: bad arguments");
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(gelu_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f16*)x, (f16*)y, n);
    return vdx_launch_status("vdx_gelu_f16");
}

// ---- fsdp_chunked_coherent.py:133-137 -------------------------------------------------------
//   x = cat([lat]*2);  x = x + context_weight * ctx.repeat(1,1,F,1,1)
__global__ void cfg_input_kernel(const f16* lat, const f16* ctx, float weight, f16* x2, int C, int F, int HW) {
    const size_t n = (size_t)C * F * HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        f16 v = lat[i];
        if (ctx) {
            const int p = (int)(i % HW);
            const int c = (int)(i / ((size_t)HW * F));
            const f16 t = rn16(__fmul_rn((float)ctx[(size_t)c * HW + p], weight));  // fp16(cw * ctx)
            v = rn16(__fadd_rn((float)v, (float)t));                                 // fp16(x + t)
        }
        x2[i] = v;
        x2[n + i] = v;
    }
}
extern "C" int vdx_cfg_input_f16(const void* lat, const void* ctx, float weight, void* x2, int C, int F, int HW,
                                 vdx_stream_t stream) {
    VDX_CHECK(lat && x2 && C > 0 && F > 0 && HW > 0, "cfg_input: bad arguments");
    const size_t n = (size_t)C * F * HW;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(cfg_input_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f16*)lat,
                       (const f16*)ctx, weight, (f16*)x2, C, F, HW);
    return vdx_launch_status("vdx_cfg_input_f16");
}

// ---- fsdp_chunked_coherent.py:141-142 -------------------------------------------------------
//   u, c = noise.chunk(2);  lat = sched.step(u + gs*(c-u), t, lat).prev_sample     (DDIM, eta 0)
// DDIMScheduler.step with fp32 0-d coefficients and fp16 tensors: every tensor op rounds to fp16.
__global__ void cfg_ddim_kernel(const f16* eps2, const f16* lat, f16* out, float gs, float s1, float sa,
                                float sp, float s1p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float u = (float)eps2[i], c = (float)eps2[n + i], x = (float)lat[i];
        const f16 t1 = rn16(__fsub_rn(c, u));                 // c - u
        const f16 t2 = rn16(__fmul_rn(gs, (float)t1));        // gs * (c - u)
        const f16 g = rn16(__fadd_rn(u, (float)t2));          // u + gs*(c-u)
        const f16 a1 = rn16(__fmul_rn(s1, (float)g));         // sqrt(1-a_t) * eps
        const f16 a2 = rn16(__fsub_rn(x, (float)a1));         // sample - ...
        const f16 x0 = rn16(__fmul_rn((float)a2, sa));        // * (1/sqrt(a_t)): torch-GPU divides by a CPU scalar this way
        const f16 d = rn16(__fmul_rn(s1p, (float)g));         // sqrt(1-a_prev) * eps
        const f16 b1 = rn16(__fmul_rn(sp, (float)x0));        // sqrt(a_prev) * x0
        out[i] = rn16(__fadd_rn((float)b1, (float)d));
    }
}
extern "C" int vdx_cfg_ddim_step_f16(const void* eps2, const void* lat, void* lat_out, float guidance,
                                     float sqrt_one_minus_at, float sqrt_at, float sqrt_aprev,
                                     float sqrt_one_minus_aprev, size_t n, vdx_stream_t stream) {
    VDX_CHECK(eps2 && lat && lat_out && n > 0, "cfg_ddim_step: bad arguments");
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(cfg_ddim_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f16*)eps2,
                       (const f16*)lat, (f16*)lat_out, guidance, sqrt_one_minus_at, 1.0f / sqrt_at, sqrt_aprev,
                       sqrt_one_minus_aprev, n);
    return vdx_launch_status("vdx_cfg_ddim_step_f16");
}

// DDIMScheduler.step alone (the reference's `sched.step(eps, t, lat).prev_sample`, :142), same
// rounding chain; used when the CFG combine was done by the caller (unchanged reference script).
__global__ void ddim_kernel(const f16* eps, const f16* lat, f16* out, float s1, float sa, float sp, float s1p,
                            size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float g = (float)eps[i], x = (float)lat[i];
        const f16 a1 = rn16(__fmul_rn(s1, g));
        const f16 a2 = rn16(__fsub_rn(x, (float)a1));
        const f16 x0 = rn16(__fmul_rn((float)a2, sa));
        const f16 d = rn16(__fmul_rn(s1p, g));
        const f16 b1 = rn16(__fmul_rn(sp, (float)x0));
        out[i] = rn16(__fadd_rn((float)b1, (float)d));
    }
}
extern "C" int vdx_ddim_step_f16(const void* eps, const void* lat, void* lat_out, float sqrt_one_minus_at,
                                 float sqrt_at, float sqrt_aprev, float sqrt_one_minus_aprev, size_t n,
                                 vdx_stream_t stream) {
    VDX_CHECK(eps && lat && lat_out && n > 0, "ddim_step: bad arguments");
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(ddim_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f16*)eps,
                       (const f16*)lat, (f16*)lat_out, sqrt_one_minus_at, 1.0f / sqrt_at, sqrt_aprev,
                       sqrt_one_minus_aprev, n);
    return vdx_launch_status("vdx_ddim_step_f16");
}

// ---- fsdp_chunked_coherent.py:204-217 -------------------------------------------------------
//   full[:,:,s:e] += latc * w   (fp16 accumulator, fp32 product);  weight[:,:,s:e] += w
__global__ void blend_acc_kernel(f16* full, float* weight, const f16* chunk, const float* w, int C, int T, int HW,
                                 int s, int len) {
    const size_t n = (size_t)C * len * HW;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid < (size_t)len) weight[s + gid] = __fadd_rn(weight[s + gid], w[gid]);
    for (size_t i = gid; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int p = (int)(i % HW);
        const int f = (int)((i / HW) % len);
        const int c = (int)(i / ((size_t)HW * len));
        const size_t di = ((size_t)c * T + s + f) * HW + p;
        float prod = __fmul_rn((float)chunk[i], w[f]);
        asm volatile("" : "+v"(prod));   // keep the fp32 product a separate rounding (no fma_mix)
        full[di] = rn16(__fadd_rn((float)full[di], prod));
    }
}
extern "C" int vdx_blend_accumulate_f16(void* full, float* weight, const void* chunk, const float* w, int C, int T,
                                        int HW, int s, int e, vdx_stream_t stream) {
    VDX_CHECK(full && weight && chunk && w, "blend_accumulate: null pointer");
    VDX_CHECK(C > 0 && T > 0 && HW > 0 && 0 <= s && s < e && e <= T, "blend_accumulate: bad range [%d,%d) of %d", s, e, T);
    const int len = e - s;
    const size_t n = (size_t)C * len * HW;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(blend_acc_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (f16*)full, weight,
                       (const f16*)chunk, w, C, T, HW, s, len);
    return vdx_launch_status("vdx_blend_accumulate_f16");
}

//   lat = full / weight.clamp(min=1e-6)   -> fp32
__global__ void blend_fin_kernel(const f16* full, const float* weight, float* out, int C, int T, int HW) {
    const size_t n = (size_t)C * T * HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int t = (int)((i / HW) % T);
        out[i] = __fdiv_rn((float)full[i], fmaxf(weight[t], 1e-6f));
    }
}
extern "C" int vdx_blend_finalize_f32(const void* full, const float* weight, float* out, int C, int T, int HW,
                                      vdx_stream_t stream) {
    VDX_CHECK(full && weight && out && C > 0 && T > 0 && HW > 0, "blend_finalize: bad arguments");
    const size_t n = (size_t)C * T * HW;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(blend_fin_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f16*)full, weight,
                       out, C, T, HW);
    return vdx_launch_status("vdx_blend_finalize_f32");
}

// ---- decoded frames -> uint8 HWC  (fsdp_chunked_coherent.py:224-225):
//   img = (sample.permute(1,2,0) * 0.5 + 0.5).clamp(0, 1);  frame = (img * 255).byte()
// Every torch op rounds to fp16 (opmath fp32); .byte() truncates.  Input: the decoder's channels-last rows
// [pixels][ld] (first 3 columns = RGB), so the NCHW sample never has to exist for the video path.
__global__ void rows_to_u8_kernel(const f16* rows, int ld, size_t npix, unsigned char* out) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const f16* src = rows + p * ld;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const f16 a = rn16(__fmul_rn((float)src[c], 0.5f));
        f16 b = rn16(__fadd_rn((float)a, 0.5f));
        b = (f16)fminf(fmaxf((float)b, 0.f), 1.f);
        const f16 d = rn16(__fmul_rn((float)b, 255.f));
        out[p * 3 + c] = (unsigned char)(int)(float)d;
    }
}

extern "C" int vdx_rows_to_u8_frames(const void* rows, int ld, size_t n_pixels, void* out_u8, vdx_stream_t stream) {
    VDX_CHECK(rows && out_u8, "rows_to_u8_frames: null pointer");
    VDX_CHECK(ld >= 3 && n_pixels > 0, "rows_to_u8_frames: bad shape");
    const size_t blocks = (n_pixels + 255) / 256;
    VDX_CHECK(blocks < (1ull << 31), "rows_to_u8_frames: too many pixels");
    hipLaunchKernelGGL(rows_to_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       (const f16*)rows, ld, n_pixels, (unsigned char*)out_u8);
    return vdx_launch_status("vdx_rows_to_u8_frames");
}

// ---- box probes (bench.py's `box` object, the distributed rehearsal's occupancy hog): NOT on the denoising path ------------
// The boxes of a pool differ by several per cent as a whole (clock held under a dense MFMA stream, HBM speed).  A bench line
// that carries the rate of a FIXED instruction stream measured in the same process lets a reader tell a slower part from a
// slower build.  The probe is the stream every matrix kernel of this library is made of: back-to-back
// v_mfma_f32_32x32x16_f16 on eight accumulators per wave, two waves per SIMD, operands that differ per lane and rotate per
// issue (a constant-operand stream clocks ~40 % higher than real data: MI355X_MICROARCH.md, tools/micro/mfma_peak.hip).
__global__ __launch_bounds__(256) void mfma_probe_kernel(float* out, int iters) {
    f16x8 a[4], b[4];
    unsigned s = (threadIdx.x + 1u) * 2654435761u ^ (blockIdx.x * 40503u);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s = s * 1664525u + 1013904223u;
            a[r][e] = (f16)(((int)(s >> 16 & 1023) - 512) * (1.0f / 512.0f));
            s = s * 1664525u + 1013904223u;
            b[r][e] = (f16)(((int)(s >> 16 & 1023) - 512) * (1.0f / 8192.0f));
        }
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 3], b[(i + (i >> 2)) & 3], acc[i], 0, 0, 0);
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += acc[i][0] + acc[i][15];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = t;
}
// FLOPs of one launch: blocks x 4 waves x iters x 8 MFMAs x 2*32*32*16
extern "C" int vdx_probe_mfma_f16(float* out, size_t out_floats, int iters, double* flops, vdx_stream_t stream) {
    const int blocks = 2 * vdx_num_cus();
    VDX_CHECK(out && out_floats >= (size_t)blocks * 256 && iters > 0 && iters <= (1 << 22), "probe_mfma: needs %d floats of scratch, 0 < iters <= 4M", blocks * 256);
    if (flops) *flops = (double)blocks * 4.0 * iters * 8.0 * 2.0 * 32 * 32 * 16;
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters);
    return vdx_launch_status("vdx_probe_mfma_f16");
}

// Occupancy hog: `blocks` workgroups of 256 threads that each hold `lds_bytes` of a CU's LDS for `micros` microseconds and
// touch no memory — what a CU held by a collective's channel kernel looks like to an exact-fit persistent grid
// (`bench.py --rehearse-dist --hog R`: DESIGN §5).  The wait is on the constant 100 MHz wall clock with s_sleep between
// polls (no issue slots taken from a co-resident wave); every wave reaches the exit condition.
__global__ __launch_bounds__(256) void occupancy_hog_kernel(long long ticks) {
    extern __shared__ char hog_lds[];
    if (ticks < 0) hog_lds[threadIdx.x] = 1;          // keeps the allocation alive; never taken
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
extern "C" int vdx_probe_occupancy_hog(int blocks, int lds_bytes, int micros, vdx_stream_t stream) {
    VDX_CHECK(blocks > 0 && blocks <= 256 && lds_bytes >= 0 && lds_bytes <= 160 * 1024 && micros > 0 && micros <= 200000,
              "probe_occupancy_hog: blocks 1..256, lds 0..160 KB, 1..200000 us");
    static const hipError_t attr_rc = hipFuncSetAttribute((const void*)occupancy_hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr_rc != hipSuccess) return vdx_fail("probe_occupancy_hog: cannot reserve LDS");
    hipLaunchKernelGGL(occupancy_hog_kernel, dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream, (long long)micros * 100);
    return vdx_launch_status("vdx_probe_occupancy_hog");
}
