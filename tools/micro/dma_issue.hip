// Dev micro-benchmark (not part of the product): what does ISSUING an LDS-DMA instruction cost its wave?
// One or two waves per SIMD, each looping over batches of 8 staging instructions from an L2-resident buffer, s_memtime around
// the loop.  Forms: global_load_lds_dwordx4 (64-bit lane addresses), buffer_load_dwordx4 ... lds (32-bit lane offsets + a
// scalar descriptor), global_load_dwordx4 into registers, the same followed by ds_write_b128.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/dma_issue.hip -o tools/micro/dma_issue && tools/micro/dma_issue
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int FORM, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(const char* src, int iters, unsigned long long* cyc, float* out) {
    extern __shared__ __attribute__((aligned(128))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* dst = smem + wave * 8192;
    const char* my = src + ((size_t)blockIdx.x * WAVES + wave) * 8192 + lane * 16;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
    const int voff = (int)(((size_t)blockIdx.x * WAVES + wave) * 8192 + lane * 16);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (FORM == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                __builtin_amdgcn_global_load_lds((gptr_t)(my + i * 1024), (lptr_t)(dst + i * 1024), 16, 0, 0);
        } else if (FORM == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(dst + i * 1024), 16, voff, i * 1024, 0, 0);
        } else {
            f32x4 r[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) r[i] = *(const f32x4*)(my + i * 1024);
            if (FORM == 3) {
#pragma unroll
                for (int i = 0; i < 8; ++i) *(f32x4*)(dst + i * 1024 + lane * 16) = r[i];
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc += r[i];
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (lane == 0) cyc[blockIdx.x * WAVES + wave] = t1 - t0;
    out[blockIdx.x * WAVES * 64 + tid] = acc[0] + acc[1] + ((const float*)smem)[tid];
}

// the same with the wait deferred: issue 8, do not wait (the next batch's issue overlaps the previous batch's flight)
template <int FORM, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_nowait(const char* src, int iters, unsigned long long* cyc, float* out) {
    extern __shared__ __attribute__((aligned(128))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* dst = smem + wave * 8192;
    const char* my = src + ((size_t)blockIdx.x * WAVES + wave) * 8192 + lane * 16;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
    const int voff = (int)(((size_t)blockIdx.x * WAVES + wave) * 8192 + lane * 16);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (FORM == 0) __builtin_amdgcn_global_load_lds((gptr_t)(my + i * 1024), (lptr_t)(dst + i * 1024), 16, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(dst + i * 1024), 16, voff, i * 1024, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (lane == 0) cyc[blockIdx.x * WAVES + wave] = t1 - t0;
    out[blockIdx.x * WAVES * 64 + tid] = ((const float*)smem)[tid];
}

static const char* NAMES[] = {"global_load_lds_dwordx4", "buffer_load_dwordx4 lds", "global_load_dwordx4 -> VGPR", "global_load_dwordx4 + ds_write_b128"};

template <int FORM, int WAVES, bool NOWAIT>
static void run() {
    const int blocks = 256, iters = 2000;
    char* src;
    unsigned long long* cyc;
    float* out;
    hipMalloc(&src, (size_t)blocks * WAVES * 8192);
    hipMemset(src, 1, (size_t)blocks * WAVES * 8192);
    hipMalloc(&cyc, blocks * WAVES * 8);
    hipMalloc(&out, blocks * WAVES * 64 * 4);
    for (int r = 0; r < 2; ++r) {
        if (NOWAIT) hipLaunchKernelGGL((k_nowait<FORM, WAVES>), dim3(blocks), dim3(WAVES * 64), WAVES * 8192, 0, src, iters, cyc, out);
        else hipLaunchKernelGGL((k<FORM, WAVES>), dim3(blocks), dim3(WAVES * 64), WAVES * 8192, 0, src, iters, cyc, out);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(blocks * WAVES);
    hipMemcpy(h.data(), cyc, blocks * WAVES * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-40s %d wave(s)/SIMD %s: %7.1f cycles per instruction (median wave)\n", NAMES[FORM], WAVES / 4, NOWAIT ? "issue only (no wait in the loop)" : "batches of 8 + vmcnt(0)          ",
           (double)h[h.size() / 2] / (iters * 8.0));
    hipFree(src); hipFree(cyc); hipFree(out);
}

int main() {
    run<0, 4, false>(); run<1, 4, false>(); run<2, 4, false>(); run<3, 4, false>();
    run<0, 8, false>(); run<1, 8, false>(); run<2, 8, false>(); run<3, 8, false>();
    run<0, 4, true>(); run<1, 4, true>(); run<0, 8, true>(); run<1, 8, true>();
    return 0;
}
