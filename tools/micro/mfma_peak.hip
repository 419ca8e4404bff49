// Dev tool: sustained MFMA rate of one MI355X for the two fp16 shapes the kernels use, 1 and 2 waves per SIMD, with the
// accumulators in arch VGPRs — the ceiling every "MFMA busy" figure of this repo has to be read against.
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_peak.hip -o tools/micro/mfma_peak ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters) {
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters) {
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
static void run(const char* name, K kern, int blocks, int iters, double flop_per_mfma, int nacc) {
    float* out;
    hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters / 10);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double fl = (double)blocks * 4 * iters * nacc * flop_per_mfma;
    printf("%-44s %4d blocks: %8.3f ms  %7.1f TFLOP/s\n", name, blocks, best, fl / best / 1e9);
    hipFree(out);
}

int main() {
    const int it = 20000;
    run("16x16x32 f16, 24 accumulators, 1 wave/SIMD", k16<24>, 256, it, 16.0 * 16 * 32 * 2, 24);
    run("16x16x32 f16, 24 accumulators, 2 waves/SIMD", k16<24>, 512, it, 16.0 * 16 * 32 * 2, 24);
    run("16x16x32 f16,  8 accumulators, 2 waves/SIMD", k16<8>, 512, it, 16.0 * 16 * 32 * 2, 8);
    run("32x32x16 f16,  8 accumulators, 1 wave/SIMD", k32<8>, 256, it, 32.0 * 32 * 16 * 2, 8);
    run("32x32x16 f16,  8 accumulators, 2 waves/SIMD", k32<8>, 512, it, 32.0 * 32 * 16 * 2, 8);
    run("32x32x16 f16,  4 accumulators, 2 waves/SIMD", k32<4>, 512, it, 32.0 * 32 * 16 * 2, 4);
    return 0;
}
