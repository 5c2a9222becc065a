// Developer tool: the f32 MFMA rate this device actually sustains (bare v_mfma_f32_32x32x2_f32 loop,
// operands in registers, 4 independent accumulators per wave), to put the conv kernels' numbers in
// context.  Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0, float b0) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    // RANDOM operands (16 distinct registers per lane, uniform [-1,1)): trivial operands read high (DVFS)
    float av[16], bv[16];
    unsigned x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    for (int i = 0; i < 16; ++i) {
        x = x * 1664525u + 1013904223u; av[i] = (a0 != 0.f) ? ((x >> 8) * (2.0f / 16777216.0f) - 1.0f) : 0.5f;
        x = x * 1664525u + 1013904223u; bv[i] = (a0 != 0.f) ? ((x >> 8) * (2.0f / 16777216.0f) - 1.0f) : 0.25f;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u * 4 + i], bv[(u * 4 + i + 5) & 15], acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float *d; hipMalloc(&d, 4096 * 256 * 4);
    for (int mode = 0; mode < 2; ++mode)
    for (int wg_per_cu = 1; wg_per_cu <= 2; ++wg_per_cu) {
        const int grid = 256 * wg_per_cu, iters = 40000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, iters, mode ? 1.0f : 0.0f, 0.25f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)grid * 4 * iters * 16 * 4096.0;
            printf("%s wg/cu=%d rep=%d: %.3f ms  %.1f TFLOP/s\n", mode ? "random  " : "constant", wg_per_cu, rep, ms, flop / ms / 1e9);
        }
    }
    return 0;
}
