// Microbenchmark: does the matrix pipe of a SIMD start slowly after it has been idle?  One wave per SIMD issues bursts of
// NB 32x32x16 f16 MFMAs (register operands) separated by an idle gap (s_sleep); the burst is timed with s_memtime.
//     hipcc -O3 --offload-arch=gfx950 bench_tools/mfma_ramp.hip -o bench_tools/_build/mfma_ramp && bench_tools/_build/mfma_ramp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int NB>
__global__ __launch_bounds__(256) void k_ramp(int gap, int iters, int allcu, float *sink, unsigned long long *times) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
    for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    h8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.01f * (lane + e)); b[e] = (_Float16)(0.02f * (lane - e)); }
    unsigned long long first = 0, rest = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k], 0, 0, 0);
        asm volatile("s_nop 0" :: "v"(acc[0][0]), "v"(acc[1][0]), "v"(acc[2][0]), "v"(acc[3][0]));
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int r = 0; r < (NB - 24) / 4; ++r)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k], 0, 0, 0);
        asm volatile("s_nop 0" :: "v"(acc[0][0]), "v"(acc[1][0]), "v"(acc[2][0]), "v"(acc[3][0]));
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        first += t1 - t0; rest += t2 - t1;
        for (int g = 0; g < gap; g += 64) __builtin_amdgcn_s_sleep(1);        // s_sleep 1 = 64 cycles
    }
    float s = 0.f;
    for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
    if (s == 12345.f) sink[0] = s;
    if (lane == 0 && blockIdx.x == 0 && threadIdx.x == 0) { times[0] = first; times[1] = rest; }
}

int main() {
    float *sink; unsigned long long *dt;
    CK(hipMalloc(&sink, 64)); CK(hipMalloc(&dt, 64));
    const int iters = 200;
    for (int allcu = 0; allcu < 2; ++allcu)
        for (int gap : {0, 128, 512, 1024, 2048, 4096, 16384}) {
            for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_ramp<120>, dim3(allcu ? 256 : 1), dim3(256), 0, 0, gap, iters, allcu, sink, dt);
            CK(hipDeviceSynchronize());
            unsigned long long t[2];
            CK(hipMemcpy(t, dt, sizeof(t), hipMemcpyDeviceToHost));
            printf("%s, idle gap %5d cycles: first 24 MFMAs of a burst %.1f cycles each, the following 96 %.1f cycles each\n",
                   allcu ? "256 workgroups" : "one workgroup ", gap, t[0] / (24.0 * iters), t[1] / (96.0 * iters));
        }
    return 0;
}
