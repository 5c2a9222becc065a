// Developer tool: what does v_mfma_f32_32x32x16_f16 sustain on random data on this device with the
// operand traffic of k_convh (8 waves per workgroup, 1 workgroup per CU, 4 accumulators per wave)?
//   mode 0: operands in registers          mode 1: + 8 ds_read_b128 per tap (12 MFMAs), as the f16x3 loop
//   mode 2: as 1 with 8 MFMAs per tap (the plain-f16 loop)     mode 3: 16x16x32 shape, 16 accumulators, 8 reads / 24 MFMAs
//   mode 4: 32x32x16, 4x2 tiles per wave (8 accumulators), 12 reads / 24 MFMAs, 4 waves per workgroup
// hipcc --offload-arch=gfx950 -O3 mfma_peak_f16.hip -o /tmp/mfma_peak_f16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
template <int NWV>
__global__ __launch_bounds__(NWV * 64) void k4(float *out, int iters, unsigned long long *clk) {
    __shared__ h8 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += NWV * 64) {
        h8 v;
        for (int e = 0; e < 8; ++e) {
            unsigned x = ((i * 8 + e) * 2654435761u) ^ 0x9E3779B9u;
            v[e] = (_Float16)(((x >> 8) & 4095) * (1.f / 4096.f) - 0.5f);
        }
        lds[i] = v;
    }
    __syncthreads();
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    h8 P[4][2], W[2][2];
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 2; ++b) P[a][b] = lds[(threadIdx.x + 256 * (a * 2 + b)) & 4095];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) W[a][b] = lds[(threadIdx.x + 256 * (a * 2 + b) + 2048) & 4095];
    unsigned long long t0 = 0, r0 = 0;
    if (threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    const int li = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        h8 Pn[4][2], Wn[2][2];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) Pn[a][b] = lds[(li + it * 67 + 256 * (a * 2 + b)) & 4095];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) Wn[a][b] = lds[(li + it * 129 + 256 * (a * 2 + b) + 2048) & 4095];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                acc[m * 2 + n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[n][1], P[m][0], acc[m * 2 + n], 0, 0, 0);
                acc[m * 2 + n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[n][0], P[m][1], acc[m * 2 + n], 0, 0, 0);
                acc[m * 2 + n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[n][0], P[m][0], acc[m * 2 + n], 0, 0, 0);
            }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) P[a][b] = Pn[a][b];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) W[a][b] = Wn[a][b];
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = __builtin_amdgcn_s_memtime() - t0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
    float s = 0;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * NWV * 64 + threadIdx.x] = s;
}

__device__ int g_zero_pct = 0;   // percentage of exact zeros in the pixel-operand table (post-ReLU activations)
template <int mode>
__global__ __launch_bounds__(512) void k(float *out, int iters, unsigned long long *clk) {
    __shared__ h8 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 512) {
        h8 v;
        for (int e = 0; e < 8; ++e) {
            unsigned x = ((i * 8 + e) * 2654435761u) ^ 0x9E3779B9u;
            v[e] = (_Float16)(((x >> 8) & 4095) * (1.f / 4096.f) - 0.5f);
            if (i < 2048 && (int)((x >> 20) % 100) < g_zero_pct) v[e] = (_Float16)0.f;
        }
        lds[i] = v;
    }
    __syncthreads();
    f32x16 acc[4];
    f32x4 acc4[16];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) acc4[i][r] = 0.f;
    h8 P[2][2], W[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) { P[a][b] = lds[threadIdx.x + 512 * (a * 2 + b)]; W[a][b] = lds[(threadIdx.x + 512 * (a * 2 + b) + 2048) & 4095]; }
    unsigned long long t0 = 0, r0 = 0;
    if (threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    const int li = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        h8 Pn[2][2], Wn[2][2];
        if constexpr (mode != 0) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    Pn[a][b] = lds[(li + it * 67 + 512 * (a * 2 + b)) & 4095];
                    Wn[a][b] = lds[(li + it * 129 + 512 * (a * 2 + b) + 2048) & 4095];
                }
        } else {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) { Pn[a][b] = P[a][b]; Wn[a][b] = W[a][b]; }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (mode == 3) {
#pragma unroll
            for (int m = 0; m < 16; ++m) {   // 16 tiles of 16x16: (m & 1) picks P, (m >> 1) & 1 picks W fragments
                acc4[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W[(m >> 1) & 1][1], P[m & 1][0], acc4[m], 0, 0, 0);
                if (m < 8) acc4[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W[(m >> 1) & 1][0], P[m & 1][1], acc4[m], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    acc[m * 2 + n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[n][1], P[m][0], acc[m * 2 + n], 0, 0, 0);
                    acc[m * 2 + n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[n][0], P[m][1], acc[m * 2 + n], 0, 0, 0);
                    if constexpr (mode != 2) acc[m * 2 + n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[n][0], P[m][0], acc[m * 2 + n], 0, 0, 0);
                }
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) { P[a][b] = Pn[a][b]; W[a][b] = Wn[a][b]; }
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = __builtin_amdgcn_s_memtime() - t0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += acc4[i][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
    float *d; hipMalloc(&d, 1024 * 512 * 4);
    unsigned long long *clk; hipMalloc(&clk, 16);
    for (int zp = 0; zp <= 75; zp += 25)
    for (int mode = (zp ? 1 : 0); mode < (zp ? 2 : 6); ++mode) {
        hipMemcpyToSymbol(HIP_SYMBOL(g_zero_pct), &zp, sizeof(int));
        if (zp) printf("pixel operand with %d %% zeros: ", zp);
        const int grid = 256, iters = 20000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e9; unsigned long long h[2] = {0, 0};
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL((k<0>), dim3(grid), dim3(512), 0, 0, d, iters, clk);
            if (mode == 1) hipLaunchKernelGGL((k<1>), dim3(grid), dim3(512), 0, 0, d, iters, clk);
            if (mode == 2) hipLaunchKernelGGL((k<2>), dim3(grid), dim3(512), 0, 0, d, iters, clk);
            if (mode == 3) hipLaunchKernelGGL((k<3>), dim3(grid), dim3(512), 0, 0, d, iters, clk);
            if (mode == 4) hipLaunchKernelGGL((k4<4>), dim3(grid), dim3(256), 0, 0, d, iters, clk);
            if (mode == 5) hipLaunchKernelGGL((k4<4>), dim3(2 * grid), dim3(256), 0, 0, d, iters, clk);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) { best = ms; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost); }
        }
        const double mfmas = mode == 2 ? 8 : (mode == 3 || mode >= 4 ? 24 : 12);
        const double flop_per = mode == 3 ? 2.0 * 16 * 16 * 32 : 2.0 * 32 * 32 * 16;
        const double flop = (double)grid * (mode == 4 ? 4 : 8) * iters * mfmas * flop_per;
        const double clock_hz = h[1] ? (double)h[0] / (double)h[1] * 1e8 : 0.0;
        const double waves_per_simd = mode == 4 ? 1 : 2;
        const double cyc_per_mfma = best * 1e-3 * clock_hz / ((double)iters * mfmas * waves_per_simd);
        printf("mode %d: %.3f ms  %.0f TFLOP/s (f16 MFMA)  shader clock %.0f MHz  %.1f cycles per MFMA per SIMD\n", mode, best,
               flop / best / 1e9, h[1] ? (double)h[0] / (double)h[1] * 100.0 : 0.0, cyc_per_mfma);
    }
    return 0;
}
