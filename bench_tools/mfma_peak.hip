// Developer tool: what f32 MFMA rate and shader clock does this device sustain when the MFMA stream is
// accompanied by the operand traffic of the conv kernels?  v_mfma_f32_32x32x2_f32, 4 accumulators/wave.
//   mode 0: operands in registers (random values)         mode 1: + 4 ds_read_b128 per 16 MFMAs
//   mode 2: + 2 global_load_dwordx4 per 16 MFMAs (L2-resident 1 MB table)     mode 3: both
// hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int mode, int SCHED>
__global__ __launch_bounds__(256) void k(float *out, const float4 *tab, int iters, unsigned long long *clk) {
    __shared__ float4 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) {
        unsigned x = (i * 2654435761u) ^ 0x9E3779B9u;
        lds[i] = make_float4((x & 1023) * 1e-3f - 0.5f, ((x >> 10) & 1023) * 1e-3f - 0.5f, ((x >> 20) & 1023) * 1e-3f - 0.5f, 0.25f);
    }
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float4 A[2], Bv[2];
    A[0] = lds[threadIdx.x]; A[1] = lds[threadIdx.x + 256]; Bv[0] = lds[threadIdx.x + 512]; Bv[1] = lds[threadIdx.x + 768];
    unsigned long long t0 = 0, r0 = 0;
    if (threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    int li = threadIdx.x;
    const float4 *gp = tab + threadIdx.x;
    if constexpr (SCHED == 2) {
        float4 A1[2], B1[2];
        for (int it = 0; it < iters; it += 2) {
            // half-iteration 0: compute on (A,Bv), prefetch into (A1,B1)
            if constexpr (mode & 1) {
                A1[0] = lds[(li + it * 64) & 4095]; A1[1] = lds[(li + it * 64 + 1024) & 4095];
                B1[0] = lds[(li + it * 64 + 2048) & 4095]; B1[1] = lds[(li + it * 64 + 3072) & 4095];
            } else { A1[0] = A[0]; A1[1] = A[1]; B1[0] = Bv[0]; B1[1] = Bv[1]; }
            if constexpr (mode & 2) { B1[0] = gp[(it * 256) & 65535]; B1[1] = gp[(it * 256 + 128) & 65535]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[m * 2 + n] = __builtin_amdgcn_mfma_f32_32x32x2f32((&A[m].x)[e], (&Bv[n].x)[e], acc[m * 2 + n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // half-iteration 1: compute on (A1,B1), prefetch into (A,Bv)
            if constexpr (mode & 1) {
                A[0] = lds[(li + it * 64 + 64) & 4095]; A[1] = lds[(li + it * 64 + 1088) & 4095];
                Bv[0] = lds[(li + it * 64 + 2112) & 4095]; Bv[1] = lds[(li + it * 64 + 3136) & 4095];
            }
            if constexpr (mode & 2) { Bv[0] = gp[(it * 256 + 256) & 65535]; Bv[1] = gp[(it * 256 + 384) & 65535]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[m * 2 + n] = __builtin_amdgcn_mfma_f32_32x32x2f32((&A1[m].x)[e], (&B1[n].x)[e], acc[m * 2 + n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else
    for (int it = 0; it < iters; ++it) {
        float4 An[2] = {A[0], A[1]}, Bn[2] = {Bv[0], Bv[1]};
        if constexpr (mode & 1) {
            An[0] = lds[(li + it * 64) & 4095]; An[1] = lds[(li + it * 64 + 1024) & 4095];
            Bn[0] = lds[(li + it * 64 + 2048) & 4095]; Bn[1] = lds[(li + it * 64 + 3072) & 4095];
        }
        if constexpr (mode & 2) { Bn[0] = gp[(it * 256) & 65535]; Bn[1] = gp[(it * 256 + 128) & 65535]; }
        if constexpr (SCHED == 0) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[m * 2 + n] = __builtin_amdgcn_mfma_f32_32x32x2f32((&A[m].x)[e], (&Bv[n].x)[e], acc[m * 2 + n], 0, 0, 0);
        A[0] = An[0]; A[1] = An[1]; Bv[0] = Bn[0]; Bv[1] = Bn[1];
        if constexpr (SCHED == 1) {
            // interleave: one MFMA, then at most two non-MFMA instructions (DS read / VMEM read / VALU), 16 times
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100 | 0x020 | 0x002, 2, 0);
            }
        }
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = __builtin_amdgcn_s_memtime() - t0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float *d; hipMalloc(&d, 4096 * 256 * 4);
    float4 *tab; hipMalloc(&tab, 65536 * 16 + 4096 * 16); hipMemset(tab, 0x3c, 65536 * 16 + 4096 * 16);
    unsigned long long *clk; hipMalloc(&clk, 16);
    for (int sched = 0; sched < 3; sched += 2)
    for (int mode = 0; mode < 4; ++mode)
        for (int wg_per_cu = 1; wg_per_cu <= 2; ++wg_per_cu) {
            const int grid = 256 * wg_per_cu, iters = 40000;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            float best = 1e9; unsigned long long h[2] = {0, 0};
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
#define L_(M, S) hipLaunchKernelGGL((k<M, S>), dim3(grid), dim3(256), 0, 0, d, (const float4 *)tab, iters, clk)
                if (sched == 0) { if (mode == 0) L_(0, 0); if (mode == 1) L_(1, 0); if (mode == 2) L_(2, 0); if (mode == 3) L_(3, 0); }
                else            { if (mode == 0) L_(0, 2); if (mode == 1) L_(1, 2); if (mode == 2) L_(2, 2); if (mode == 3) L_(3, 2); }
#undef L_
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) { best = ms; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost); }
            }
            const double flop = (double)grid * 4 * iters * 16 * 4096.0;
            printf("sched %d mode %d wg/cu=%d: %.3f ms  %.1f TFLOP/s  shader clock %.0f MHz\n", sched, mode, wg_per_cu, best, flop / best / 1e9,
                   h[1] ? (double)h[0] / (double)h[1] * 100.0 : 0.0);
        }
    return 0;
}
