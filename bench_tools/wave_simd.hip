// Which SIMD does each wave of a 512-thread workgroup land on?  (HW_REG_HW_ID: wave_id [3:0], simd_id [5:4], cu_id [11:8], ...)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(unsigned *out) {
    extern __shared__ char smem[];
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
    if (threadIdx.x == 9999) smem[0] = 1;
}
int main() {
    unsigned *d; hipMalloc(&d, 256 * 8 * 4);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 150000);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 150000, 0, d);
    unsigned h[256 * 8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < 256; ++b) {
        int cnt[4] = {0, 0, 0, 0}, cntB[4] = {0, 0, 0, 0};
        for (int w = 0; w < 8; ++w) { const int s = (h[b * 8 + w] >> 4) & 3; (w < 4 ? cnt : cntB)[s]++; }
        bool ok = true;
        for (int s = 0; s < 4; ++s) ok = ok && cnt[s] == 1 && cntB[s] == 1;
        if (!ok) ++bad;
        if (b < 6) { printf("workgroup %d: SIMD of waves 0..7:", b); for (int w = 0; w < 8; ++w) printf(" %d", (h[b * 8 + w] >> 4) & 3); printf("\n"); }
    }
    printf("workgroups whose waves 0-3 (and 4-7) are NOT one per SIMD: %d of 256\n", bad);
    return 0;
}
