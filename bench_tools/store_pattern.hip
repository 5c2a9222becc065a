// Developer tool: HBM write rate of the conv epilogues' store pattern (16-byte pieces at a 512-byte pixel
// stride, the two halves of every 32-byte sector written by consecutive instructions) against fully
// contiguous wave stores (1 KB per instruction), 268 MB per launch like layer 1's output.
// hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o /tmp/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// mode 0: the epilogue pattern: a wave owns 32 pixels x 128 channels (512 B per pixel): 16 "octet pair" steps,
//         lane (li, h) writes hi then lo of octet 2 m + h of pixel li
// mode 1: the same bytes, each instruction 1 KB contiguous
template <int MODE>
__global__ void k(char *out, size_t npix) {
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
    const size_t nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    const u32x4 v = {(unsigned)wave, (unsigned)lane, 1u, 2u};
    for (size_t t = wave; t < npix / 32; t += nw) {
        char *base = out + t * 32 * 512;
        if (MODE == 0) {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                char *p = base + (size_t)li * 512 + (2 * m + h) * 32;
                *reinterpret_cast<u32x4 *>(p) = v;
                *reinterpret_cast<u32x4 *>(p + 16) = v;
            }
        } else {
#pragma unroll
            for (int m = 0; m < 16; ++m) *reinterpret_cast<u32x4 *>(base + (size_t)m * 1024 + lane * 16) = v;
        }
    }
}
int main() {
    const size_t npix = (size_t)128 * 64 * 64;
    char *d; hipMalloc(&d, npix * 512);
    for (int mode = 0; mode < 2; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(2048), dim3(256), 0, 0, d, npix);
            else hipLaunchKernelGGL(k<1>, dim3(2048), dim3(256), 0, 0, d, npix);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("mode %d: %.1f us  %.2f TB/s\n", mode, best * 1e3, npix * 512.0 / best / 1e9);
    }
    return 0;
}
