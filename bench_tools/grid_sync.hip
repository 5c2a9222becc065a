// Developer microbenchmark: cost of a cooperative-groups grid barrier on gfx950 (one workgroup per CU at most), with a
// cross-workgroup data exchange through global memory between barriers.
//   hipcc --offload-arch=gfx950 -O3 bench_tools/grid_sync.hip -o bench_tools/_build/grid_sync && bench_tools/_build/grid_sync
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;

__global__ void k_sync(float *buf, int iters, int n) {
    cg::grid_group grid = cg::this_grid();
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        // write my slot, read my neighbour's after the barrier
        for (int i = threadIdx.x; i < n; i += blockDim.x) buf[(size_t)blockIdx.x * n + i] = acc + it + i;
        grid.sync();
        const int nb = (blockIdx.x + 1) % gridDim.x;
        for (int i = threadIdx.x; i < n; i += blockDim.x) acc += buf[(size_t)nb * n + i];
        grid.sync();
    }
    if (acc == -1.f) buf[0] = acc;
}

int main() {
    float *buf;
    hipMalloc(&buf, 256 * 65536 * sizeof(float));
    for (int wgs : {16, 64, 128, 256}) {
        for (int n : {256, 16384}) {
            int iters = 200;
            void *args[] = {&buf, &iters, &n};
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchCooperativeKernel((void *)k_sync, dim3(wgs), dim3(256), args, 0, nullptr);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipError_t rc = hipLaunchCooperativeKernel((void *)k_sync, dim3(wgs), dim3(256), args, 0, nullptr);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("wgs %3d, %6d floats per workgroup: %.2f us per (write, barrier, read, barrier) [%s]\n", wgs, n, 1e3 * ms / iters, hipGetErrorString(rc));
        }
    }
    return 0;
}
