// Developer tool: k_convw2 (conv_wino2.hpp, position teams in ping-pong) against k_convw (conv_wino.hpp) on the same random
// operands — bit identity of the output and kernel time (HIP events), per tile shape.  Stand-alone (no Python):
//     hipcc -O3 -std=c++17 --offload-arch=gfx950 bench_tools/wino2_dev.hip -o bench_tools/_build/wino2_dev
//     bench_tools/_build/wino2_dev [members]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
#include <random>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

namespace qgx {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
extern __shared__ __attribute__((aligned(16))) char conv_smem[];
__device__ __forceinline__ void range_guard(float mx, unsigned *range, unsigned bit) {
    if (mx > 65504.f) atomicOr(range, bit);
}
__device__ __forceinline__ unsigned pack_h2(float a, float b) {
    h2 v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(unsigned, v);
}
#include "../pyqg_generative_amd/csrc/conv_wino.hpp"
#include "../pyqg_generative_amd/csrc/conv_wino2.hpp"
}
using namespace qgx;

template <int NN, int TW, int R, int EXP = 0>
static void run_shape(int B, const std::vector<_Float16> &hin_all, const void *dw, const float *dbias, const float *dscale,
                      const float *dshift, unsigned *drange, int reps) {
    const size_t npix = (size_t)B * NN * NN;
    const size_t in_bytes = npix * 512, out_bytes = npix * 256;
    void *din, *dout1, *dout2;
    CK(hipMalloc(&din, in_bytes)); CK(hipMalloc(&dout1, out_bytes)); CK(hipMalloc(&dout2, out_bytes));
    CK(hipMemcpy(din, hin_all.data(), in_bytes, hipMemcpyHostToDevice));
    CK(hipMemset(dout1, 0xff, out_bytes)); CK(hipMemset(dout2, 0xee, out_bytes));
    ConvWArgs a = {};
    a.in = din; a.w = dw; a.bias = dbias; a.scale = dscale; a.shift = dshift;
    for (int p = 0; p < 8; ++p) a.pscale[p] = ldexpf(1.f, -14 - (p % 3));
    a.ascale = 1.f; a.range = drange; a.range_bit = 2;
    const int total_tiles = B * (NN / R) * (NN / TW);
    const int grid = total_tiles < 256 ? total_tiles : 256;
    const size_t lds1 = convw_lds_bytes(NN, TW, R), lds2 = convw2_lds_bytes(NN, TW, R);
    auto k1 = k_convw<NN, TW, R, 0, false>;
    auto k2 = k_convw2<NN, TW, R, EXP>;
    CK(hipFuncSetAttribute((const void *)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
    CK(hipFuncSetAttribute((const void *)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms1 = 0, ms2 = 0;
    for (int which = 0; which < 2; ++which) {
        a.out = which ? dout2 : dout1;
        for (int i = 0; i < 3; ++i) {
            if (which) hipLaunchKernelGGL(k2, dim3(grid), dim3(512), lds2, 0, a, total_tiles);
            else hipLaunchKernelGGL(k1, dim3(grid), dim3(512), lds1, 0, a, total_tiles);
        }
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) {
            if (which) hipLaunchKernelGGL(k2, dim3(grid), dim3(512), lds2, 0, a, total_tiles);
            else hipLaunchKernelGGL(k1, dim3(grid), dim3(512), lds1, 0, a, total_tiles);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        (which ? ms2 : ms1) = ms / reps;
    }
    std::vector<unsigned> o1(out_bytes / 4), o2(out_bytes / 4);
    CK(hipMemcpy(o1.data(), dout1, out_bytes, hipMemcpyDeviceToHost));
    CK(hipMemcpy(o2.data(), dout2, out_bytes, hipMemcpyDeviceToHost));
    size_t ndiff = 0, first = 0;
    for (size_t i = 0; i < o1.size(); ++i) if (o1[i] != o2[i]) { if (!ndiff) first = i; ++ndiff; }
    double sum = 0; for (size_t i = 0; i < o1.size(); i += 97) sum += (double)(o1[i] & 0xffff);
    printf("N=%d tile %dx%d B=%d tiles=%d: k_convw %.1f us, k_convw2 %.1f us (%.2fx); words differing %zu of %zu (first %zu) checksum %.0f\n",
           NN, R, TW, B, total_tiles, 1e3 * ms1, 1e3 * ms2, ms1 / ms2, ndiff, o1.size(), first, sum);
    fflush(stdout);
    CK(hipFree(din)); CK(hipFree(dout1)); CK(hipFree(dout2));
}

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 128;
    const int only = argc > 2 ? atoi(argv[2]) : 0;
    const int reps = 20;
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    // activations: ReLU output of layer 1 (half zeros), hi / lo split, [pixel][octet 16][hi 8 | lo 8]
    const size_t maxpix = (size_t)B * 64 * 64 > (size_t)32 * 128 * 128 ? (size_t)B * 64 * 64 : (size_t)32 * 128 * 128;
    std::vector<_Float16> hin(maxpix * 256);
    for (size_t px = 0; px < maxpix; ++px)
        for (int o = 0; o < 16; ++o)
            for (int e = 0; e < 8; ++e) {
                float x = nd(rng); x = x > 0.f ? 3.f * x : 0.f;
                const _Float16 hi = (_Float16)x;
                hin[(px * 16 + o) * 16 + e] = hi;
                hin[(px * 16 + o) * 16 + 8 + e] = (_Float16)(x - (float)hi);
            }
    // weights [chunk 8][ky 5][p 8][part 2][h 2][cout 64][8]
    const size_t nw = (size_t)8 * 5 * 8 * 2 * 2 * 64 * 8;
    std::vector<_Float16> hw(nw);
    for (size_t i = 0; i < nw; i += 1) {
        const int part = (i / (2 * 64 * 8)) & 1;
        const float x = nd(rng) * 3000.f;
        const _Float16 hi = (_Float16)x;
        hw[i] = part == 0 ? hi : (_Float16)((x - (float)hi) + nd(rng));
    }
    void *dw; CK(hipMalloc(&dw, nw * 2)); CK(hipMemcpy(dw, hw.data(), nw * 2, hipMemcpyHostToDevice));
    std::vector<float> bias(64), scale(64), shift(64);
    for (int i = 0; i < 64; ++i) { bias[i] = nd(rng); scale[i] = 1.f + 0.1f * nd(rng); shift[i] = 0.1f * nd(rng); }
    float *dbias, *dscale, *dshift; unsigned *drange;
    CK(hipMalloc(&dbias, 256)); CK(hipMalloc(&dscale, 256)); CK(hipMalloc(&dshift, 256)); CK(hipMalloc(&drange, 64));
    CK(hipMemcpy(dbias, bias.data(), 256, hipMemcpyHostToDevice));
    CK(hipMemcpy(dscale, scale.data(), 256, hipMemcpyHostToDevice));
    CK(hipMemcpy(dshift, shift.data(), 256, hipMemcpyHostToDevice));
    CK(hipMemset(drange, 0, 64));
    if (!only || only == 64) {
        run_shape<64, 64, 8>(B, hin, dw, dbias, dscale, dshift, drange, reps);
        run_shape<64, 64, 8>(3, hin, dw, dbias, dscale, dshift, drange, reps);      // one and two tiles per workgroup, ragged
        run_shape<64, 64, 8>(40, hin, dw, dbias, dscale, dshift, drange, reps);
        run_shape<64, 64, 4>(16, hin, dw, dbias, dscale, dshift, drange, reps);
    }
    if (!only || only == 96) {
        run_shape<96, 32, 12>(32, hin, dw, dbias, dscale, dshift, drange, reps);
        run_shape<96, 32, 16>(24, hin, dw, dbias, dscale, dshift, drange, reps);
    }
    if (!only || only == 32) {
        run_shape<32, 32, 16>(128, hin, dw, dbias, dscale, dshift, drange, reps);
        run_shape<32, 32, 8>(64, hin, dw, dbias, dscale, dshift, drange, reps);
    }
    if (!only || only == 48) run_shape<48, 16, 16>(64, hin, dw, dbias, dscale, dshift, drange, reps);
    if (!only || only == 128) {
        run_shape<128, 64, 8>(16, hin, dw, dbias, dscale, dshift, drange, reps);
        run_shape<128, 64, 4>(4, hin, dw, dbias, dscale, dshift, drange, reps);
    }
#if defined(QGX_W2_STAMPS) || defined(QGX_W2_EXPS)
    {   // one stamped launch of the headline shape: per phase (id, cycles since the previous stamp) of team A and team B
        // (-DQGX_W2_EXPS: the timing experiment alone, without the stamps' own cost)
#ifdef QGX_W2_STAMPS
        unsigned long long *dst; CK(hipMalloc(&dst, 16 * 512 * 8)); CK(hipMemset(dst, 0, 16 * 512 * 8));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_w2_stamps), &dst, sizeof(dst)));
#endif
        const int exp = argc > 3 ? atoi(argv[3]) : 0;
        if (exp == 1) run_shape<64, 64, 8, 1>(B, hin, dw, dbias, dscale, dshift, drange, 5);
        else if (exp == 2) run_shape<64, 64, 8, 2>(B, hin, dw, dbias, dscale, dshift, drange, 5);
        else if (exp == 3) run_shape<64, 64, 8, 3>(B, hin, dw, dbias, dscale, dshift, drange, 5);
        else if (exp == 4) run_shape<64, 64, 8, 4>(B, hin, dw, dbias, dscale, dshift, drange, 5);
        else if (exp == 5) run_shape<64, 64, 8, 5>(B, hin, dw, dbias, dscale, dshift, drange, 5);
        else if (exp == 6) run_shape<64, 64, 8, 6>(B, hin, dw, dbias, dscale, dshift, drange, 5);
        else if (exp == 7) run_shape<64, 64, 8, 7>(B, hin, dw, dbias, dscale, dshift, drange, 5);
        else if (exp == 8) run_shape<64, 64, 8, 8>(B, hin, dw, dbias, dscale, dshift, drange, 5);
        else if (exp == 9) run_shape<64, 64, 8, 9>(B, hin, dw, dbias, dscale, dshift, drange, 5);
        else if (exp == 10) run_shape<64, 64, 8, 10>(B, hin, dw, dbias, dscale, dshift, drange, 5);
        else if (exp == 11) run_shape<64, 64, 8, 11>(B, hin, dw, dbias, dscale, dshift, drange, 5);
        else if (exp == 12) run_shape<64, 64, 8, 12>(B, hin, dw, dbias, dscale, dshift, drange, 5);
        else if (exp == 13) run_shape<64, 64, 8, 13>(B, hin, dw, dbias, dscale, dshift, drange, 5);
        else if (exp == 14) run_shape<64, 64, 8, 14>(B, hin, dw, dbias, dscale, dshift, drange, 5);
        else if (exp == 15) run_shape<64, 64, 8, 15>(B, hin, dw, dbias, dscale, dshift, drange, 5);
        else run_shape<64, 64, 8>(B, hin, dw, dbias, dscale, dshift, drange, 5);
#ifdef QGX_W2_STAMPS
        std::vector<unsigned long long> st(16 * 512);
        CK(hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost));
        for (int wg = 0; wg < 2; ++wg)
            for (int team = 0; team < 2; ++team) {
                const unsigned long long *sp = st.data() + (wg * 2 + team) * 512;
                printf("workgroup %d team %c:", wg, 'A' + team);
                for (int i = 1; i < 512 && sp[i]; ++i) {
                    const unsigned long long t0 = sp[i - 1] & 0xffffffffffffffull, t1 = sp[i] & 0xffffffffffffffull;
                    printf(" %llu:%llu", sp[i] >> 56, t1 - t0);
                    if ((sp[i] >> 56) == 9) printf("\n    ");
                }
                printf("\n");
            }
#endif
    }
#endif
    unsigned flags; CK(hipMemcpy(&flags, drange, 4, hipMemcpyDeviceToHost));
    printf("range flags %u\n", flags);
    return 0;
}
