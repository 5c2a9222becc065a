// Generator layer 2 as the 1-D Winograd convolution with its input transform running under its matrix instructions
// (conv_wino2.hpp: two teams of position waves in ping-pong; bit-identical to conv_wino.hpp::k_convw).
//
// A translation unit of its own because it is built with -fno-slp-vectorize: the SLP vectoriser packs the transform's
// float32 adds and fused multiply-adds into v_pk_add_f32 / v_pk_fma_f32, and beside a wave that issues MFMAs on the same
// SIMD a packed-f32 instruction takes 29 cycles instead of 5 (bench_tools/coissue.hip, MI355X_MICROARCH.md
// "price of one filler beside MFMAs").  Replaces, for this one layer, the arithmetic of AndrewCNN.forward
// (cnn_tools.py:125-176, circular padding cnn_tools.py:79-98) exactly as conv_wino.hpp does.
#include "common.hpp"
#include <unordered_map>

namespace qgx {

#include "conv_types.hpp"
#include "conv_wino.hpp"        // ConvWArgs, mix_sum / mix_rest (k_convw itself is instantiated in conv.hip only)
#include "conv_wino2.hpp"

template <int NN, int TW, int R>
static int launch_convw2_n(const ConvWArgs &a, int total_tiles, hipStream_t st) {
    constexpr size_t lds = convw2_lds_bytes(NN, TW, R);
    static_assert(lds <= 160 * 1024 - 256, "k_convw2: LDS");
    void (*kern)(ConvWArgs, int) = k_convw2<NN, TW, R>;
    {   // the dynamic-LDS cap of a kernel is raised once per kernel, device and host thread (conv.hip::ensure_dynamic_lds)
        static thread_local int seen[16] = {0};
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (!seen[dev & 15]) {
            QGX_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            seen[dev & 15] = 1;
        }
    }
    const int grid = total_tiles < 256 ? total_tiles : 256;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a, total_tiles);
    QGX_HIP(hipGetLastError());
    return QGX_OK;
}

// the tile shapes this kernel is built for (those where it measured faster than k_convw: bench_tools/wino2_dev.hip);
// *done = false: not one of them, the caller takes k_convw
int launch_convw2(int N, int TW, int R, const ConvWArgs &a, int total_tiles, hipStream_t st, bool *done) {
    *done = true;
    if (N == 64 && TW == 64 && R == 8) return launch_convw2_n<64, 64, 8>(a, total_tiles, st);
    if (N == 96 && TW == 32 && R == 12) return launch_convw2_n<96, 32, 12>(a, total_tiles, st);
    *done = false;
    return QGX_OK;
}

}  // namespace qgx
