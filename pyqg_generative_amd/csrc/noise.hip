// On-device latent noise: z <- a z + b xi, xi ~ N(0,1) (Philox4x32-10 + Box-Muller).
//
// Replaces the host numpy draws of the reference
// (pyqg_generative/models/cgan_regression.py:154-155, cvae_regression.py:128-129,
// mean_var_model.py:102-103) and the AR1 / constant time samplers' arithmetic
// (pyqg_generative/tools/stochastic_pyqg.py:43-49, :62-71).  The reference stream is
// numpy's unseeded global MT19937, so there is no bit pattern to reproduce; the
// counter layout below is pinned by oracle/samplers_ref.py::philox_normal:
//   counter = (quad index, step lo, global member id, step hi), key = (seed lo, seed hi)
//   4 outputs -> 2 Box-Muller pairs -> normals at elements 4*quad .. 4*quad+3.
#include "common.hpp"
#include "philox.hpp"

namespace qgx {

template <typename T>
__global__ void k_noise(T *z, const T *xi_ext, int n_per_member, uint64_t seed, uint64_t member_offset,
                        uint64_t step, T a, T b) {
    const int quads = n_per_member / 4;
    const int member = blockIdx.y;
    const int quad = blockIdx.x * blockDim.x + threadIdx.x;
    if (quad >= quads) return;
    const size_t o = (size_t)member * n_per_member + 4 * (size_t)quad;
    T x[4];
    if (xi_ext) {
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = xi_ext[o + e];
    } else {
        float xn[4];
        philox_normal4(seed, member_offset + member, step, (uint32_t)quad, xn);
        x[0] = (T)xn[0]; x[1] = (T)xn[1]; x[2] = (T)xn[2]; x[3] = (T)xn[3];
    }
    if (a == (T)0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) z[o + e] = b * x[e];
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) z[o + e] = a * z[o + e] + b * x[e];
    }
}

int noise_update(void *z, const void *xi_ext, bool is_double, int B, int n_per_member, uint64_t seed,
                 uint64_t member_offset, uint64_t step, double a, double b, hipStream_t st) {
    QGX_REQUIRE(z && B > 0 && n_per_member > 0 && n_per_member % 4 == 0,
                "noise: n_per_member=%d must be a positive multiple of 4", n_per_member);
    const int quads = n_per_member / 4;
    dim3 grid((quads + 255) / 256, B), block(256);
    if (is_double)
        hipLaunchKernelGGL(k_noise<double>, grid, block, 0, st, (double *)z, (const double *)xi_ext,
                           n_per_member, seed, member_offset, step, a, b);
    else
        hipLaunchKernelGGL(k_noise<float>, grid, block, 0, st, (float *)z, (const float *)xi_ext,
                           n_per_member, seed, member_offset, step, (float)a, (float)b);
    QGX_HIP(hipGetLastError());
    return QGX_OK;
}

int noise_normal(void *z, bool is_double, int B, int n_per_member, uint64_t seed, uint64_t member_offset,
                 uint64_t step, double a, double b, hipStream_t st) {
    return noise_update(z, nullptr, is_double, B, n_per_member, seed, member_offset, step, a, b, st);
}

}  // namespace qgx

extern "C" int qgx_noise_normal(void *z_dev, int is_double, int B, int n_per_member, uint64_t seed,
                                uint64_t member_offset, uint64_t step, double a, double b, void *stream) {
    return qgx::noise_normal(z_dev, is_double != 0, B, n_per_member, seed, member_offset, step, a, b,
                             (hipStream_t)stream);
}
