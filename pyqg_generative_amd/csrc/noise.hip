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

namespace qgx {

__device__ __forceinline__ void philox_round(uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3,
                                             uint32_t k0, uint32_t k1) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
}

__device__ __forceinline__ float unit_open(uint32_t u) {   // (0,1]
    return ((float)(u >> 8) + 1.0f) * 5.9604644775390625e-08f;
}

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
        uint32_t c0 = (uint32_t)quad, c1 = (uint32_t)step, c2 = (uint32_t)(member_offset + member),
                 c3 = (uint32_t)(step >> 32);
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            philox_round(c0, c1, c2, c3, k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        const float u0 = unit_open(c0), u1 = unit_open(c1), u2 = unit_open(c2), u3 = unit_open(c3);
        const float r0 = sqrtf(-2.0f * logf(u0)), r1 = sqrtf(-2.0f * logf(u2));
        float s0, cs0, s1, cs1;
        sincosf(6.283185307179586f * u1, &s0, &cs0);
        sincosf(6.283185307179586f * u3, &s1, &cs1);
        x[0] = (T)(r0 * cs0); x[1] = (T)(r0 * s0); x[2] = (T)(r1 * cs1); x[3] = (T)(r1 * s1);
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
