// Philox4x32-10 + Box-Muller device helpers shared by noise.hip and the fused input-prep kernel.
// Counter layout pinned by oracle/samplers_ref.py::philox_normal:
//   counter = (quad index, step lo, global member id, step hi), key = (seed lo, seed hi)
//   4 outputs -> 2 Box-Muller pairs -> normals at elements 4*quad .. 4*quad+3.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

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

// four standard normals for (seed, global member, step, quad)
__device__ __forceinline__ void philox_normal4(uint64_t seed, uint64_t member, uint64_t step, uint32_t quad,
                                               float (&x)[4]) {
    uint32_t c0 = quad, c1 = (uint32_t)step, c2 = (uint32_t)member, c3 = (uint32_t)(step >> 32);
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
    x[0] = r0 * cs0; x[1] = r0 * s0; x[2] = r1 * cs1; x[3] = r1 * s1;
}

}  // namespace qgx
