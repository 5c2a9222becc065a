// Vector types and the two helpers shared by the generator's translation units (conv.hip, conv_wino2.hip).
// Included inside namespace qgx.
#pragma once

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

extern __shared__ __attribute__((aligned(16))) char conv_smem[];

// f16x3 range guard.  A stored activation whose hi part leaves the f16 range (|x| > 65504) becomes inf and
// the next layer turns it into inf / NaN — or, behind a ReLU, into an innocent-looking zero.  Every epilogue
// that stores 16-bit activations passes the largest magnitude it stored through here; an overflow raises a
// sticky per-layer bit that qgx_generator_range_read reports (the facade re-runs in exact f32 / aborts).
__device__ __forceinline__ void range_guard(float mx, unsigned *range, unsigned bit) {
    if (mx > 65504.f) atomicOr(range, bit);
}

__device__ __forceinline__ unsigned pack_h2(float a, float b) {
    h2 v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(unsigned, v);
}
