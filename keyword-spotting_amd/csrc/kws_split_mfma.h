// Device helpers shared by the matrix-core kernels (kws_dscnn.hip, kws_cnntrad.hip): the exact three-way bf16
// split of f32 operands, the 32x32x16 bf16 MFMA wrapper, accumulator row mapping, wavefront lane shifts.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kws {
namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));

// one v_max_f32 (the C++ forms compile to a canonicalising v_max plus the real one)
__device__ __forceinline__ float relu(float x) {
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}

// D layout of the 32x32 MFMAs: accumulator register r of a lane in half-wave `half` is output row
// (r & 3) + 8 (r >> 2) + 4 half; the column is lane & 31.
__device__ __forceinline__ int row_of(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }


// lane i <- lane i-1 / lane i+1 across the whole wavefront (0 shifted in at the ends)
__device__ __forceinline__ float from_lane_below(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138 /*wave_shr:1*/, 0xf, 0xf, true));
}
__device__ __forceinline__ float from_lane_above(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130 /*wave_shl:1*/, 0xf, 0xf, true));
}

// Exact three-way split of eight f32 values into bf16 pieces (y == hi + mid + lo, each piece the top 16 bits of
// the running remainder), packed as MFMA B operands.  bf16 x bf16 products are exact in the matrix core's f32
// accumulate, so the six products with combined order <= 2 reproduce the f32 product to ~2^-24 relative.
__device__ __forceinline__ uint32_t pack_top16(float even, float odd) {
    return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, odd), __builtin_bit_cast(uint32_t, even), 0x07060302u);
}
__device__ __forceinline__ float top16(float v) {
    return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, v) & 0xffff0000u);
}
__device__ __forceinline__ void split3(const float (&y)[8], uintx4& hi, uintx4& mid, uintx4& lo) {
    float r1[8], r2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        r1[j] = y[j] - top16(y[j]);
        r2[j] = r1[j] - top16(r1[j]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        hi[i] = pack_top16(y[2 * i], y[2 * i + 1]);
        mid[i] = pack_top16(r1[2 * i], r1[2 * i + 1]);
        lo[i] = pack_top16(r2[2 * i], r2[2 * i + 1]);
    }
}
__device__ __forceinline__ floatx16 mfma_bf16(const uintx4& a, const uintx4& b, floatx16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

}  // namespace
}  // namespace kws
