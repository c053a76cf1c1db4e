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

// ---- f16 pairs ---------------------------------------------------------------------------------------------------
// An f32 operand x, scaled by a power of two e into f16's range, as hi = f16(x e) and a second f16 piece for the residual
// x e - hi (exact in f32).  f16 x f16 products are exact in the matrix core's f32 accumulate and the matrix core honours
// f16 SUBNORMAL inputs (tools/mfma_f16_denorm_probe.hip), so hi*hi + hi*lo + lo*hi reproduces the f32 product to 2^-22
// relative or 2^-25 absolute (in scaled units), whichever is larger: three MFMAs per k-block instead of the six of the
// three-way bf16 split.  Two flavours of the residual piece: PLAIN lo = f16(x e - hi) goes to the same accumulator as
// hi*hi (kws_dscnn.hip); SCALED lo' = f16((x e - hi) 2^11) goes to a second accumulator (kws_cnntrad.hip).
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef _Float16 halfx2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ floatx16 mfma_f16(const uintx4& a, const uintx4& b, floatx16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(halfx8, a), __builtin_bit_cast(halfx8, b), c, 0, 0, 0);
}
// (y0, y1) * e -> one dword of hi pieces and one of PLAIN lo pieces; e is wavefront-uniform (a scalar register).  Six VALU
// instructions for two values: two multiplies by e (exact), v_cvt_pk_f16_f32 rounds both to the hi dword, v_fma_mix_f32 forms
// y e - hi with the f16 piece read as an operand, a second v_cvt_pk_f16_f32 packs the residuals (see split_pair8).
__device__ __forceinline__ void split_pair2(float y0, float y1, float e, uint32_t& hi, uint32_t& lo) {
    float t0, t1;
    uint32_t h, l;
    asm("v_mul_f32 %2, %4, %6\n\t"
        "v_mul_f32 %3, %5, %6\n\t"
        "v_cvt_pk_f16_f32 %0, %2, %3\n\t"
        "v_fma_mix_f32 %2, %0, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %3, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_cvt_pk_f16_f32 %1, %2, %3\n\t"
        "s_nop 1"  // (rule R6, see split_pair8)
        : "=&v"(h), "=&v"(l), "=&v"(t0), "=&v"(t1)
        : "v"(y0), "v"(y1), "s"(e));
    hi = h;
    lo = l;
}
// Eight values at once, as ONE asm block (four separate blocks get an s_nop from the compiler between each other): the four
// dwords' instructions are interleaved, so an instruction's source was written four instructions earlier.
// v_cvt_pk_f16_f32 (gfx950) rounds two values in one full-rate instruction; v_fma_mixlo/mixhi_f16, which this used until late
// in round 3, issue at the transcendental rate (tools/valu_rate.hip: 6.4 SIMD cycles each with two wavefronts on the SIMD
// against 3.4 for v_cvt_pk_f16_f32 and v_fma_mix_f32, 2.3 for v_mul_f32): 72 instead of 129 SIMD cycles per eight values, same
// bits (x * e is exact, both forms round to nearest even once).
#define KWS_SPLIT8_TAIL                                                                                     \
        "v_cvt_pk_f16_f32 %0, %8, %9\n\t"                                                                   \
        "v_cvt_pk_f16_f32 %1, %10, %11\n\t"                                                                 \
        "v_cvt_pk_f16_f32 %2, %12, %13\n\t"                                                                 \
        "v_cvt_pk_f16_f32 %3, %14, %15\n\t"                                                                 \
        "v_fma_mix_f32 %8, %0, -1.0, %8 op_sel_hi:[1,0,0]\n\t"                                              \
        "v_fma_mix_f32 %10, %1, -1.0, %10 op_sel_hi:[1,0,0]\n\t"                                            \
        "v_fma_mix_f32 %12, %2, -1.0, %12 op_sel_hi:[1,0,0]\n\t"                                            \
        "v_fma_mix_f32 %14, %3, -1.0, %14 op_sel_hi:[1,0,0]\n\t"                                            \
        "v_fma_mix_f32 %9, %0, -1.0, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"                               \
        "v_fma_mix_f32 %11, %1, -1.0, %11 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"                             \
        "v_fma_mix_f32 %13, %2, -1.0, %13 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"                             \
        "v_fma_mix_f32 %15, %3, -1.0, %15 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"                             \
        "v_cvt_pk_f16_f32 %4, %8, %9\n\t"                                                                   \
        "v_cvt_pk_f16_f32 %5, %10, %11\n\t"                                                                 \
        "v_cvt_pk_f16_f32 %6, %12, %13\n\t"                                                                 \
        "v_cvt_pk_f16_f32 %7, %14, %15"
#ifndef KWS_X_NO_SPLIT_NOP  // (the switch exists for tests/test_isa_hazards.py: without the wait the lint must report rule R6)
#define KWS_SPLIT8_NOP "\n\ts_nop 1"  // a just-written VGPR needs two wait states before a matrix instruction reads it as an
                                      // operand; the compiler pads only one after an asm block
#else
#define KWS_SPLIT8_NOP ""
#endif
__device__ __forceinline__ void split_pair8(const float (&y)[8], float e, uintx4& hi, uintx4& lo) {
    uint32_t h0, h1, h2, h3, l0, l1, l2, l3;
    float t0, t1, t2, t3, t4, t5, t6, t7;
    asm("v_mul_f32 %8, %16, %24\n\t"
        "v_mul_f32 %9, %17, %24\n\t"
        "v_mul_f32 %10, %18, %24\n\t"
        "v_mul_f32 %11, %19, %24\n\t"
        "v_mul_f32 %12, %20, %24\n\t"
        "v_mul_f32 %13, %21, %24\n\t"
        "v_mul_f32 %14, %22, %24\n\t"
        "v_mul_f32 %15, %23, %24\n\t"
        KWS_SPLIT8_TAIL KWS_SPLIT8_NOP
        : "=&v"(h0), "=&v"(h1), "=&v"(h2), "=&v"(h3), "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3), "=&v"(t0), "=&v"(t1), "=&v"(t2),
          "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
        : "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]), "v"(y[7]), "s"(e));
    hi[0] = h0; hi[1] = h1; hi[2] = h2; hi[3] = h3;
    lo[0] = l0; lo[1] = l1; lo[2] = l2; lo[3] = l3;
}
// The same for values that arrive already scaled (the DS-CNN folds a block's operand scale into its depthwise table): the
// residuals overwrite the values.
__device__ __forceinline__ void split_pair8_scaled(float (&y)[8], uintx4& hi, uintx4& lo) {
    uint32_t h0, h1, h2, h3, l0, l1, l2, l3;
    asm(KWS_SPLIT8_TAIL KWS_SPLIT8_NOP
        : "=&v"(h0), "=&v"(h1), "=&v"(h2), "=&v"(h3), "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3), "+v"(y[0]), "+v"(y[1]), "+v"(y[2]),
          "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7]));
    hi[0] = h0; hi[1] = h1; hi[2] = h2; hi[3] = h3;
    lo[0] = l0; lo[1] = l1; lo[2] = l2; lo[3] = l3;
}
// the power of two s with bound * s < 2^15 (bound >= 0), as an exponent kept within +-100 (so that 2^-k is a normal float too)
__device__ __forceinline__ int pow2_exp_for(float bound) {
    const int e = (int)((__builtin_bit_cast(uint32_t, bound) >> 23) & 0xffu);  // bound < 2^(e - 126)
    const int k = 141 - e;
    return k < -100 ? -100 : (k > 100 ? 100 : k);
}
__device__ __forceinline__ float pow2f(int k) {
    k = k < -126 ? -126 : (k > 127 ? 127 : k);
    return __builtin_bit_cast(float, (uint32_t)(k + 127) << 23);
}

}  // namespace
}  // namespace kws
