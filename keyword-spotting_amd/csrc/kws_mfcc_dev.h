// Device-side building blocks of the MFCC front end for gfx950: the packed 512-point transform, the spectrum split,
// the sparse mel stage and the log / DCT tail of one frame pair (mfcc_pair), shared by the batched kernels
// (kws_mfcc.hip) and by the streaming push, whose one-frame front end runs inside the DS-CNN kernel's prologue
// (kws_dscnn.hip).  Everything lives in an anonymous namespace: each translation unit gets its own inlined copy.
// See kws_mfcc.hip for the work decomposition and the numerics.
#pragma once
#include "kws_internal.h"
#include "kws_mfcc_f64_dev.h"

namespace kws {
namespace {

// 8-byte aligned, so a complex value moves with one ds_read_b64 / ds_write_b64 (an unaligned pair becomes
// ds_read2_b32: twice the LDS cycles and the 32-bank conflict rules).
// A 2-vector, so a complex value lives in an aligned register pair: the compiler then maps complex adds, the
// rotations by -i / (1 -+ i)/sqrt2 and the twiddle products onto packed instructions (v_pk_add_f32 / v_pk_mul_f32 /
// v_pk_fma_f32 with op_sel / neg modifiers for the swaps and signs) without moves to build the pairs; with a scalar
// struct the vectoriser paired components of different values and a sixth of the kernel's VALU instructions were
// v_mov_b32.
typedef float cf __attribute__((ext_vector_type(2)));

__device__ __forceinline__ cf cadd(cf a, cf b) { return a + b; }
__device__ __forceinline__ cf csub(cf a, cf b) { return a - b; }
// (fma(a.x, b.x, -(a.y*b.y)), fma(a.x, b.y, a.y*b.x)) in two packed instructions: the component swaps are op_sel
// operands (op_sel picks the source half of the low lane, op_sel_hi of the high lane) and the one-sided negation is
// neg_lo -- the compiler only folds whole-vector negations and builds (-b.y, b.x) with an xor and a move instead.
__device__ __forceinline__ cf cmul(cf a, cf b) {
    cf t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(a), "v"(b));  // (-a.y*b.y, a.y*b.x)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "v"(b), "v"(t));            // a.x*(b.x, b.y) + t
    return r;
}
// a + (-i) b = (a.x + b.y, a.y - b.x) and a - (-i) b = (a.x - b.y, a.y + b.x): the rotation by -i is the operand
// swap, one packed instruction each (same reason as cmul)
__device__ __forceinline__ cf add_mi(cf a, cf b) {
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ cf sub_mi(cf a, cf b) {
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// In-place 8-point forward DFT, natural order in and out: v[k] = sum_n v[n] * exp(-2*pi*i*n*k/8).
__device__ __forceinline__ void dft8(cf (&v)[8]) {
    constexpr float R = 0.70710678118654752440f;
    const cf b0 = cadd(v[0], v[4]), b4 = csub(v[0], v[4]);
    const cf b1 = cadd(v[1], v[5]), c5 = csub(v[1], v[5]);
    const cf b2 = cadd(v[2], v[6]), b6 = csub(v[2], v[6]);
    const cf b3 = cadd(v[3], v[7]), c7 = csub(v[3], v[7]);
    // odd branch pre-twiddles W8^n (the one of b6, -i, is folded into its uses)
    const cf b5 = add_mi(c5, c5) * R;     // * (1 - i)/sqrt2:  ((x + y) R, (y - x) R)
    const cf b7 = sub_mi(c7, c7) * -R;    // * (-1 - i)/sqrt2: ((y - x) R, -(x + y) R)
    // even outputs: 4-point DFT of b0..b3
    const cf d0 = cadd(b0, b2), d1 = csub(b0, b2), d2 = cadd(b1, b3), d3 = csub(b1, b3);
    v[0] = cadd(d0, d2);
    v[4] = csub(d0, d2);
    v[2] = add_mi(d1, d3);
    v[6] = sub_mi(d1, d3);
    // odd outputs: 4-point DFT of b4, -i b6, b5, b7
    const cf e0 = add_mi(b4, b6), e1 = sub_mi(b4, b6), e2 = cadd(b5, b7), e3 = csub(b5, b7);
    v[1] = cadd(e0, e2);
    v[5] = csub(e0, e2);
    v[3] = add_mi(e1, e3);
    v[7] = sub_mi(e1, e3);
}

// Complex row stride of the two register<->LDS exchanges.  72 (64 + 8) makes the strided column gather of the
// first exchange conflict-free for ds_read_b64; in the second exchange the column index is additionally XORed
// with 2*(k1 & 3), which makes its ds_write_b64 conflict-free while the gather stays four aligned ds_read_b128
// (LDS-array cycles per tools/lds_model.py: 48 for the exchange, the minimum).
constexpr int XROW = 72;

// Per-wavefront LDS scratch (bytes): exchange / spectrum / power buffer, log-mel vectors.
constexpr int SCR_XBUF = 0, SCR_PBUF = 0, SCR_LBUF = 4608, SCR_BYTES = 4608 + 512;
static_assert(64 * MEL_STRIDE * 8 <= SCR_BYTES, "the power buffer (64 chunks of MEL_STRIDE float2) may run over the log-mel vectors, which are written after its last read, but not out of the wavefront's scratch");
static_assert(KWS_MFCC_WAVES * SCR_BYTES >= NFFT * 4, "frame loads may run up to NFFT floats past the staged span, into the scratch");

typedef float floatx2 __attribute__((ext_vector_type(2)));
// LDS byte address of a pointer into shared memory (for hand-written ds_* instructions)
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

// LDS instructions of one wavefront execute in order, so data written by one lane is visible to a later
// read of another lane of the SAME wavefront without s_barrier or s_waitcnt; only the compiler has to keep
// the order.
__device__ __forceinline__ void wave_lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Spectrum buffer index swizzle: lane (k1, q) stores bin k1 + 8q + 64d; without the XOR sixteen lanes of a
// ds_write_b64 group hit four bank pairs (4-way conflict).
__device__ __forceinline__ int zswz(int k) { return k ^ ((k >> 3) & 7); }

// 512-point complex FFT across one wavefront.
//   in : lane l holds z[64*n1 + l] in v[n1]
//   out: lane l (k1 = l>>3, c = l&7) holds Z[k1 + 8*c + 64*d] in v[d]
// xbuf: 8*XROW complex of LDS private to the wavefront; t1[i] = W512^(lane*i); tw2 = LDS table [8][8] of
// W64^(q*i).
__device__ __forceinline__ void fft512(cf (&v)[8], cf* xbuf, const cf (&t1)[8], const cf* tw2, int lane) {
    const int k1 = lane >> 3, q = lane & 7;
    dft8(v);  // over n1 -> k1 (register index)
#pragma unroll
    for (int i = 1; i < 8; ++i) v[i] = cmul(v[i], t1[i]);  // W512^(lane*k1)
#ifdef KWS_MFCC_XCHG_PERMLANE
    // First exchange in registers -- an EXPERIMENT, not built by default (profiles/r02_mfcc_exchange_experiment.txt: LDS
    // instructions -15 %, LDS-array cycles -10 %, LDS issue stalls -39 %, VALU instructions +10 %, kernel time +1 %: the
    // kernel is bound by the SUM of its VALU and LDS time per wavefront, not by LDS alone).  Register index k1 (3 bits) <-> lane bits [5:3], the low
    // lane bits stay.  Three swap stages: lane bit 5 <-> register bit 2 with v_permlane32_swap (upper half of v[i] <->
    // lower half of v[i+4]), lane bit 4 <-> register bit 1 with v_permlane16_swap (odd 16-lane rows of v[i] <-> even rows
    // of v[i+2]), lane bit 3 <-> register bit 0 through row_ror:8 (= lane ^ 8 inside a 16-lane row) fused into the selects.
    {
        uint32_t r[8][2];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            r[i][0] = __builtin_bit_cast(uint32_t, (float)v[i].x);
            r[i][1] = __builtin_bit_cast(uint32_t, (float)v[i].y);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(r[i][h]), "+v"(r[i + 4][h]));
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (!(i & 2))
#pragma unroll
                for (int h = 0; h < 2; ++h) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(r[i][h]), "+v"(r[i + 2][h]));
        const bool up = (lane & 8) != 0;  // this lane keeps r[odd] and receives into r[even]
#pragma unroll
        for (int i = 0; i < 8; i += 2)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t send = up ? r[i][h] : r[i + 1][h];
                const uint32_t got = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)send, 0x128 /*row_ror:8*/, 0xf, 0xf, false);
                r[i][h] = up ? got : r[i][h];
                r[i + 1][h] = up ? r[i + 1][h] : got;
            }
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = cf{__builtin_bit_cast(float, r[i][0]), __builtin_bit_cast(float, r[i][1])};
    }
#else
#pragma unroll
    for (int i = 0; i < 8; ++i) xbuf[i * XROW + lane] = v[i];
    wave_lds_order();
    // lane (k1, b=q): gather y[k1][8a + b], a = 0..7.  Eight ds_read_b64 by hand: the compiler would pair them
    // into ds_read2_b64, which moves the same bytes in twice the LDS-array cycles.
    {
        floatx2 r0, r1, r2, r3, r4, r5, r6, r7;
        asm volatile(
            "ds_read_b64 %0, %8\n\t"
            "ds_read_b64 %1, %8 offset:64\n\t"
            "ds_read_b64 %2, %8 offset:128\n\t"
            "ds_read_b64 %3, %8 offset:192\n\t"
            "ds_read_b64 %4, %8 offset:256\n\t"
            "ds_read_b64 %5, %8 offset:320\n\t"
            "ds_read_b64 %6, %8 offset:384\n\t"
            "ds_read_b64 %7, %8 offset:448\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
            : "v"(lds_addr(xbuf + k1 * XROW + q))
            : "memory");
        v[0] = {r0.x, r0.y}; v[1] = {r1.x, r1.y}; v[2] = {r2.x, r2.y}; v[3] = {r3.x, r3.y};
        v[4] = {r4.x, r4.y}; v[5] = {r5.x, r5.y}; v[6] = {r6.x, r6.y}; v[7] = {r7.x, r7.y};
    }
    wave_lds_order();
#endif
    dft8(v);  // over a -> c
#pragma unroll
    // W64^(b*c).  The table is symmetric (W64^(q*i)): read as tw2[i][q], the eight distinct addresses of one
    // instruction are one contiguous 64-byte run (conflict-free); read as tw2[q][i] they are 64 bytes apart and
    // collide four ways (measured: ~100 LDS cycles per frame pair, a fifth of the kernel's LDS time).
    for (int i = 1; i < 8; ++i) v[i] = cmul(v[i], tw2[i * 8 + q]);
    const int sw = k1 & 3;  // column swizzle of the second exchange, in units of complex pairs
#pragma unroll
    for (int c = 0; c < 8; ++c) xbuf[k1 * XROW + 8 * c + (q ^ (2 * sw))] = v[c];
    wave_lds_order();
    // lane (k1, c=q): gather u[k1][c][b], b = 0..7; the pair (2p, 2p+1) sits in pair slot p ^ sw
    {
        const float4* row4 = reinterpret_cast<const float4*>(xbuf + k1 * XROW + 8 * q);
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            const float4 f = row4[pr ^ sw];
            v[2 * pr] = {f.x, f.y};
            v[2 * pr + 1] = {f.z, f.w};
        }
    }
    wave_lds_order();
    dft8(v);  // over b -> d
}

// Full-wavefront sum without LDS: scan inside the 16-lane rows, fold the rows, broadcast lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_shift_add(float v) {
    const float o = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true));
    return v + o;
}
__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_shift_add<0x111, 0xf>(v);  // row_shr:1
    v = dpp_shift_add<0x112, 0xf>(v);  // row_shr:2
    v = dpp_shift_add<0x114, 0xf>(v);  // row_shr:4
    v = dpp_shift_add<0x118, 0xf>(v);  // row_shr:8
    v = dpp_shift_add<0x142, 0xa>(v);  // row_bcast:15 -> rows 1, 3
    v = dpp_shift_add<0x143, 0xc>(v);  // row_bcast:31 -> rows 2, 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// The same for two values at once, the DPP operand fused into the addition (the builtin form costs a move per shift and
// two extra instructions per row broadcast).  The two chains alternate; with one s_nop a write and the DPP read of the
// same register are two wait states apart, as the VALU -> DPP hazard requires (inline asm hides it from the compiler).
__device__ __forceinline__ void wave_sum2(float& a, float& b) {
#define KWS_DPP_STEP(ctrl)                                                      \
    "v_add_f32_dpp %0, %0, %0 " ctrl "\n\t"                                      \
    "v_add_f32_dpp %1, %1, %1 " ctrl "\n\t"                                      \
    "s_nop 0\n\t"
    asm("s_nop 1\n\t"
        KWS_DPP_STEP("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
        KWS_DPP_STEP("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1")
        KWS_DPP_STEP("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1")
        KWS_DPP_STEP("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1")
        KWS_DPP_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")  // rows 1, 3 += lane 15 of the row below; rows 0, 2 keep their value
        KWS_DPP_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")  // rows 2, 3 += lane 31
        : "+v"(a), "+v"(b));
#undef KWS_DPP_STEP
    a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a), 63));
    b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, b), 63));
}


// Level equalisation of a packed frame pair.  The two real frames share one complex transform, so every float32
// rounding error of the transform is relative to the LARGER frame: a quiet frame packed with a loud one (a speech
// onset, two microphones at different gains) would get the loud frame's rounding noise, -140 dB below the LOUD frame,
// in its own spectrum -- measured 2e-2 in the cepstra of the frame before a burst.  Both frames are therefore scaled
// to energies in [1, 4) by exact powers of two before the transform, and the powers are scaled back (again exactly)
// when they are formed: a frame's error is then relative to its own level whatever its partner is, and a pair whose
// frames already have the same exponent gives bit-identical results to the unscaled transform.
// Returns the factors the POWERS must be multiplied by: 2^(2 s_a), 2^(2 s_b) where the frames were multiplied by 2^-s.
struct PairLevel {
    float pow_a, pow_b;
};
__device__ __forceinline__ PairLevel equalise_levels(cf (&v)[8]) {
    cf e2 = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) asm("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(e2) : "v"(v[i]));
    float ea = e2.x, eb = e2.y;
    wave_sum2(ea, eb);  // wave-uniform
    auto shift_of = [](float e) -> int {
        const int ex = (int)((__builtin_bit_cast(uint32_t, e) >> 23) & 0xffu);  // biased exponent of the energy
        if (ex == 0 || ex == 255) return 0;                                        // zero / denormal / non-finite: leave as is
        const int s = (ex - 127) >> 1;                                             // floor(log2(energy) / 2)
        return s < -30 ? -30 : (s > 30 ? 30 : s);                                  // 2^(+-60) on the powers stays far from the f32 limits
    };
    const int sa = shift_of(ea), sb = shift_of(eb);
    // Equal exponents (stationary signals: most pairs): scaling both frames by the same power of two commutes with every
    // rounding of the transform, so the unscaled transform gives the same bits -- skip the multiplies (wave-uniform branch).
    if (sa == sb) return {1.0f, 1.0f};
    const cf down = {__builtin_bit_cast(float, (uint32_t)(127 - sa) << 23), __builtin_bit_cast(float, (uint32_t)(127 - sb) << 23)};
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = v[i] * down;
    return {__builtin_bit_cast(float, (uint32_t)(127 + 2 * sa) << 23), __builtin_bit_cast(float, (uint32_t)(127 + 2 * sb) << 23)};
}

// First-pass twiddles of this lane, W512^(lane*i).
__device__ __forceinline__ void load_twiddles(const float2* __restrict__ tw, int lane, cf (&t1)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float2 a = tw[(lane * i) & 511];
        t1[i] = {a.x, a.y};
    }
}
// Second-pass twiddle table W64^(q*i) = W512^(8*q*i), [8][8] complex, into LDS (64 threads fill it).
__device__ __forceinline__ void fill_tw2(const float2* __restrict__ tw, cf* tw2, int idx) {
    if (idx < 64) {
        const float2 a = tw[(8 * (idx >> 3) * (idx & 7)) & 511];
        tw2[idx] = {a.x, a.y};
    }
}

// Separate the two real spectra packed in Z and write, for every bin 0..256, the pair
// (1/512*|A|^2, 1/512*|B|^2) (or the magnitudes) to pbuf, which ALIASES zbuf: all reads of the spectrum
// are issued before the first write (LDS executes a wavefront's instructions in order).  Returns this
// lane's share of the two frame energies.
// nza / nzb: whether frame a / b has any non-zero sample.  An all-zero frame must give exactly 0 (the
// reference then floors to eps); computed through the packed transform it would instead pick up the
// partner frame's float32 rounding noise (-140 dB), so it is forced.
// pslot[j]: power-buffer slot of bin lane + 64j (the MFCC path stores the bins grouped by mel chunk; the
// spectrum operators pass the identity); slot256: where bin 256 goes, or -1 to drop it.
// lv: the exact power-of-two factors that undo equalise_levels (applied to the powers before any square root).
__device__ __forceinline__ void split_power(const cf (&v)[8], cf* zbuf, float2* pbuf, int lane, int power, bool nza,
                                            bool nzb, const int (&pslot)[4], int slot256, const PairLevel& lv, float& ea,
                                            float& eb, float* pk_a = nullptr, float* pk_b = nullptr) {
    const int k1 = lane >> 3, q = lane & 7;
#pragma unroll
    for (int d = 0; d < 8; ++d) zbuf[zswz(k1 + 8 * q + 64 * d)] = v[d];
    wave_lds_order();
    cf z[4], w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = lane + 64 * j;
        z[j] = zbuf[zswz(k)];
        w[j] = zbuf[zswz((NFFT - k) & (NFFT - 1))];
    }
    wave_lds_order();
    const float scale = power ? (1.0f / (4.0f * NFFT)) : 0.25f;
    const float scale_a = scale * lv.pow_a, scale_b = scale * lv.pow_b;  // products of powers of two: exact
    float pa[4], pb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float ar = z[j].x + w[j].x, ai = z[j].y - w[j].y;  // 2*A
        const float br = z[j].y + w[j].y, bi = z[j].x - w[j].x;  // 2*B (up to a unit factor)
        pa[j] = fmaf(ar, ar, ai * ai) * scale_a;
        pb[j] = fmaf(br, br, bi * bi) * scale_b;
        if (!power) {
            pa[j] = sqrtf(pa[j]);
            pb[j] = sqrtf(pb[j]);
        }
    }
    if (!(nza && nzb)) {  // wave-uniform and rare (silence): a real branch, not ten selects on every frame pair
        asm volatile("" ::: "memory");  // keeps the compiler from converting the branch into selects
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            pa[j] = nza ? pa[j] : 0.f;
            pb[j] = nzb ? pb[j] : 0.f;
        }
    }
    if (pk_a) {  // this lane's largest bin power per frame: the precision flag measures the weakest mel band against the frame's peak
        // (v_max3_f32 in asm: fmaxf costs a canonicalising v_max per operand on top of the maximum itself)
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(*pk_a) : "v"(pa[0]), "v"(pa[1]), "v"(pa[2]));
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(*pk_b) : "v"(pb[0]), "v"(pb[1]), "v"(pb[2]));
        asm("v_max_f32 %0, %0, %1" : "+v"(*pk_a) : "v"(pa[3]));
        asm("v_max_f32 %0, %0, %1" : "+v"(*pk_b) : "v"(pb[3]));
    }
    ea = 0.f;
    eb = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        pbuf[pslot[j]] = make_float2(pa[j], pb[j]);
        ea += pa[j];
        eb += pb[j];
    }
    // bin 256 = Z[256] lives in lane 0, register 4 (k1 = 0, q = 0, d = 4); it is its own mirror image
    if (lane == 0) {
        float pa = (2.f * v[4].x) * (2.f * v[4].x) * scale_a, pb = (2.f * v[4].y) * (2.f * v[4].y) * scale_b;
        if (!power) {
            pa = sqrtf(pa);
            pb = sqrtf(pb);
        }
        pa = nza ? pa : 0.f;
        pb = nzb ? pb : 0.f;
        if (slot256 >= 0) pbuf[slot256] = make_float2(pa, pb);
        ea += pa;
        eb += pb;
        if (pk_a) {  // bin 256 joins this lane's peaks
            asm("v_max_f32 %0, %0, %1" : "+v"(*pk_a) : "v"(pa));
            asm("v_max_f32 %0, %0, %1" : "+v"(*pk_b) : "v"(pb));
        }
    }
    wave_lds_order();
}

constexpr float PSF_EPS = 2.220446049250313e-16f;  // numpy.finfo(float).eps, exactly 2^-52

// Sample -> float32 in [-1, 1): int16 PCM is scaled like librosa/soundfile do (x / 32768, exact);
// float32 input is taken as is (a signal the caller already decoded / augmented).
__device__ __forceinline__ float to_unit(int16_t s) { return (float)s * (1.0f / 32768.0f); }
__device__ __forceinline__ float to_unit(float s) { return s; }

// Views of the LDS a wavefront needs for one frame pair: its private scratch and the workgroup tables.
struct PairScratch {
    cf* xbuf;          // 8*XROW complex: exchange buffer / spectrum / power spectrum
    float2* pbuf;      // aliases xbuf
    float* lbuf;       // 2 x 64 centred log-mel values
    const float* dctb; // [numcep][nfp]
    const cf* tw2;     // [8][8]
    int nfp;
};

// Per-lane constants of the sparse mel stage: lane c owns chunk c (<= 8 bins of one inter-edge segment).
struct MelLane {
    float rw[MEL_CHUNK], fw[MEL_CHUNK];  // rising / falling weights of the chunk's bins (0 beyond its length)
    int pslot[4];                        // power-buffer slots of bins lane + 64j
    uint32_t gth;                        // filter `lane`: chunk ranges r0 | nr<<8 | f0<<16 | nf<<24
    float m1, m2, m4;                    // 1 if chunk lane+1 / +2 / +4 lies in the same inter-edge segment, else 0
    bool deep;                           // some segment has more than 4 chunks (wave-uniform)
};
__device__ __forceinline__ void load_mel_lane(const FrontendTables& t, int lane, MelLane& m) {
#pragma unroll
    for (int i = 0; i < MEL_CHUNK; ++i) {
        m.rw[i] = t.mel_rw[i * 64 + lane];
        m.fw[i] = t.mel_fw[i * 64 + lane];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) m.pslot[j] = t.mel_slot[lane + 64 * j];
    m.gth = t.mel_gather[lane];
    const int seg = t.mel_seg[lane];  // bit d: chunk lane + 2^d is in the same segment; bit 7: any segment > 4 chunks
    m.m1 = (seg & 1) ? 1.f : 0.f;
    m.m2 = (seg & 2) ? 1.f : 0.f;
    m.m4 = (seg & 4) ? 1.f : 0.f;
    m.deep = __any((seg & 128) != 0);
}

// A wavefront's scratch starts out as whatever the previous kernel left in LDS.  Power-buffer slots past a chunk's
// length (and the chunks of idle lanes) are never written but are read and multiplied by zero weights, so they must
// hold finite values: clear the scratch once per workgroup.
__device__ __forceinline__ void zero_scratch(unsigned char* scr, int lane) {
    static_assert(SCR_BYTES % (64 * 16) == 0, "one 16-byte store per lane and round");
#pragma unroll
    for (int i = 0; i < SCR_BYTES / (64 * 16); ++i)
        reinterpret_cast<float4*>(scr)[i * 64 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// Sums of v over the chunks lane, lane+1, ... that belong to the same segment (suffix sum by doubling: after the
// steps the FIRST chunk of every segment holds the segment total), for the four accumulators of a frame pair.
// Segments of up to 8 chunks; the host keeps every segment inside one 16-lane DPP row, so the shifts by 1, 2 and 4
// lanes are row_shl operands of the multiply-add itself: three instructions per sum (shifting across the whole
// wavefront takes 1 + 2 + 4 wave_shl:1 moves on top).
__device__ __forceinline__ void segment_suffix_sums(float& a, float& b, float& c, float& d, const MelLane& m) {
    // four independent chains interleaved: three other instructions sit between a write and its DPP read (the VALU ->
    // DPP hazard needs two wait states; inline asm hides it from the compiler, hence the leading s_nop)
    asm("s_nop 1\n\t"
        "v_fmac_f32_dpp %0, %0, %4 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %1, %1, %4 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %2, %2, %4 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %3, %3, %4 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %0, %0, %5 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %1, %1, %5 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %2, %2, %5 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %3, %3, %5 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d)
        : "v"(m.m1), "v"(m.m2));
    if (m.deep)
        asm("s_nop 1\n\t"
            "v_fmac_f32_dpp %0, %0, %4 row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "v_fmac_f32_dpp %1, %1, %4 row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "v_fmac_f32_dpp %2, %2, %4 row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "v_fmac_f32_dpp %3, %3, %4 row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:1"
            : "+v"(a), "+v"(b), "+v"(c), "+v"(d)
            : "v"(m.m4));
}

// Two sums and two maxima (of non-negative values) over the wavefront at once: four chains alternate, so a register is read by
// DPP three instructions after it was written and no s_nop is owed -- the two maxima ride in the slots wave_sum2 pads.
__device__ __forceinline__ void wave_sum2_max2(float& a, float& b, float& ma, float& mb) {
#define KWS_DPP_STEP(ctrl)                                                      \
    "v_add_f32_dpp %0, %0, %0 " ctrl "\n\t"                                      \
    "v_add_f32_dpp %1, %1, %1 " ctrl "\n\t"                                      \
    "v_max_f32_dpp %2, %2, %2 " ctrl "\n\t"                                      \
    "v_max_f32_dpp %3, %3, %3 " ctrl "\n\t"
    asm("s_nop 1\n\t"
        KWS_DPP_STEP("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
        KWS_DPP_STEP("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1")
        KWS_DPP_STEP("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1")
        KWS_DPP_STEP("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1")
        KWS_DPP_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
        KWS_DPP_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
        "s_nop 0"  // (the last write of %3 and whatever reads it next)
        : "+v"(a), "+v"(b), "+v"(ma), "+v"(mb));
#undef KWS_DPP_STEP
    a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a), 63));
    b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, b), 63));
    ma = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ma), 63));
    mb = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mb), 63));
}

// Precision flag of the float32 front end.  A float32 transform leaves rounding noise a fixed distance below
// the frame's strongest spectral component, so a mel band far below that PEAK carries a large relative error, which log, DCT
// and lifter turn into a cepstral error.  Measured against the float64 reference on 1.19 M frames (tools/fe_precision_audit.py):
// with r = log(largest bin power) - min_j log(mel_j), frames with r <= 10.0 stay within 6.7e-5, r in (10.2, 10.4] reaches
// 8.9e-5, (10.4, 10.6] 1.1e-4.  (Round 3 first used the span max - min of the log-mel values themselves: it cannot tell white
// noise, whose peak bin sits well below its strongest band, from a tone, whose peak IS its strongest band, and left four frames
// of 2.4 M at 1.2-1.5e-4.)  Frames over the threshold are redone in float64 by the refinement kernel (batched) or on the spot
// (streaming).
// v: the frame's centred log-mel value in lanes of the frame, 0 (= filter 0's value) in the others; lim: the frame's
// log(peak) - log(mel_0) - threshold in every lane of the frame.  FULL = false: two frames, one per 32-lane half -- the
// result is valid in lanes 31 and 63; FULL = true: one frame over 64 lanes, lane 63.  Returns the ballot of (min < lim).
// Running minimum by DPP row scans; a lane without a source keeps its value (no bound_ctrl); a write and the DPP read of the
// same register are two wait states apart (VALU -> DPP, hidden from the compiler).
template <bool FULL>
__device__ __forceinline__ unsigned long long peak_over(float v, float lim) {
    float mn = v;
#define KWS_MM_STEP(ctrl)                                      \
    "v_min_f32_dpp %0, %0, %0 " ctrl "\n\t"                    \
    "s_nop 1\n\t"
    asm("s_nop 1\n\t"
        KWS_MM_STEP("row_shr:1 row_mask:0xf bank_mask:0xf")
        KWS_MM_STEP("row_shr:2 row_mask:0xf bank_mask:0xf")
        KWS_MM_STEP("row_shr:4 row_mask:0xf bank_mask:0xf")
        KWS_MM_STEP("row_shr:8 row_mask:0xf bank_mask:0xf")
        "v_min_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf"  // rows 1, 3: lane 15 of the row below joins
        : "+v"(mn));
    if constexpr (FULL)
        asm("s_nop 1\n\t"
            "v_min_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"  // rows 2, 3: lane 31 joins
            : "+v"(mn));
#undef KWS_MM_STEP
    return __ballot(mn < lim);
}
// the same for two frames held in two registers (more than 32 filters): the two chains alternate
__device__ __forceinline__ void peak_over2(float va, float vb, float lim_a, float lim_b, bool& over_a, bool& over_b) {
    float a = va, b = vb;
#define KWS_MM_STEP(ctrl)                                      \
    "v_min_f32_dpp %0, %0, %0 " ctrl "\n\t"                    \
    "v_min_f32_dpp %1, %1, %1 " ctrl "\n\t"                    \
    "s_nop 0\n\t"
    asm("s_nop 1\n\t"
        KWS_MM_STEP("row_shr:1 row_mask:0xf bank_mask:0xf")
        KWS_MM_STEP("row_shr:2 row_mask:0xf bank_mask:0xf")
        KWS_MM_STEP("row_shr:4 row_mask:0xf bank_mask:0xf")
        KWS_MM_STEP("row_shr:8 row_mask:0xf bank_mask:0xf")
        KWS_MM_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
        KWS_MM_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
        : "+v"(a), "+v"(b));
#undef KWS_MM_STEP
    over_a = (__ballot(a < lim_a) >> 63) & 1ull;
    over_b = (__ballot(b < lim_b) >> 63) & 1ull;
}

// One packed frame pair, from the (pre-emphasised, zero-padded) samples in v to the cepstra in global memory:
// FFT -> split -> power -> sparse mel -> log -> DCT x lifter, c0 = log(frame energy).
// out_a / out_b: rows of numcep floats for frame a / b (out_b is not touched when has_b is false).
// Returns the precision flags of the pair (wave-uniform): bit 0 = frame a, bit 1 = frame b has a mel band more than p.refine_span
// (natural-log units of power) below its largest spectral bin.
__device__ __forceinline__ uint32_t mfcc_pair(cf (&v)[8], bool nza, bool nzb, bool has_b, const FrontendParams& p,
                                          const PairScratch& sc, const cf (&t1)[8], const MelLane& ml, int lane,
                                          float* __restrict__ out_a, float* __restrict__ out_b,
                                          float* lds_out_a = nullptr) {
    cf* xbuf = sc.xbuf;
    float2* pbuf = sc.pbuf;
    float* lbuf = sc.lbuf;
    const float* dctb = sc.dctb;
    const cf* tw2 = sc.tw2;
    const uint32_t gth = ml.gth;
    const int nfp = sc.nfp;
    const PairLevel lv = equalise_levels(v);
#if !defined(KWS_X_MFCC_STOP) || KWS_X_MFCC_STOP >= 1
    fft512(v, xbuf, t1, tw2, lane);
#endif
#if defined(KWS_X_MFCC_STOP) && KWS_X_MFCC_STOP <= 1   // counter attribution (tools/pmc_mfcc_variant.sh): stop after the transform; wrong results
    if (lane < p.numcep) out_a[lane] = v[0].x + v[1].y + v[2].x + v[3].y + v[4].x + v[5].y + v[6].x + v[7].y + lv.pow_a;
    return 0u;
#endif

    float ea, eb, pka = 0.f, pkb = 0.f;  // frame energies; largest bin power of each frame (the precision flag's reference)
    if (p.refine_span > 0.f) {  // wave-uniform
        split_power(v, xbuf, pbuf, lane, 1, nza, nzb, ml.pslot, -1, lv, ea, eb, &pka, &pkb);
        wave_sum2_max2(ea, eb, pka, pkb);
    } else {
        split_power(v, xbuf, pbuf, lane, 1, nza, nzb, ml.pslot, -1, lv, ea, eb);
        wave_sum2(ea, eb);
    }
#if defined(KWS_X_MFCC_STOP) && KWS_X_MFCC_STOP == 2    // stop after the power spectrum
    if (lane < p.numcep) out_a[lane] = ea + eb + pbuf[lane].x;
    return 0u;
#endif

    // sparse mel: this lane's chunk of <= 8 bins is one contiguous 64-byte run of the power buffer (four
    // conflict-free ds_read_b128); slots past the chunk's length hold stale finite values and meet zero weights
    float ra = 0.f, fa_ = 0.f, rb = 0.f, fb_ = 0.f;
    {
        const float4* pc = reinterpret_cast<const float4*>(pbuf + lane * MEL_STRIDE);
#pragma unroll
        for (int h = 0; h < MEL_CHUNK / 2; ++h) {
            const float4 pw = pc[h];  // (bin 2h: frame a, frame b), (bin 2h+1: frame a, frame b)
            ra = fmaf(ml.rw[2 * h], pw.x, ra);
            fa_ = fmaf(ml.fw[2 * h], pw.x, fa_);
            rb = fmaf(ml.rw[2 * h], pw.y, rb);
            fb_ = fmaf(ml.fw[2 * h], pw.y, fb_);
            ra = fmaf(ml.rw[2 * h + 1], pw.z, ra);
            fa_ = fmaf(ml.fw[2 * h + 1], pw.z, fa_);
            rb = fmaf(ml.rw[2 * h + 1], pw.w, rb);
            fb_ = fmaf(ml.fw[2 * h + 1], pw.w, fb_);
        }
    }
    // Filter j = rising sum over segment j + falling sum over segment j+1.  The chunks of a segment are adjacent
    // lanes: a suffix sum by doubling (DPP shifts, no LDS) leaves each segment's total in its first chunk, and lane
    // j fetches its two totals through the LDS crossbar (ds_bpermute: no memory, no bank conflicts).
    segment_suffix_sums(ra, fa_, rb, fb_, ml);
    float ma, mb;
    uint32_t flags = 0u;
    {
        const int r0 = gth & 255, nr = (gth >> 8) & 255, q0 = (gth >> 16) & 255, nq = gth >> 24;
        const float gra = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * r0, __builtin_bit_cast(int, ra)));
        const float grb = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * r0, __builtin_bit_cast(int, rb)));
        const float gfa = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * q0, __builtin_bit_cast(int, fa_)));
        const float gfb = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * q0, __builtin_bit_cast(int, fb_)));
        const float sa = (nr ? gra : 0.f) + (nq ? gfa : 0.f);
        const float sb = (nr ? grb : 0.f) + (nq ? gfb : 0.f);
        if (p.nfilt <= 32) {
            // both frames through ONE log: frame b's filterbank energies move to lanes 32.. (v_permlane32_swap
            // exchanges the upper half of its first operand with the lower half of its second)
            const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, sa), __builtin_bit_cast(unsigned, sb), false, false);
            const float s = __builtin_bit_cast(float, (unsigned)sw[0]);
            const float l = logf(s == 0.f ? PSF_EPS : s);
            // DCT rows k >= 1 are orthogonal to constants: removing the common mode L_0 removes the float32
            // table-rounding error a -36 log-floor would otherwise amplify.
            ma = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, l), 0));
            mb = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, l), 32));
#if defined(KWS_X_MFCC_STOP) && KWS_X_MFCC_STOP == 3    // stop after mel + log
            if (lane < p.numcep) out_a[lane] = l + ea + eb;
            return 0u;
#endif
            const int f = lane >> 5, j = lane & 31;
            const float cv = j < p.nfilt ? l - (f ? mb : ma) : 0.f;
            lbuf[64 * f + j] = cv;
            if (p.refine_span > 0.f) {  // wave-uniform
                // log(peak) - log(mel_0) - threshold per frame, both frames through one hardware logarithm (v_log_f32: a threshold
                // needs no more; an all-zero frame: log 0 = -inf, never flagged)
                const float lim = 0.6931471806f * __builtin_amdgcn_logf(f ? pkb : pka) - (f ? mb : ma) - p.refine_span;  // (v_log_f32, bare: a peak power is no subnormal)
                const unsigned long long over = peak_over<false>(cv, lim);
                flags = (uint32_t)((over >> 31) & 1u) | ((uint32_t)((over >> 63) & 1u) << 1);
            }
        } else {
            const float la = logf(sa == 0.f ? PSF_EPS : sa), lb = logf(sb == 0.f ? PSF_EPS : sb);
            ma = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, la)));
            mb = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, lb)));
            const float cva = lane < p.nfilt ? la - ma : 0.f, cvb = lane < p.nfilt ? lb - mb : 0.f;
            lbuf[lane] = cva;
            lbuf[64 + lane] = cvb;
            if (p.refine_span > 0.f) {
                bool oa, ob;
                peak_over2(cva, cvb, 0.6931471806f * __builtin_amdgcn_logf(pka) - ma - p.refine_span, 0.6931471806f * __builtin_amdgcn_logf(pkb) - mb - p.refine_span, oa, ob);
                flags = (uint32_t)oa | ((uint32_t)ob << 1);
            }
        }
    }
    wave_lds_order();

    // DCT-II(ortho) x lifter: lane -> (frame f = lane>>5, coefficient i = lane&31)
    {
        const int f = lane >> 5, i = lane & 31;
        if (i < p.numcep && (f == 0 || has_b)) {
            const float4* L4 = reinterpret_cast<const float4*>(lbuf + 64 * f);
            const float4* D4 = reinterpret_cast<const float4*>(dctb + i * nfp);
            // the reference row length (26 filters -> 7 float4) gets a compile-time trip count: loads in batches of
            // four pairs ahead of their multiply-adds (two LDS round trips instead of seven; same order of additions)
            float acc = 0.f;
            if (nfp == 28) {
#pragma unroll
                for (int j0 = 0; j0 < 7; j0 += 4) {
                    float4 d[4], l[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (j0 + j < 7) d[j] = D4[j0 + j], l[j] = L4[j0 + j];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (j0 + j < 7) {
                            acc = fmaf(d[j].x, l[j].x, acc);
                            acc = fmaf(d[j].y, l[j].y, acc);
                            acc = fmaf(d[j].z, l[j].z, acc);
                            acc = fmaf(d[j].w, l[j].w, acc);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                for (int j = 0; j < nfp / 4; ++j) {
                    const float4 d = D4[j], l = L4[j];
                    acc = fmaf(d.x, l.x, acc);
                    acc = fmaf(d.y, l.y, acc);
                    acc = fmaf(d.z, l.z, acc);
                    acc = fmaf(d.w, l.w, acc);
                }
            }
            if (i == 0) {
                if (p.append_energy) {
                    const float e = f ? eb : ea;
                    acc = logf(e == 0.f ? PSF_EPS : e);
                } else {
                    float dsum = 0.f;
                    for (int j = 0; j < p.nfilt; ++j) dsum += dctb[j];
                    acc = fmaf(f ? mb : ma, dsum, acc);
                }
            }
            (f ? out_b : out_a)[i] = acc;
            if (lds_out_a && f == 0) lds_out_a[i] = acc;  // the streaming push inside the DS-CNN kernel: the row goes straight to the feature map
        }
    }
    wave_lds_order();
    return has_b ? flags : (flags & 1u);
}


// ------------------------------------------------------------------------------------------------
// Streaming front end (BASELINE config 5: 10 ms hops), the work of ONE wavefront: every push brings frame_step new
// samples per stream and completes exactly one new frame per stream.  The wavefront serves stream sa and, when has_b,
// stream sb (their frames are the two halves of one packed FFT).  hops = pushes seen BEFORE this one; the new frame is
// frame hops - 2 of the continuous signal and covers samples [step*(hops-2), step*(hops-2) + frame_len).  Samples of
// the current hop are taken from `hop`, older ones from the per-stream PCM ring, which is also updated here.  The
// cepstra go to row (frame index mod num_frames) of each stream's feature ring and, when lds_out_a is not null, stream
// sa's also to those numcep floats in LDS.  smem: stream_frame_lds_bytes(p) bytes of LDS, private to the wavefront.
// Returns the frame's index in the continuous signal, or -1 while the stream is younger than one frame.
constexpr int STREAM_F64_BYTES = 16 * NFFT + 8 * 128;  // float64 redo of a flagged frame: the transform buffer and two log-mel vectors
__host__ __device__ __forceinline__ size_t stream_frame_lds_bytes(const FrontendParams& p) {
    const int nfp = (p.nfilt + 3) & ~3;
    return sizeof(float) * (size_t)(((p.numcep * nfp + 3) & ~3) + 2 * 64) + SCR_BYTES + STREAM_F64_BYTES;
}

// float64 recomputation of ONE frame (the streaming push's flagged frames): get(i) = pre-emphasised sample
// i < frame_len, float32 as the reference forms it.  The frame rides alone in its transform (imaginary half empty), the
// tables are read where they lie in global memory -- a rare path.  X: NFFT d2 of LDS, L: 128 doubles, private to the wave.
template <typename F>
__device__ __forceinline__ void f64_redo_frame(const FrontendParams& p, const FrontendTables& t, d2* X, double* L, F&& get,
                                               float* __restrict__ out, float* lds_out, int lane) {
    bool nz = false;
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) {
        const int i = 64 * n1 + lane;
        const float y = i < p.frame_len ? get(i) : 0.f;
        nz |= y != 0.f;
        X[sw512(i)] = d2{(double)y, 0.0};
    }
    nz = __any(nz);
    wave_order();
    spectrum_pair<true, NFFT>(X, X, reinterpret_cast<const d2*>(t.tw64), NFFT, 9, p.frame_len, 1.0 / (double)NFFT, lane);
    const F64Tabs tb = {reinterpret_cast<const d2*>(t.tw64), t.mel_w64, t.dct64, t.mel_edges};
    f64_tail<NBINS>(p, tb, NBINS, X, L, nz, false, false, out, out, lds_out, lane);
}
__device__ __forceinline__ long stream_frame_wave(const FrontendParams& p, const FrontendTables& t,
                                                  const int16_t* __restrict__ hop, int sa, int sb, bool has_b,
                                                  int16_t* __restrict__ pcm_ring, int ring_len,
                                                  float* __restrict__ feat_ring, int hops, unsigned char* smem, int lane,
                                                  float* lds_out_a, int* refine_ctr = nullptr) {
    const int nfp = (p.nfilt + 3) & ~3;
    float* dctb = reinterpret_cast<float*>(smem);
    cf* tw2 = reinterpret_cast<cf*>(dctb + ((p.numcep * nfp + 3) & ~3));
    unsigned char* scr = reinterpret_cast<unsigned char*>(tw2 + 64);
    const int step = p.frame_step;
    const long base = (long)step * hops;            // absolute index of the first sample of this hop

    {   // both float4 of a lane in flight together (13 x 28 floats = 91 float4 at the reference's geometry)
        const int n4 = (p.numcep * nfp + 3) / 4;
        const float4* src = reinterpret_cast<const float4*>(t.dct_pad);
        for (int i0 = 0; i0 < n4; i0 += 128) {
            const int i = i0 + lane, j = i + 64;
            const float4 a = i < n4 ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 b = j < n4 ? src[j] : make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < n4) reinterpret_cast<float4*>(dctb)[i] = a;
            if (j < n4) reinterpret_cast<float4*>(dctb)[j] = b;
        }
    }
    fill_tw2(t.twiddle, tw2, lane);
    zero_scratch(scr, lane);
    cf t1[8];
    load_twiddles(t.twiddle, lane, t1);
    MelLane ml;
    load_mel_lane(t, lane, ml);

    // Sample at offset d from the start of this hop (d < step; absolute index base + d) of stream s: d >= 0 -> d_hop,
    // d < 0 -> the per-stream ring, before the stream began -> 0.  All index arithmetic is 32-bit relative to the hop
    // (a 64-bit modulo per sample was most of this kernel's time), one load per sample through a selected pointer, and
    // the previous sample of the pre-emphasis comes from the neighbouring lane: 16 independent loads per lane instead
    // of 32 dependent branches.
    const long f_start = base + step - ((p.frame_len + step - 1) / step) * step;  // = step * (hops - 2) for 400/160
    const bool frame_ok = f_start >= 0;
    const long fidx = f_start / step;
    const int base_mod = (int)(base % ring_len);
    const int d0 = (int)(f_start - base);                                 // <= 0, >= -ring_len
    const int first_d = base < (long)ring_len ? -(int)base : -ring_len;   // offsets below it precede the stream (or the ring)
    auto sample = [&](int s, int d) -> float {
        int r = base_mod + d;
        r += r < 0 ? ring_len : 0;
        const int dc = d < first_d ? first_d : d;  // keep the address in bounds; the value is dropped below
        const int16_t* ptr = d >= 0 ? hop + (size_t)s * step + d : pcm_ring + (size_t)s * ring_len + (dc == d ? r : 0);
        const float x = to_unit(*ptr);
        return d < first_d ? 0.f : x;
    };
    cf v[8];
    bool nza = false, nzb = false;
    float carry_a = 0.f, carry_b = 0.f;  // sample just before this lane block (lane 63 of the previous block)
    // The hop's copy for the ring append is fetched HERE, together with the frame's own loads, and held in registers: when
    // the hop lives in pinned host memory (kws_stream_push_host_i16) every dependent load is a PCIe round trip, and this
    // way the wavefront pays one, not two.
    int16_t app_a[8], app_b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        app_a[j] = 0;
        app_b[j] = 0;
        if (64 * j < step) {  // wave-uniform
            const int i = 64 * j + lane;
            if (i < step) {
                app_a[j] = hop[(size_t)sa * step + i];
                if (has_b) app_b[j] = hop[(size_t)sb * step + i];
            }
        }
    }
    if (frame_ok) {
        carry_a = sample(sa, d0 - 1);
        if (has_b) carry_b = sample(sb, d0 - 1);
    }
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) {
        const int i = 64 * n1 + lane;
        float ya = 0.f, yb = 0.f;
        if (frame_ok && 64 * n1 < p.frame_len) {  // wave-uniform
            const bool in = i < p.frame_len;
            const int d = d0 + (in ? i : 0);
            const bool first = f_start + i == 0;  // the stream's very first sample is not pre-emphasised
            const float ca = sample(sa, d);
            float pa = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ca), 0x138, 0xf, 0xf, false));  // wave_shr:1
            pa = lane == 0 ? carry_a : pa;
            carry_a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ca), 63));
            ya = in ? (first ? ca : __fsub_rn(ca, __fmul_rn(p.preemph, pa))) : 0.f;
            if (has_b) {
                const float cb = sample(sb, d);
                float pb = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, cb), 0x138, 0xf, 0xf, false));
                pb = lane == 0 ? carry_b : pb;
                carry_b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cb), 63));
                yb = in ? (first ? cb : __fsub_rn(cb, __fmul_rn(p.preemph, pb))) : 0.f;
            }
        }
        v[n1].x = ya;
        v[n1].y = yb;
        nza |= ya != 0.f;
        nzb |= yb != 0.f;
    }
    nza = __any(nza);
    nzb = __any(nzb);
    // Append the hop to the rings now, so that the stores complete under the transform: the slots they overwrite,
    // [base, base + step) mod ring_len, are older than anything the frame read (the ring is one hop longer than the
    // frame's reach, kws_stream_open), and every read above is earlier in this wavefront's program order.
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = 64 * j + lane;
        if (64 * j < step && i < step) {
            int pos = base_mod + i;
            pos -= pos >= ring_len ? ring_len : 0;
            pcm_ring[(size_t)sa * ring_len + pos] = app_a[j];
            if (has_b) pcm_ring[(size_t)sb * ring_len + pos] = app_b[j];
        }
    }
    wave_lds_order();
    if (frame_ok) {
        const PairScratch sc = {reinterpret_cast<cf*>(scr + SCR_XBUF), reinterpret_cast<float2*>(scr + SCR_PBUF),
                                reinterpret_cast<float*>(scr + SCR_LBUF),
                                dctb, tw2, nfp};
        const int row = (int)(fidx % p.num_frames);
        float* out_a = feat_ring + ((size_t)sa * p.num_frames + row) * p.numcep;
        float* out_b = feat_ring + ((size_t)sb * p.num_frames + row) * p.numcep;
        const uint32_t flags = mfcc_pair(v, nza, nzb, has_b, p, sc, t1, ml, lane, out_a, out_b, lds_out_a);
        if (flags) {  // wave-uniform and rare: a frame float32 cannot hold to 1e-4 is redone in float64, here and now
            // The hop is in the ring by now (the append above), older samples were never overwritten: gather again.
            d2* X = reinterpret_cast<d2*>(scr + SCR_BYTES);
            double* L = reinterpret_cast<double*>(scr + SCR_BYTES + 16 * NFFT);
            auto frame_of = [&](int s) {
                return [&, s](int i) -> float {
                    const float cur = sample(s, d0 + i);
                    return f_start + i == 0 ? cur : __fsub_rn(cur, __fmul_rn(p.preemph, sample(s, d0 + i - 1)));
                };
            };
            if (flags & 1u) f64_redo_frame(p, t, X, L, frame_of(sa), out_a, lds_out_a, lane);
            if (flags & 2u) f64_redo_frame(p, t, X, L, frame_of(sb), out_b, nullptr, lane);
            if (refine_ctr && lane == 0) atomicAdd(refine_ctr + 5, (int)__builtin_popcount(flags));
        }
    }
    return frame_ok ? fidx : -1;
}

}  // namespace
}  // namespace kws
