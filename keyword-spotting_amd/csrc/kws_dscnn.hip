// DS-CNN forward for gfx950 (MI355X): one 512-thread workgroup per clip, every activation resident in the
// CU's 160 KiB LDS; conv1 and the pointwise 1x1 convolutions on the matrix cores, depthwise 3x3 on the VALU
// straight into the MFMA B-operand registers.
//
// Replaces DepthwiseSeparableConv.forward (reference kws/libs/models.py:160-183; rows a9-a15 of
// SURVEY.md section 8) for the [1,99,10] MFCC map:
//   conv1  1->64, 10x10, stride 2, pad 2, ReLU                      -> 64 x 47 x 3
//   4 x { depthwise 3x3 pad 1 ; pointwise 1x1 *padding=1* ; ReLU }  -> 64 x (49x5, 51x7, 53x9, 55x11)
//   global average pool, Linear(64 -> C), argmax (first maximum wins)
//
// Three arithmetic routes for the GEMMs, same f32-grade results (include/kws_hip.h, kws_set_pointwise_math):
//   f16 pairs (product, MODE 5): every f32 operand, scaled by a per-clip power of two, = hi + lo, two f16 pieces (22 bits);
//     three piece products on v_mfma_f32_32x32x16_f16 into one f32 accumulator.  The activations live in LDS in per-clip
//     power-of-two units; their exponents are decided two layers ahead from measured maxima and weight-derived bounds, so no
//     input overflows f16 (PairCtx, kws_dscnn_fwd_kernel).
//   split-bf16 (MODE 4, the product path of rounds 1-2): every f32 operand = hi + mid + lo, three bf16 pieces that reproduce it
//     exactly; the six piece products of combined order <= 2 on v_mfma_f32_32x32x16_bf16 (f32 accumulate) give the f32 product
//     to 2^-24.  16x the f32 MFMA rate, and the 16-bit matrix pipe runs beside the VALU (the f32 MFMA shares its datapath).
//   f32: v_mfma_f32_32x32x2_f32.
// A third variant runs the GEMMs on the VALU: an independent check of the operand mappings (tests only).
//
// The relu(bias) ring.  The reference's 1x1 convolution with padding=1 surrounds each block's output
// with a ring equal to relu(bias) (models.py:104-106).  The ring is never stored: each channel plane in
// LDS holds only the "interior" H x W values followed by two extra slots, [P] = relu(bias[c]) and
// [P+1] = 0.  A depthwise tap that falls on the ring reads slot P, one that falls outside the padded
// map reads slot P+1, so a stencil tap is an unconditional LDS read at a per-lane precomputed address.
//
// MFMA mapping, split path (32x32x16, D[i][j] += A[i][k] * B[k][j]): i = output channel, j = position, k = input
// channel.  Lane l supplies B[k = 8(l>>5) + e][j = l&31], e = 0..7: it computes the depthwise output of column j
// for input channels 16m + 8(l>>5) + e itself (m = k-block), splits the eight values into bf16 pieces and feeds
// them to the matrix core without touching LDS; A = pre-split weights from a two-k-block register ring.
// (f32 path, 32x32x2: lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31]; its 32 k-steps walk the
// same channels in the same order.)
// D: column = lane&31 (position), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (output channel).
//
// Stencil with 3 LDS reads instead of 9.  The 32 MFMA columns of a tile are 30 consecutive output
// positions plus one halo column on each side.  A lane reads only its own column (rows h-1, h, h+1);
// because the depthwise weights are the same in all 32 lanes of a half-wave, each lane forms the two
// 3-tap column sums its right and left neighbours need, and the neighbours' sums arrive through one fused DPP
// multiply-add each.  A neighbour that belongs to another row (x == 0 or x == W-1) is outside the zero-padded
// map, so its contribution is multiplied by a per-lane 0/1 mask.  The reads of step s+2 are in flight while
// step s is evaluated and the matrix core works through the MFMAs of the previous k-block.
//
// LDS map (floats): planes hold P+2 floats and are interleaved in channel pairs (see pidx)
//   Z3 (block3 out, 51x7)  @ 0      .. 22976     Z2 (block2 out, 49x5) @ 22976 .. 38784
//   Z1 (block1 out, 47x3)  @ 0      .. 9152      Z0 (conv1 out, 47x3)  @ 9152  .. 18304
//   padded MFCC 103x14     @ 18304  .. 19746     (conv1 phase only)
//   conv1 operand windows  @ 0      .. 3090      (f16 pairs, conv1 phase only: where Z1 goes afterwards; conv1_build_windows)
//   misc                   @ 38784  .. 40960     2 x depthwise table, 2 x pointwise bias, pooled, per-wavefront stage maxima (f16 pairs)
#include <type_traits>

#include "kws_internal.h"
#include "kws_mfcc_dev.h"
#include "kws_split_mfma.h"

namespace kws {
namespace {


#ifndef KWS_DSCNN_WAVES
#define KWS_DSCNN_WAVES 8
#endif
#ifndef KWS_X_DSCNN_STAMP_TID   // diagnostics builds (tools/build_variant.sh): which thread writes the phase stamps
#define KWS_X_DSCNN_STAMP_TID 0
#endif
constexpr int NW = KWS_DSCNN_WAVES;  // wavefronts per workgroup (8 = 2 per SIMD; 12 = 3 per SIMD measured slower)
constexpr int NT = NW * 64;
constexpr int TW = 30;               // output positions per tile (32 MFMA columns - 2 halo columns)

constexpr int P0 = C1_H * C1_W;                  // 141
constexpr int FEAT_H = 103, FEAT_W = 14;         // MFCC zero-padded by 2 (top/left) and up to the conv1 reach
constexpr int OFF_Z3 = 0, OFF_Z2 = 22976, OFF_Z1 = 0, OFF_Z0 = 9152, OFF_FEAT = 18304;
constexpr int OFF_DWTAB = 38784;                 // [2][64][12]  double-buffered per block
constexpr int OFF_PWB = OFF_DWTAB + 2 * 768;     // [2][64]      pointwise bias, double-buffered
constexpr int OFF_POOLED = OFF_PWB + 2 * 64;     // [64]
constexpr int OFF_POOLBUF = OFF_DWTAB;           // [NW][64] aliases depthwise buffer 0 (idle during block 4)
constexpr int OFF_WMAX = OFF_POOLED + 64 + 8;    // [4][NW] f16-pair arithmetic: per-wavefront maxima of a stage's stored output.  Sets: features 0,
                                                 // conv1 1, block 1 2 (+ its leftover combine 3), block 2 0 (+ combine 1): a set is
                                                 // rewritten two barriers after its last reader at the earliest
constexpr int LDS_FLOATS = 40960;                // 160 KiB
static_assert(OFF_WMAX + 4 * NW <= LDS_FLOATS, "LDS overflow");
static_assert(NW * 64 <= 768, "pool scratch must fit one depthwise buffer");
static_assert(OFF_FEAT + FEAT_H * FEAT_W <= OFF_Z2, "feature pad overlaps Z2");
static_assert(CH * 12 <= 2 * NT, "table staging assumes at most two elements per thread");

// Activation planes in LDS are stored as channel PAIRS interleaved per position: element (c, p) of a map whose
// planes hold S floats lives at (c >> 1) * 2S + 2p + (c & 1).  One ds_read_b64 then fetches a column's value for
// two consecutive channels (the split path walks channels two at a time) and the epilogue stores two output
// channels with one ds_write_b64: half the tap reads and stores, at twice the bytes per LDS cycle.
__device__ __forceinline__ constexpr int pidx(int c, int p, int S) { return (c >> 1) * 2 * S + 2 * p + (c & 1); }

// Geometry of block N (1..4): output plane H x W (all of it is the next block's interior).
template <int N>
struct Blk {
    static constexpr int H = 45 + 2 * N, W = 1 + 2 * N;           // 47x3, 49x5, 51x7, 53x9
    static constexpr bool RING = N > 1;                            // block 1 reads conv1's output: no ring
    static constexpr int HI = RING ? H - 2 : H, WI = RING ? W - 2 : W;  // stored input plane
    static constexpr int PIN = HI * WI, SIN = PIN + 2;
    static constexpr int POUT = H * W, SOUT = POUT + 2;
    static constexpr int OFF_IN = N == 1 ? OFF_Z0 : N == 2 ? OFF_Z1 : N == 3 ? OFF_Z2 : OFF_Z3;
    static constexpr int OFF_OUT = N == 1 ? OFF_Z1 : N == 2 ? OFF_Z2 : OFF_Z3;  // block 4 stores nothing
    static constexpr int TILES = (POUT + TW - 1) / TW;
    static constexpr int BUF = (N - 1) & 1;                       // which depthwise / bias buffer it reads
};

// Leftover tiles (round 3).  Block 1 has 5 tiles and block 2 has 9 for 8 wavefronts: the fifth / ninth tile costs a whole
// extra unit on one wavefront while others idle (block 2: 12.5 k cycles for 9 tiles, block 3: 13 k for 12).  With
// KWS_DSCNN_KSPLIT_LEFTOVER that tile is cut along K instead: four wavefronts take one k-block (16 input channels, 8 steps)
// each, write their 64 x positions partial sums to a dead region of LDS, and after the block's barrier all threads add the
// four partials in a fixed order, add the bias, apply ReLU and store (leftover_combine; one more barrier).  Block 1: tiles
// 0-3 on wavefronts 0-3, the leftover on 4-7 (one per SIMD); block 2: tiles 0-7 on all eight, the leftover as a second,
// quarter-size unit of the older wavefronts 0-3.
#ifndef KWS_DSCNN_KSPLIT_LEFTOVER
#define KWS_DSCNN_KSPLIT_LEFTOVER 1
#endif
template <int N>
struct Leftover {
    static constexpr bool HAS = KWS_DSCNN_KSPLIT_LEFTOVER && (N == 1 || N == 2);
    static constexpr int TILE = N == 1 ? 4 : 8;                     // the tile that is K-split
    static constexpr int P0T = TILE * TW;                           // its first position
    static constexpr int NP = HAS ? Blk<N>::POUT - P0T : 1;         // its positions: 21 (block 1), 5 (block 2)
    static constexpr int WAVE0 = N == 1 ? 4 : 0;                    // wavefronts WAVE0 .. WAVE0 + 3 take k-blocks 0 .. 3
    static constexpr int OFF_PART = N == 1 ? OFF_Z2 : OFF_Z0;       // [4][64][NP] partial sums, in a plane that is dead during block N
};
static_assert(Blk<1>::TILES == 5 && Blk<2>::TILES == 9, "the leftover tiles are the fifth of block 1 and the ninth of block 2");
static_assert(4 * CH * Leftover<1>::NP <= 38784 - OFF_Z2 && OFF_Z0 + 4 * CH * Leftover<2>::NP <= OFF_FEAT, "partial sums fit their dead planes");

// Depthwise 3x3 (+bias) at this lane's column from its three own-column inputs: nine multiply-adds and two
// fused DPP multiply-adds that pull the neighbouring lanes' column sums across the wavefront (0 shifted in at the
// ends).  Written as one asm block so that (a) the shift and the multiply-add are one instruction each
// (v_fmac_f32_dpp; the compiler emits v_mov_b32_dpp + v_fmac), and (b) each DPP source is written three
// instructions before it is read -- the VALU-write -> DPP-read hazard needs two wait states and the hazard
// recognizer does not look inside inline asm.  w0..w8 row-major taps, b bias.  TO_MFMA: the result is fed straight
// to a matrix-core instruction (f32 path), which needs two more wait states after the last VALU write.
template <bool TO_MFMA = false>
__device__ __forceinline__ float stencil3x3(float w0, float w1, float w2, float w3, float w4, float w5, float w6,
                                            float w7, float w8, float b, float up, float mid, float dn, float mask_l,
                                            float mask_r) {
    float c, to_right, to_left;
    asm("v_mul_f32 %1, %3, %13\n\t"          // to_right = w0*up   (what lane+1 needs: its (.., -1) taps)
        "v_mul_f32 %2, %5, %13\n\t"          // to_left  = w2*up   (what lane-1 needs: its (.., +1) taps)
        "v_fma_f32 %0, %4, %13, %12\n\t"     // c = w1*up + b
        "v_fmac_f32 %1, %6, %14\n\t"         // to_right += w3*mid
        "v_fmac_f32 %2, %8, %14\n\t"         // to_left  += w5*mid
        "v_fmac_f32 %0, %7, %14\n\t"         // c += w4*mid
        "v_fmac_f32 %1, %9, %15\n\t"         // to_right += w6*dn
        "v_fmac_f32 %2, %11, %15\n\t"        // to_left  += w8*dn
        "v_fmac_f32 %0, %10, %15\n\t"        // c += w7*dn
        "v_fmac_f32_dpp %0, %1, %16 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"   // c += to_right[lane-1]*mask_l
        "v_fmac_f32_dpp %0, %2, %17 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"        // c += to_left[lane+1]*mask_r
        : "=&v"(c), "=&v"(to_right), "=&v"(to_left)
        : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(w4), "v"(w5), "v"(w6), "v"(w7), "v"(w8), "v"(b), "v"(up), "v"(mid),
          "v"(dn), "v"(mask_l), "v"(mask_r));
    if constexpr (TO_MFMA) asm volatile("s_nop 1" : "+v"(c));
    return c;
}
// The depthwise table is interleaved in channel PAIRS: ten (channel c, channel c + 1) pairs -- nine taps and the bias -- are 80
// bytes, FIVE ds_read_b128 for two stencil steps where a table per channel took six.  (A wavefront's issue slots are what
// this kernel is made of; `tools/experiments/dscnn_unit_ablations.patch`: without the weight reloads it runs 9.5 % faster.)
struct DwPair {
    float4 l[5];  // (w0, w1) (w2, w3) (w4, w5) (w6, w7) (w8, b), each a (channel c, channel c + 1) pair
};
template <bool TO_MFMA, int E>  // E: which channel of the pair
__device__ __forceinline__ float stencil3x3_of_pair(const DwPair& w, float up, float mid, float dn, float mask_l, float mask_r) {
    if constexpr (E == 0)
        return stencil3x3<TO_MFMA>(w.l[0].x, w.l[0].z, w.l[1].x, w.l[1].z, w.l[2].x, w.l[2].z, w.l[3].x, w.l[3].z, w.l[4].x, w.l[4].z, up, mid, dn,
                                   mask_l, mask_r);
    else
        return stencil3x3<TO_MFMA>(w.l[0].y, w.l[0].w, w.l[1].y, w.l[1].w, w.l[2].y, w.l[2].w, w.l[3].y, w.l[3].w, w.l[4].y, w.l[4].w, up, mid, dn,
                                   mask_l, mask_r);
}
// Sum over each 32-lane half of the wavefront without touching LDS: inclusive scan inside the 16-lane rows
// (row_shr 1,2,4,8), then row 0 -> row 1 and row 2 -> row 3 (row_bcast:15).  Lanes 31 and 63 hold the totals.
// (dpp_shift_add<CTRL, ROW_MASK>: kws_mfcc_dev.h)
__device__ __forceinline__ float half_wave_sum_to_last_lane(float v) {
    v = dpp_shift_add<0x111, 0xf>(v);  // row_shr:1
    v = dpp_shift_add<0x112, 0xf>(v);  // row_shr:2
    v = dpp_shift_add<0x114, 0xf>(v);  // row_shr:4
    v = dpp_shift_add<0x118, 0xf>(v);  // row_shr:8
    v = dpp_shift_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    return v;
}

// Depthwise table [32 channel pairs][24] and pointwise bias [64] of block n (1..4) go to LDS buffer (n-1)&1 in two
// halves so the global-memory latency hides under a whole phase: fetch() issues the loads into three
// registers at the start of the previous phase, store() writes them to LDS after that phase's units.
struct BlockTables {
    float d0, d1, b;
};
__device__ __forceinline__ void fetch_block_tables(const DscnnWeights& w, int n, int tid, BlockTables& r) {
    const float* src = w.dw_w + (n - 1) * CH * 12;
    r.d0 = tid < CH * 12 ? src[tid] : 0.f;
    r.d1 = NT + tid < CH * 12 ? src[NT + tid] : 0.f;
    r.b = tid < CH ? w.pw_b[(n - 1) * CH + tid] : 0.f;
}
// s_dww / s_dwb / s_pwb (f16-pair arithmetic; 1 otherwise): the block's activations are kept in LDS scaled by per-clip powers
// of two and its depthwise OUTPUT is wanted in the operand units 2^ky of the matrix instructions (below 2^15): the depthwise
// weights carry the factor 2^(ky - input units), the depthwise bias 2^ky, so the stencil's result needs no scaling before it is
// split (powers of two: the same bits as scaling afterwards); the pointwise bias (= accumulator seed and ring value) is stored
// in the units of the block's output.
__device__ __forceinline__ void store_block_tables(float* lds, int n, int tid, const BlockTables& r, float s_dwb = 1.f, float s_pwb = 1.f,
                                                   float s_dww = 1.f) {
    float* dwtab = lds + OFF_DWTAB + ((n - 1) & 1) * 768;
    if (tid < CH * 12) dwtab[tid] = r.d0 * ((tid % 24) >> 1 == 9 ? s_dwb : s_dww);  // (pair-interleaved rows of 24: the biases at 18, 19)
    if (NT + tid < CH * 12) dwtab[NT + tid] = r.d1 * (((NT + tid) % 24) >> 1 == 9 ? s_dwb : s_dww);
    if (tid < CH) lds[OFF_PWB + ((n - 1) & 1) * 64 + tid] = r.b * s_pwb;
}
// f16-pair arithmetic: what a stage needs to know about the clip's scales (all powers of two)
struct PairCtx {
    float s_dww = 1.f;      // next block's depthwise weight factor = its operand scale over this block's output units
    float s_dwb = 1.f;      // next block's depthwise bias factor = its operand scale
    float s_pwb = 1.f;      // next block's pointwise bias factor = the next block's output units
    float inv_out = 1.f;    // block 4: pooled sums back to true units
};
// Units of a stage: its operand exponent ky (operand * 2^ky < 2^15) plus the layer's weight exponent.  The stage's stored
// output -- bias included, which the operand bound knows nothing about -- must stay a finite float in those units, and every
// factor derived from them a normal one: 2^sg * bz < 2^100 (bz: bound on the stage's output in true units) and sg <= 120,
// enforced by LOWERING the operand scale (always safe; it binds only for bias-dominated or vanishing stages).
__device__ __forceinline__ void cap_units(int& ky, int& sg, int k_w, float bz) {
    sg = ky + k_w;
    const int eb = (int)((__builtin_bit_cast(uint32_t, bz) >> 23) & 0xffu) - 126;  // bz < 2^eb
    int limit = 100 - eb;
    limit = limit > 120 ? 120 : limit;
    if (sg > limit) {
        ky -= sg - limit;
        sg = limit;
    }
}
// wavefront maximum of non-negative values -> per-wavefront slot (read by everyone after the stage's barrier).  DPP row
// scans, no LDS round trips: six dependent ds_bpermute exchanges sat at the end of every wavefront's stage, in front of the barrier.
__device__ __forceinline__ void publish_wave_max(float* lds, int set, int wv, int lane, float mx) {
    auto step = [](float m, auto ctrl, auto row_mask) {
        return fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, m), __builtin_bit_cast(int, m), decltype(ctrl)::value,
                                                                              decltype(row_mask)::value, 0xf, false)));
    };
    mx = step(mx, std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xf>{});  // row_shr:1
    mx = step(mx, std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xf>{});  // row_shr:2
    mx = step(mx, std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xf>{});  // row_shr:4
    mx = step(mx, std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xf>{});  // row_shr:8
    mx = step(mx, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});  // row_bcast:15 into rows 1, 3
    mx = step(mx, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});  // row_bcast:31 into rows 2, 3
    if (lane == 63) lds[OFF_WMAX + set * NW + wv] = mx;
}
// the maximum over n_sets consecutive sets (NW values each): two 16-byte reads per set
__device__ __forceinline__ float read_stage_max(const float* lds, int set0, int n_sets) {
    static_assert(NW % 4 == 0 && OFF_WMAX % 4 == 0, "the stage maxima are read as float4s");
    float m = 0.f;
    const float4* q = reinterpret_cast<const float4*>(lds + OFF_WMAX + set0 * NW);
    for (int i = 0; i < (NW / 4) * n_sets; ++i) {
        const float4 v = q[i];
        m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    }
    return m;
}

// Pointwise weights of the running block as MFMA A operands.
//   f32 path   (32x32x2 f32):   wa[ct][s] = W[cout = ct*32 + (l&31)][cin = 16(s>>3) + 8(l>>5) + (s&7)], held for the
//     whole block (the K order of the f32 MFMA steps is free; this one is the lane -> channel walk of the split
//     path, so every variant shares the depthwise stage).
//   split path (32x32x16 bf16): piece p (0 hi, 1 mid, 2 lo) of W[cout = ct*32 + (l&31)][cin = 16m + 8(l>>5) + j],
//     j = 0..7 -- eight bf16 per lane and (ct, m, p), pre-split on the host (exactly: hi + mid + lo == W).  Only
//     two k-blocks m are in registers at a time: ring[m & 1] is fetched one k-block ahead from global memory
//     (L1/L2-resident; the same bytes per block as the f32 path loads), which frees 48 registers.
// NP pieces per operand: 3 = bf16 hi/mid/lo (modes 4, 6), 2 = f16 pair (mode 5)
template <int NP>
struct PwRing {
    uintx4 ring[2][2][NP];  // [ring slot][channel tile][piece] of one k-block
};
struct PwRegsF32 {
    float wa[2][32];
};
template <int MODE>
using PwOperands = std::conditional_t<(MODE >= 4), PwRing<(MODE == 5 ? 2 : 3)>, PwRegsF32>;
__device__ __forceinline__ void load_pointwise(const DscnnWeights& w, int n, int lane, PwRegsF32& o) {
    const float* pw = w.pw_w + (n - 1) * CH * CH + 8 * (lane >> 5) * CH + (lane & 31);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int s = 0; s < 32; ++s) o.wa[ct][s] = pw[(16 * (s >> 3) + (s & 7)) * CH + ct * 32];
}
template <int NP>
__device__ __forceinline__ void load_afrag(const DscnnWeights& w, int n, int m, int lane, uintx4 (&f)[2][NP]) {
    const uintx4* src = reinterpret_cast<const uintx4*>(NP == 2 ? w.pw_pair : w.pw_split) + (size_t)(n - 1) * (2 * 4 * NP * 64) + lane;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int p = 0; p < NP; ++p) f[ct][p] = src[((ct * 4 + m) * NP + p) * 64];
}
// first operands of block n: the whole block (f32) or its k-block 0 (split)
__device__ __forceinline__ void load_block_head(const DscnnWeights& w, int n, int lane, PwRegsF32& o) { load_pointwise(w, n, lane, o); }
template <int NP>
__device__ __forceinline__ void load_block_head(const DscnnWeights& w, int n, int lane, PwRing<NP>& o) { load_afrag(w, n, 0, lane, o.ring[0]); }

// ------------------------------------------------------------------------------------------------
// conv1: D[cout][pos] = sum_k W[cout][k] * im2col[k][pos], k = kh*10 + kw, as 50 MFMA k-steps.
template <bool MFMA>
__device__ __forceinline__ void conv1_phase(const DscnnWeights& w, float* lds, int tid, const float (&a)[50]) {
    const float* featp = lds + OFF_FEAT;
    float* z0 = lds + OFF_Z0;
    if constexpr (MFMA) {
        const int lane = tid & 63, wv = tid >> 6, half = lane >> 5, col = lane & 31;
        const int ct = wv & 1;  // units u = wv, wv + NW share the output-channel tile (NW is even)
        for (int u = wv; u < 10; u += NW) {
            const int pt = u >> 1;
            const int pos = pt * 32 + col;
            const int posc = pos < P0 ? pos : P0 - 1;
            const int oh = posc / C1_W, ow = posc % C1_W;
            const float* base = featp + (2 * oh) * FEAT_W + 2 * ow + half;
            floatx16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < 50; ++s) {
                const float b = base[((2 * s) / 10) * FEAT_W + (2 * s) % 10];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b, acc, 0, 0, 0);
            }
            if (pos < P0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = ct * 32 + row_of(r, half);
                    z0[pidx(co, pos, P0 + 2)] = relu(acc[r] + w.c1_b[co]);
                }
            }
        }
    } else {
        for (int idx = tid; idx < CH * P0; idx += NT) {
            const int co = idx / P0, pos = idx % P0;
            const int oh = pos / C1_W, ow = pos % C1_W;
            float acc = w.c1_b[co];
            for (int kh = 0; kh < C1_K; ++kh)
                for (int kw = 0; kw < C1_K; ++kw)
                    acc = fmaf(w.c1_w[(kh * C1_K + kw) * CH + co], featp[(2 * oh + kh) * FEAT_W + 2 * ow + kw], acc);
            z0[pidx(co, pos, P0 + 2)] = relu(acc);
        }
    }
    if (tid < CH) {  // extra slots of the conv1 planes: no ring in block 1, slot P+1 is the zero pad
        z0[pidx(tid, P0, P0 + 2)] = 0.f;
        z0[pidx(tid, P0 + 1, P0 + 2)] = 0.f;
    }
}

// conv1 on the bf16 matrix pipe (split path).  K order: the half-wave h takes kernel rows 5h..5h+4, so both
// halves walk the same 56 offsets f = 8kb + j -> (kh%5 = f/10, kw = f%10) (f >= 50: zero weights) and lane
// (col, h) of k-block kb supplies im2col values feat[2oh + 5h + f/10][2ow + f%10], j = 0..7, split into three bf16
// pieces like the pointwise operands.  c1f: the channel tile wv & 1, [kb][piece], loaded at kernel start.
//
// Work split: 141 positions = 5 tiles of 32, two channel tiles each.  As ten (tile, channel tile) units on eight
// wavefronts two wavefronts run two units back to back and every unit gathers and splits its tile's im2col values
// again.  Instead wavefronts 0-3 take tiles 0-3 for BOTH channel tiles (one gather + split feeds twelve MFMAs per
// k-block; the other tile's A fragments stream from L2 through a two-deep ring), wavefronts 4 and 5 take tile 4 for
// one channel tile each, 6 and 7 have no conv1 work: one round, and the busiest SIMD (a dual and a single unit)
// carries the matrix work of three single units but two split streams instead of three.
// p_lo / p_hi: the positions this workgroup computes (the whole map, or the rows of one time tile: see PosRange).
// NP = 3: bf16 hi/mid/lo, six products per k-block.  NP = 2: f16 pairs, three (kws_split_mfma.h); the features are multiplied
// by the clip's scale sx inside the split, the accumulators are in units sig0 = sx * (the layer's weight scale), the bias is
// added in those units and the output is STORED in them (block 1's depthwise bias is scaled to match); mx collects the
// largest stored value of this wavefront.
template <bool DUAL, int NP>
__device__ __forceinline__ void conv1_unit_split(const DscnnWeights& w, const float* featp, float* z0, int ptile, int ct, int lane,
                                                 const uintx4 (&c1f)[7][NP], int p_lo, int p_hi, float sx, float sig0, float& mx) {
    constexpr bool PAIR = NP == 2;
    const int half = lane >> 5, col = lane & 31;
    const int pos = p_lo + ptile * 32 + col;
    const int posc = pos < p_hi ? pos : p_hi - 1;
    const int oh = posc / C1_W, ow = posc % C1_W;
    const float* base = featp + (2 * oh + 5 * half) * FEAT_W + 2 * ow;
    const floatx16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    floatx16 acc = zero, acc2 = zero;  // two chains per channel tile keep the matrix pipe fed
    floatx16 occ = zero, occ2 = zero;  // the other channel tile (DUAL)
    const uintx4* osrc = reinterpret_cast<const uintx4*>(PAIR ? w.c1_pair : w.c1_split) + (size_t)(ct ^ 1) * (7 * NP * 64) + lane;
    uintx4 of[2][NP];                  // its A fragments: k-block kb in of[kb & 1], requested two k-blocks ahead
    auto load_other = [&](int kb) {
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) of[kb & 1][pc] = osrc[(kb * NP + pc) * 64];
    };
    if (DUAL) {
        load_other(0);
        load_other(1);
    }
    float y[2][8];
    auto gather = [&](int kb, float (&dst)[8]) {  // offsets f, f+1 (f even) are neighbours in one row: 8-byte reads
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            const float2 v = *reinterpret_cast<const float2*>(base + ((8 * kb + j) / 10) * FEAT_W + (8 * kb + j) % 10);
            dst[j] = v.x;
            dst[j + 1] = v.y;
        }
    };
    uintx4 bf[2][NP];  // [buffer][piece] B operands: k-block kb multiplies while kb+1 is being split
    gather(0, y[0]);
    gather(1, y[1]);
    if constexpr (PAIR)
        split_pair8(y[0], sx, bf[0][0], bf[0][1]);
    else
        split3(y[0], bf[0][0], bf[0][1], bf[0][NP - 1]);
    gather(2, y[0]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kb = 0; kb < 7; ++kb) {
        const int cur = kb & 1, nxt = cur ^ 1;
        // the piece products of this k-block (per channel tile), smallest first, spread over the next k-block's split
        auto product = [&](int q) {
            // triple: (lo,hi) (hi,lo) (mid,mid) (mid,hi) (hi,mid) (hi,hi); pair: (hi,lo) (lo,hi) (hi,hi)
            const int pa = PAIR ? (q == 1 ? 1 : 0) : (q == 0 ? 2 : (q == 2 || q == 3) ? 1 : 0);
            const int pb = PAIR ? (q == 0 ? 1 : 0) : ((q == 0 || q == 3 || q == 5) ? 0 : (q == 1 ? 2 : 1));
            auto mm = [&](const uintx4& a, const uintx4& b, floatx16 c) {
                if constexpr (PAIR)
                    return mfma_f16(a, b, c);
                else
                    return mfma_bf16(a, b, c);
            };
            if (q & 1)
                acc2 = mm(c1f[kb][pa], bf[cur][pb], acc2);
            else
                acc = mm(c1f[kb][pa], bf[cur][pb], acc);
            __builtin_amdgcn_sched_barrier(0);
            if (DUAL) {
                if (q & 1)
                    occ2 = mm(of[cur][pa], bf[cur][pb], occ2);
                else
                    occ = mm(of[cur][pa], bf[cur][pb], occ);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if constexpr (PAIR) {
            product(0);
            if (kb + 1 < 7) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    uint32_t h, l;
                    split_pair2(y[nxt][2 * i], y[nxt][2 * i + 1], sx, h, l);
                    bf[nxt][0][i] = h;
                    bf[nxt][1][i] = l;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            product(1);
            if (kb + 1 < 7) {
#pragma unroll
                for (int i = 2; i < 4; ++i) {
                    uint32_t h, l;
                    split_pair2(y[nxt][2 * i], y[nxt][2 * i + 1], sx, h, l);
                    bf[nxt][0][i] = h;
                    bf[nxt][1][i] = l;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (kb + 3 < 7) gather(kb + 3, y[nxt]);
            __builtin_amdgcn_sched_barrier(0);
            product(2);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                product(i);
                if (kb + 1 < 7) {
                    const float a0 = y[nxt][2 * i], a1 = y[nxt][2 * i + 1];
                    const float r0 = a0 - top16(a0), r1 = a1 - top16(a1);
                    bf[nxt][0][i] = pack_top16(a0, a1);
                    bf[nxt][1][i] = pack_top16(r0, r1);
                    bf[nxt][NP - 1][i] = pack_top16(r0 - top16(r0), r1 - top16(r1));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            product(4);
            if (kb + 3 < 7) gather(kb + 3, y[nxt]);
            __builtin_amdgcn_sched_barrier(0);
            product(5);
        }
        if (DUAL && kb + 2 < 7) {
            load_other(kb + 2);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    acc += acc2;
    occ += occ2;
    if (pos < p_hi) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {  // accumulator rows r, r+1 are adjacent output channels: one 8-byte store
            const int co = ct * 32 + row_of(r, half);
            const float v0 = relu(fmaf(w.c1_b[co], sig0, acc[r])), v1 = relu(fmaf(w.c1_b[co + 1], sig0, acc[r + 1]));
            *reinterpret_cast<float2*>(z0 + pidx(co, pos, P0 + 2)) = make_float2(v0, v1);
            if constexpr (PAIR) mx = fmaxf(mx, fmaxf(v0, v1));
            if (DUAL) {
                const int oo = (ct ^ 1) * 32 + row_of(r, half);
                const float u0 = relu(fmaf(w.c1_b[oo], sig0, occ[r])), u1 = relu(fmaf(w.c1_b[oo + 1], sig0, occ[r + 1]));
                *reinterpret_cast<float2*>(z0 + pidx(oo, pos, P0 + 2)) = make_float2(u0, u1);
                if constexpr (PAIR) mx = fmaxf(mx, fmaxf(u0, u1));
            }
        }
    }
}

// ---- conv1 on f16 pairs from PRE-SPLIT WINDOWS (PAIR only) ---------------------------------------------------------------
// Gathered and split per unit, conv1's B operand cost four 8-byte LDS reads and twelve VALU instructions per k-block in front
// of every three (six) MFMAs, and the phase ran at the latency of that chain.  Instead the scaled, zero-padded feature map is
// split ONCE per clip into LDS (the region block 1's output takes later):
//   W8[piece][row 0..102][s 0..2]  16 bytes: the eight features (row, 2s .. 2s + 7) as f16
//   P2[piece][row][s]               4 bytes: the two features (row, 2s + 8), (row, 2s + 9)
// and K is ordered to match: half-wave h takes kernel rows 5h .. 5h + 4; k-block kb < 5 = kernel row 5h + kb, taps kw 0..7 --
// one aligned ds_read_b128 per piece; k-block 5 = taps kw 8, 9 of kernel rows 5h .. 5h + 3 (four dwords per piece), k-block 6 =
// taps kw 8, 9 of kernel row 5h + 4 and six zeros.  (c1_pair is laid out in this order by kws_load_dscnn.)
constexpr int C1W_ROWS = FEAT_H;                                   // 103 padded feature rows
constexpr int OFF_C1W8 = OFF_Z1;                                   // floats; [2][103][3][4 dwords]
constexpr int OFF_C1P2 = OFF_C1W8 + 2 * C1W_ROWS * 3 * 4;          // [2][103][3] dwords
static_assert(OFF_C1P2 + 2 * C1W_ROWS * 3 <= OFF_Z0, "conv1's operand windows live where block 1's output goes later");
static_assert(2 * (C1_W - 1) + 9 < FEAT_W && 2 * (C1_H - 1) + 9 < FEAT_H, "window reach inside the padded map");

__device__ __forceinline__ void conv1_build_windows(float* lds, int tid, float sx) {
    const float* featp = lds + OFF_FEAT;
    uint32_t* w8 = reinterpret_cast<uint32_t*>(lds + OFF_C1W8);
    uint32_t* p2 = reinterpret_cast<uint32_t*>(lds + OFF_C1P2);
    for (int i = tid; i < C1W_ROWS * 3; i += NT) {
        const float* src = featp + (i / 3) * FEAT_W + 2 * (i % 3);
        const float y[8] = {src[0], src[1], src[2], src[3], src[4], src[5], src[6], src[7]};
        uintx4 hi, lo;
        split_pair8(y, sx, hi, lo);
        *reinterpret_cast<uintx4*>(w8 + i * 4) = hi;
        *reinterpret_cast<uintx4*>(w8 + (C1W_ROWS * 3 + i) * 4) = lo;
        uint32_t h, l;
        split_pair2(src[8], src[9], sx, h, l);
        p2[i] = h;
        p2[C1W_ROWS * 3 + i] = l;
    }
}

template <bool DUAL>
__device__ __forceinline__ void conv1_unit_pairwin(const DscnnWeights& w, const float* lds, float* z0, int ptile, int ct, int lane,
                                                   const uintx4 (&c1f)[7][2], int p_lo, int p_hi, float sig0, float& mx) {
    const int half = lane >> 5, col = lane & 31;
    const int pos = p_lo + ptile * 32 + col;
    const int posc = pos < p_hi ? pos : p_hi - 1;
    const int oh = posc / C1_W, ow = posc % C1_W;
    const int wi = (2 * oh + 5 * half) * 3 + ow;  // window of kernel row 5h at this position; kernel row 5h + i: + 3i
    const uintx4* w8 = reinterpret_cast<const uintx4*>(lds + OFF_C1W8) + wi;
    const uint32_t* p2 = reinterpret_cast<const uint32_t*>(lds + OFF_C1P2) + wi;
    const floatx16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    floatx16 acc = zero, acc2 = zero, occ = zero, occ2 = zero;  // two chains per channel tile; occ*: the other channel tile (DUAL)
    const uintx4* osrc = reinterpret_cast<const uintx4*>(w.c1_pair) + (size_t)(ct ^ 1) * (7 * 2 * 64) + lane;
    // the other tile's A fragments, all seven k-blocks requested up front (L2 hits, but ~600 cycles away: with the operand split
    // gone a k-block is too short to hide them two k-blocks ahead; the registers are free in this phase)
    uintx4 of[DUAL ? 7 : 1][2];
    if (DUAL) {
#pragma unroll
        for (int kb = 0; kb < 7; ++kb) {
            of[kb][0] = osrc[(kb * 2 + 0) * 64];
            of[kb][1] = osrc[(kb * 2 + 1) * 64];
        }
    }
    // the biases of this lane's accumulator rows (rows 4q .. 4q+3 = channels 8q + 4 half + 0..3: one float4 each), requested
    // now: read in the epilogue they were an L2 round trip at the end of every unit
    float4 cb[4], ob[DUAL ? 4 : 1];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        cb[q] = *reinterpret_cast<const float4*>(w.c1_b + ct * 32 + 8 * q + 4 * half);
        if (DUAL) ob[DUAL ? q : 0] = *reinterpret_cast<const float4*>(w.c1_b + (ct ^ 1) * 32 + 8 * q + 4 * half);
    }
    uintx4 bq[3][2];  // [k-block mod 3][piece], fetched two k-blocks ahead
    auto b_load = [&](int kb, uintx4 (&d)[2]) {
        if (kb < 5) {
            d[0] = w8[kb * 3];
            d[1] = w8[kb * 3 + C1W_ROWS * 3];
        } else if (kb == 5) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                d[0][i] = p2[i * 3];
                d[1][i] = p2[i * 3 + C1W_ROWS * 3];
            }
        } else {
            d[0] = uintx4{p2[4 * 3], 0u, 0u, 0u};
            d[1] = uintx4{p2[4 * 3 + C1W_ROWS * 3], 0u, 0u, 0u};
        }
    };
    b_load(0, bq[0]);
    b_load(1, bq[1]);
#pragma unroll
    for (int kb = 0; kb < 7; ++kb) {
        if (kb + 2 < 7) b_load(kb + 2, bq[(kb + 2) % 3]);
        const uintx4 &bh = bq[kb % 3][0], &bl = bq[kb % 3][1];
        // (hi, lo) (lo, hi) (hi, hi), the two channel tiles interleaved
        acc2 = mfma_f16(c1f[kb][0], bl, acc2);
        if (DUAL) occ2 = mfma_f16(of[DUAL ? kb : 0][0], bl, occ2);
        acc = mfma_f16(c1f[kb][1], bh, acc);
        if (DUAL) occ = mfma_f16(of[DUAL ? kb : 0][1], bh, occ);
        acc2 = mfma_f16(c1f[kb][0], bh, acc2);
        if (DUAL) occ2 = mfma_f16(of[DUAL ? kb : 0][0], bh, occ2);
    }
    acc += acc2;
    occ += occ2;
    if (pos < p_hi) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {  // accumulator rows r, r+1 are adjacent output channels: one 8-byte store
            const int co = ct * 32 + row_of(r, half);
            const float4 b4 = cb[r >> 2];
            const float b0 = (r & 2) ? b4.z : b4.x, b1 = (r & 2) ? b4.w : b4.y;
            const float v0 = relu(fmaf(b0, sig0, acc[r])), v1 = relu(fmaf(b1, sig0, acc[r + 1]));
            *reinterpret_cast<float2*>(z0 + pidx(co, pos, P0 + 2)) = make_float2(v0, v1);
            mx = fmaxf(mx, fmaxf(v0, v1));
            if (DUAL) {
                const int oo = (ct ^ 1) * 32 + row_of(r, half);
                const float4 o4 = ob[DUAL ? (r >> 2) : 0];
                const float c0 = (r & 2) ? o4.z : o4.x, c1 = (r & 2) ? o4.w : o4.y;
                const float u0 = relu(fmaf(c0, sig0, occ[r])), u1 = relu(fmaf(c1, sig0, occ[r + 1]));
                *reinterpret_cast<float2*>(z0 + pidx(oo, pos, P0 + 2)) = make_float2(u0, u1);
                mx = fmaxf(mx, fmaxf(u0, u1));
            }
        }
    }
}

// Rows [lo, hi) of a map, as flattened positions [lo * W, hi * W): what one workgroup of a time-tile cluster computes of a
// stage (the streaming push at few streams, see kws_dscnn_fwd_kernel).  The full map when the workgroup owns the clip.
struct PosRange {
    int lo, hi;
};

template <bool RANGED, int NP>
__device__ __forceinline__ void conv1_phase_split(const DscnnWeights& w, float* lds, int tid, const uintx4 (&c1f)[7][NP],
                                                  PosRange rg, float sx = 1.f, float sig0 = 1.f) {
    static_assert(P0 > 4 * 32 && P0 <= 5 * 32 && NW >= 6, "conv1 work split: four dual tiles + one tile in two halves");
    const float* featp = lds + OFF_FEAT;
    float* z0 = lds + OFF_Z0;
    const int lane = tid & 63, wv = tid >> 6;
    float mx = 0.f;
    if constexpr (NP == 2) {  // f16 pairs: operands from the pre-split windows (built by the caller, behind a barrier)
        if constexpr (RANGED) {
            const int n_pt = (rg.hi - rg.lo + 31) / 32;
            for (int u = wv; u < 2 * n_pt; u += NW) conv1_unit_pairwin<false>(w, lds, z0, u >> 1, u & 1, lane, c1f, rg.lo, rg.hi, sig0, mx);
        } else if (wv < 4)
            conv1_unit_pairwin<true>(w, lds, z0, wv, wv & 1, lane, c1f, 0, P0, sig0, mx);
        else if (wv < 6)
            conv1_unit_pairwin<false>(w, lds, z0, 4, wv & 1, lane, c1f, 0, P0, sig0, mx);
    } else if constexpr (RANGED) {
        // a time tile holds at most 4 position tiles of 32: one (tile, channel tile) unit per wavefront, one round -- the
        // shortest critical path (a dual unit carries twice the matrix work); c1f holds channel tile wv & 1
        const int n_pt = (rg.hi - rg.lo + 31) / 32;
        for (int u = wv; u < 2 * n_pt; u += NW) conv1_unit_split<false, NP>(w, featp, z0, u >> 1, u & 1, lane, c1f, rg.lo, rg.hi, sx, sig0, mx);
    } else if (wv < 4)
        conv1_unit_split<true, NP>(w, featp, z0, wv, wv & 1, lane, c1f, 0, P0, sx, sig0, mx);
    else if (wv < 6)
        conv1_unit_split<false, NP>(w, featp, z0, 4, wv & 1, lane, c1f, 0, P0, sx, sig0, mx);
    if constexpr (NP == 2) publish_wave_max(lds, 1, wv, lane, mx);
    if (tid < CH) {  // extra slots of the conv1 planes: no ring in block 1, slot P+1 is the zero pad
        z0[pidx(tid, P0, P0 + 2)] = 0.f;
        z0[pidx(tid, P0 + 1, P0 + 2)] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------------
// A quarter of the leftover tile of block N (see Leftover): k-block M (input channels 16M .. 16M+15) of tile Leftover<N>::TILE
// for both output-channel tiles.  Eight stencil steps, one split, twelve MFMAs, the raw partial sums (no bias, no ReLU) of the
// tile's valid columns to part[M][cout][position in tile].  af: the pre-split weights of k-block M (requested long before).
template <int N, int NP>
__device__ __forceinline__ void leftover_partial_unit(float* lds, int lane, int M, const uintx4 (&af)[2][NP]) {
    using G = Blk<N>;
    using L = Leftover<N>;
    const int half = lane >> 5, col = lane & 31;
    const float* dwtab = lds + OFF_DWTAB + G::BUF * 768;
    const float4* dwt4 = reinterpret_cast<const float4*>(dwtab) + half * 24;
    const int pos = L::P0T - 1 + col;
    const bool valid = col >= 1 && col <= TW && pos < G::POUT;
    const int posc = pos < G::POUT ? pos : G::POUT - 1;  // (pos >= P0T - 1 >= 0)
    const int h = posc / G::W, x = posc % G::W;
    const float mask_l = x > 0 ? 1.f : 0.f, mask_r = x < G::W - 1 ? 1.f : 0.f;
    int ta[3];  // own-column tap addresses (rows h-1, h, h+1) of channel pair 8 * half, as float indices into lds
#pragma unroll
    for (int dh = -1; dh <= 1; ++dh) {
        const int o = G::RING ? 1 : 0;
        const int hh = h + dh - o, xx = x - o;
        const bool inside = (unsigned)hh < (unsigned)G::HI && (unsigned)xx < (unsigned)G::WI;
        const bool in_map = (unsigned)(h + dh) < (unsigned)G::H;
        const int a = inside ? hh * G::WI + xx : ((G::RING && in_map) ? G::PIN : G::PIN + 1);
        ta[dh + 1] = G::OFF_IN + pidx(half * 8, a, G::SIN);
    }
    float y[8];
#pragma unroll
    for (int j = 0; j < 8; j += 2) {  // channels (cs, cs + 1) = 16M + j, + 1 (+ 8 * half through the addresses)
        const int cs = 16 * M + j;
        const int o = cs * G::SIN;    // pair-interleaved planes: channel pair cs / 2 starts at (cs / 2) * 2 * SIN
        const float2 up = *reinterpret_cast<const float2*>(lds + ta[0] + o);
        const float2 mid = *reinterpret_cast<const float2*>(lds + ta[1] + o);
        const float2 dn = *reinterpret_cast<const float2*>(lds + ta[2] + o);
        DwPair wp;
#pragma unroll
        for (int i = 0; i < 5; ++i) wp.l[i] = dwt4[(cs >> 1) * 6 + i];
        y[j] = stencil3x3_of_pair<false, 0>(wp, up.x, mid.x, dn.x, mask_l, mask_r);
        y[j + 1] = stencil3x3_of_pair<false, 1>(wp, up.y, mid.y, dn.y, mask_l, mask_r);
    }
    floatx16 acc0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, acc1 = acc0;
    if constexpr (NP == 2) {  // f16 pair: (hi,lo) (lo,hi) (hi,hi)
        uintx4 bh, bl;
        split_pair8_scaled(y, bh, bl);  // (ends with the two wait states a matrix operand needs)
        acc0 = mfma_f16(af[0][0], bl, acc0);
        acc1 = mfma_f16(af[1][0], bl, acc1);
        acc0 = mfma_f16(af[0][1], bh, acc0);
        acc1 = mfma_f16(af[1][1], bh, acc1);
        acc0 = mfma_f16(af[0][0], bh, acc0);
        acc1 = mfma_f16(af[1][0], bh, acc1);
    } else {
        uintx4 bh, bm, bl;
        split3(y, bh, bm, bl);
#pragma unroll
        for (int q = 0; q < 6; ++q) {  // the six piece products, smallest first
            const int pa = q == 0 ? 2 : (q == 2 || q == 3) ? 1 : 0;
            const uintx4& b = (q == 0 || q == 3 || q == 5) ? bh : (q == 1 ? bl : bm);
            acc0 = mfma_bf16(af[0][pa], b, acc0);
            acc1 = mfma_bf16(af[1][pa], b, acc1);
        }
    }
    float* part = lds + L::OFF_PART + M * (CH * L::NP);
    if (valid) {  // (plain stores of the accumulators: the compiler waits out the matrix-core write itself)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            part[row_of(r, half) * L::NP + (col - 1)] = acc0[r];
            part[(32 + row_of(r, half)) * L::NP + (col - 1)] = acc1[r];
        }
    }
}

// After the block's barrier: the leftover tile's output = relu(bias + the four k-block partials, added in a fixed order).
template <int N, bool PAIR = false>
__device__ __forceinline__ void leftover_combine(float* lds, int tid) {
    using G = Blk<N>;
    using L = Leftover<N>;
    const float* part = lds + L::OFF_PART;
    const float* pwb = lds + OFF_PWB + G::BUF * 64;
    float* zout = lds + G::OFF_OUT;
    float mx = 0.f;
    for (int i = tid; i < CH * L::NP; i += NT) {
        const int co = i / L::NP, j = i - co * L::NP;
        const float s = (part[i] + part[CH * L::NP + i]) + (part[2 * CH * L::NP + i] + part[3 * CH * L::NP + i]);
        const float v = relu(s + pwb[co]);
        zout[pidx(co, L::P0T + j, G::SOUT)] = v;
        mx = fmaxf(mx, v);
    }
    if constexpr (PAIR) publish_wave_max(lds, N == 1 ? 3 : 1, tid >> 6, tid & 63, mx);  // (pwb is stored in the accumulators' units)
}

// ------------------------------------------------------------------------------------------------
// One depthwise-separable block.  pwo: pointwise operands of THIS block on entry (f32: all of them; split:
// k-block 0 in ring[0]); on exit (N < 4) the loads of the next block's have been issued into it, so they fly
// across the barrier.
// MODE: 0 = pointwise GEMM on the VALU (cross-check of the MFMA operand mappings), 1 = f32 MFMA,
// 4 = split-bf16 MFMA (product path), 2 / 3 = timing ablations of mode 1 (matrix core only / stencil only;
// wrong results by construction).
// act4 (diagnostics instantiation only, block 4): global [64][53*9] that receives the block's output, which the product
// path never stores (it is pooled in registers).
// RANGED: only the positions rg.lo .. rg.hi - 1 (whole rows) are computed -- one time tile of a workgroup cluster.
template <int N, int MODE, bool RANGED = false>
__device__ __forceinline__ void block_phase(const DscnnWeights& w, float* lds, int tid, PwOperands<MODE>& pwo,
                                            float* __restrict__ act4 = nullptr, PosRange rg = PosRange{0, 0}, PairCtx pc = PairCtx{}) {
    using G = Blk<N>;
    constexpr bool PAIR = MODE == 5;          // f16 pairs: three products per k-block, activations in per-clip scaled units
    constexpr int NP = PAIR ? 2 : 3;
    constexpr int NPROD = PAIR ? 3 : 6;
    // the leftover tile of blocks 1 / 2 is K-split over four wavefronts (product paths on whole maps only)
    constexpr bool KSL = Leftover<N>::HAS && (MODE == 4 || MODE == 5) && !RANGED;
    const int p_lo = RANGED ? rg.lo : 0, p_hi = RANGED ? rg.hi : G::POUT;
#ifdef KWS_X_DSCNN_SKIP_LEFTOVER  // timing experiment (wrong results): block 2 without its ninth tile, the upper bound of what
                                  // spreading that tile over idle wavefronts could win
    const int n_tiles = RANGED ? (p_hi - p_lo + TW - 1) / TW : (N == 2 ? 8 : G::TILES);
#else
    const int n_tiles = RANGED ? (p_hi - p_lo + TW - 1) / TW : (KSL ? Leftover<N>::TILE : G::TILES);
#endif
    constexpr bool MFMA = MODE != 0;
    constexpr bool SPLIT = MODE >= 4;  // input channel of step s: 16(s>>3) + 8*half + (s&7) instead of 2s + half
    // timing ablation of the split path (wrong results by construction): 6 = split + matrix core without the stencil
    constexpr bool NO_STENCIL = MODE == 6;
    const int lane = tid & 63, wv = tid >> 6, half = lane >> 5, col = lane & 31;
    float* zout = lds + G::OFF_OUT;
    const float* dwtab = lds + OFF_DWTAB + G::BUF * 768;
    const float* pwb = lds + OFF_PWB + G::BUF * 64;
    float* poolbuf = lds + OFF_POOLBUF;

    // The other table buffer is idle during this block: the next block's tables are fetched now and
    // stored after the units.  Ring and zero slots of the output planes.  No barrier needed before the
    // units: everything they read was staged during the previous phase.
    BlockTables next_tables;
    if constexpr (N < 4) {
        fetch_block_tables(w, N + 1, tid, next_tables);
        if (tid < CH) {
            zout[pidx(tid, G::POUT, G::SOUT)] = relu(pwb[tid]);
            zout[pidx(tid, G::POUT + 1, G::SOUT)] = 0.f;
        }
    } else if (!MFMA) {
        poolbuf[wv * CH + lane] = 0.f;  // the VALU path accumulates into its wave's scratch row
    }

    // accumulator rows 4q..4q+3 of tile ct are output channels ct*32 + 8q + 4*half + (0..3): one float4
    const float4* bias4 = reinterpret_cast<const float4*>(pwb) + half;
    // lane (column, half) walks the input channels 16m + 8*half + j (m = 0..3, j = 0..7) in 32 steps s = 8m + j
    const float4* dwt4 = reinterpret_cast<const float4*>(dwtab) + half * 24;  // (channel pair 4 * half; six float4 per pair, five of them read)
    float psum[2][16];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) psum[ct][r] = 0.f;
    float stage_max = 0.f;  // PAIR, blocks 1 and 2: this wavefront's largest stored output

    for (int t = wv; t < n_tiles; t += NW) {
        // column j of the tile is output position p_lo + t*TW - 1 + j: columns 0 and 31 are halo.  A halo column past the
        // range's ends is clamped into it: a range ends on a row end, where the neighbour's contribution is masked anyway,
        // and the clamp keeps the lane's own reads on rows this workgroup has computed.
        const int pos = p_lo + t * TW - 1 + col;
        const bool valid = col >= 1 && col <= TW && pos < p_hi;
        const int posc = pos < p_lo ? p_lo : (pos < p_hi ? pos : p_hi - 1);
        const int h = posc / G::W, x = posc % G::W;
        const float mask_l = x > 0 ? 1.f : 0.f, mask_r = x < G::W - 1 ? 1.f : 0.f;
        // own-column tap addresses (rows h-1, h, h+1), as float indices into lds, for channel pairs
        // 0..15 (lo) and 16..31 (hi): two bases keep every ds_read inside the 64 KiB immediate window;
        // the empty asm stops the compiler from re-deriving one base register per step.
        int tlo[3], thi[3];
#pragma unroll
        for (int dh = -1; dh <= 1; ++dh) {
            const int o = G::RING ? 1 : 0;
            const int hh = h + dh - o, xx = x - o;
            const bool inside = (unsigned)hh < (unsigned)G::HI && (unsigned)xx < (unsigned)G::WI;
            const bool in_map = (unsigned)(h + dh) < (unsigned)G::H;
            const int a = inside ? hh * G::WI + xx : ((G::RING && in_map) ? G::PIN : G::PIN + 1);
            tlo[dh + 1] = G::OFF_IN + pidx(half * 8, a, G::SIN);
            thi[dh + 1] = tlo[dh + 1] + 32 * G::SIN;
            asm volatile("" : "+v"(tlo[dh + 1]));
            asm volatile("" : "+v"(thi[dh + 1]));
        }

        auto cs_of = [](int s) { return 16 * (s >> 3) + (s & 7); };  // channel of step s minus the half's offset 8*half
        // the own-column inputs of channels (cs, cs+1), cs even, in three 8-byte reads (pair-interleaved planes)
        struct TapPair {
            float2 up, mid, dn;
        };
        auto tap_pair_load = [&](int sp, TapPair& tp) {  // sp: pair of steps (2sp, 2sp+1)
            const int cs = cs_of(2 * sp);
            const int* ta = cs < 32 ? tlo : thi;
            const int o = (cs & 31) * G::SIN;
            if constexpr (!NO_STENCIL) {
                tp.up = *reinterpret_cast<const float2*>(lds + ta[0] + o);
                tp.dn = *reinterpret_cast<const float2*>(lds + ta[2] + o);
            }
            tp.mid = *reinterpret_cast<const float2*>(lds + ta[1] + o);
        };
        // the depthwise weights of the step pair's two channels: five 16-byte reads (DwPair)
        auto wts_pair_load = [&](int sp, DwPair& wp) {
            const int cs = cs_of(2 * sp);
#pragma unroll
            for (int i = 0; i < 5; ++i) wp.l[i] = dwt4[(cs >> 1) * 6 + i];
        };
        // depthwise 3x3 (+bias) of step 2sp + odd at this lane's column -> one MFMA B operand element
        auto dw_eval = [&](const DwPair& wp, const TapPair& tq, auto odd) -> float {
            constexpr int E = decltype(odd)::value;
            return stencil3x3_of_pair<(MFMA && !SPLIT), E>(wp, E ? tq.up.y : tq.up.x, E ? tq.mid.y : tq.mid.x, E ? tq.dn.y : tq.dn.x, mask_l, mask_r);
        };

        if constexpr (MFMA) {
            floatx16 acc0, acc1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 b0 = bias4[2 * q], b1 = bias4[8 + 2 * q];
                acc0[4 * q + 0] = b0.x; acc0[4 * q + 1] = b0.y; acc0[4 * q + 2] = b0.z; acc0[4 * q + 3] = b0.w;
                acc1[4 * q + 0] = b1.x; acc1[4 * q + 1] = b1.y; acc1[4 * q + 2] = b1.z; acc1[4 * q + 3] = b1.w;
            }
            // software pipeline, two steps deep: reads of step s+2 are issued before step s is evaluated
            DwPair wq0, wq1;   // depthwise weights of step pairs, two pairs in flight
            TapPair tq0, tq1;  // inputs of step pairs, two pairs in flight
            if constexpr (!NO_STENCIL) {
                wts_pair_load(0, wq0);
                wts_pair_load(1, wq1);
            }
            tap_pair_load(0, tq0);
            tap_pair_load(1, tq1);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (SPLIT) {
                // eight depthwise outputs fill one k-block of 16 input channels (8 per half-wave); they are split
                // into three bf16 pieces and multiplied with the pre-split weights: 6 products x 2 channel tiles.
                // The bf16 matrix pipe runs beside the VALU, so the stencil of the next k-block overlaps them.
                // The 12 MFMAs of k-block m are issued one per half step while the VALU evaluates the stencil of
                // k-block m+1 (sched_barrier pins that interleave; left alone, the compiler issues them back to back
                // and the wavefront sits behind the busy matrix pipe).  Smallest products first.
                float y[8];
                uintx4 bh, bm, bl;
                auto product = [&](int ct, int m, int q) {  // q-th of the piece products of k-block m, channel tile ct
                    // triple: (lo,hi) (hi,lo) (mid,mid) (mid,hi) (hi,mid) (hi,hi); pair: (hi,lo) (lo,hi) (hi,hi)
                    const int pa = PAIR ? (q == 1 ? 1 : 0) : (q == 0 ? 2 : (q == 2 || q == 3) ? 1 : 0);
                    const uintx4& b = PAIR ? (q == 0 ? bl : bh) : ((q == 0 || q == 3 || q == 5) ? bh : (q == 1 ? bl : bm));
                    if constexpr (PAIR) {
                        if (ct == 0)
                            acc0 = mfma_f16(pwo.ring[m & 1][0][pa], b, acc0);
                        else
                            acc1 = mfma_f16(pwo.ring[m & 1][1][pa], b, acc1);
                    } else {
                        if (ct == 0)
                            acc0 = mfma_bf16(pwo.ring[m & 1][0][pa], b, acc0);
                        else
                            acc1 = mfma_bf16(pwo.ring[m & 1][1][pa], b, acc1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                };
#pragma unroll
                for (int s = 0; s < 32; ++s) {
                    const int m = s >> 3, j = s & 7;
                    DwPair& wp = (s & 2) ? wq1 : wq0;
                    TapPair& tq = (s & 2) ? tq1 : tq0;  // step pair s >> 1
                    const bool feed = m > 0 && j < NPROD;
                    if (feed) product(0, m - 1, j);
                    if constexpr (NO_STENCIL)
                        y[j] = (s & 1) ? tq.mid.y : tq.mid.x;
                    else
                        y[j] = (s & 1) ? dw_eval(wp, tq, std::integral_constant<int, 1>{}) : dw_eval(wp, tq, std::integral_constant<int, 0>{});
                    __builtin_amdgcn_sched_barrier(0);
                    if (feed) product(1, m - 1, j);
                    if ((s & 1) && s + 3 < 32) {  // both steps of the pair are done: its registers take the pair after the next
                        if constexpr (!NO_STENCIL) wts_pair_load((s >> 1) + 2, wp);
                        tap_pair_load((s >> 1) + 2, tq);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (j == 6) {
                        // the products of k-block m-1 are done: its ring slot takes k-block m+1, or k-block 0 of
                        // this wave's next unit / of the next block
                        if (m < 3)
                            load_afrag(w, N, m + 1, lane, pwo.ring[(m + 1) & 1]);
                        else if (t + NW < n_tiles)
                            load_afrag(w, N, 0, lane, pwo.ring[0]);
                        else if (KSL && N == 2 && wv < 4)
                            load_afrag(w, N, wv, lane, pwo.ring[0]);  // this wavefront's quarter of the leftover tile comes next
                        else if (N < 4)
                            load_afrag(w, N < 4 ? N + 1 : N, 0, lane, pwo.ring[0]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (j == 7) {
                        if constexpr (PAIR) {
                            split_pair8_scaled(y, bh, bl);  // (the next product reads bl a stencil evaluation later; the operand
                                                            // scale sits in the depthwise table: store_block_tables)
                        } else {
                            split3(y, bh, bm, bl);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
                for (int q = 0; q < NPROD; ++q) {
                    product(0, 3, q);
                    product(1, 3, q);
                }
            } else {
                auto& wa = pwo.wa;
#pragma unroll
                for (int s = 0; s < 32; s += 2) {  // one pair of k-steps per iteration
                    TapPair& tq = (s & 2) ? tq1 : tq0;
                    DwPair& wp = (s & 2) ? wq1 : wq0;
                    float y0, y1;
                    if constexpr (MODE == 2) {  // timing ablation: matrix core only (results are wrong)
                        y0 = tq.mid.x;
                        y1 = tq.mid.y;
                    } else {
                        y0 = dw_eval(wp, tq, std::integral_constant<int, 0>{});
                        y1 = dw_eval(wp, tq, std::integral_constant<int, 1>{});
                        if (s + 4 < 32) {
                            wts_pair_load((s >> 1) + 2, wp);
                            tap_pair_load((s >> 1) + 2, tq);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (MODE == 3) {  // timing ablation: stencil only (results are wrong)
                        acc0[0] += y0 * wa[0][s];
                        acc1[0] += y1 * wa[1][s + 1];
                    } else {
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[0][s], y0, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[1][s], y0, acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[0][s + 1], y1, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[1][s + 1], y1, acc1, 0, 0, 0);
                    }
                }
            }
            auto epilogue = [&]() {
                // relu() is inline asm: the compiler's hazard recognizer does not see that it reads MFMA results, and the
                // hardware does not interlock a VALU read behind a matrix-core write (XDL write -> VALU read: up to 18 wait
                // states for a 16-pass MFMA).  The wait is spelled out here; the +v ties pin it after the last MFMA.
                // (Round 1's `valid ? relu(x) : 0` happened to put an exec-mask branch in between; a branch-free select
                // read stale accumulators: nondeterministic sums.)
#ifndef KWS_X_NO_MFMA_EPILOGUE_NOP  // (the switch exists for tests/test_isa_hazards.py: without the wait the lint must fail)
                asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc0), "+v"(acc1));
#endif
                if constexpr (N < 4) {
                    if constexpr (PAIR && N <= 2) {
                        // the largest stored value (halo columns hold outputs of real positions too): the scale of the block
                        // after the next is derived from it.  One exec-masked region for the stores, none for the maximum.
                        float o0[16], o1[16];
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            o0[r] = relu(acc0[r]);
                            o1[r] = relu(acc1[r]);
                        }
#pragma unroll
                        for (int r = 0; r < 16; r += 2) stage_max = fmaxf(stage_max, fmaxf(fmaxf(o0[r], o0[r + 1]), fmaxf(o1[r], o1[r + 1])));
                        if (valid) {
#pragma unroll
                            for (int r = 0; r < 16; r += 2) {
                                *reinterpret_cast<float2*>(zout + pidx(row_of(r, half), pos, G::SOUT)) = make_float2(o0[r], o0[r + 1]);
                                *reinterpret_cast<float2*>(zout + pidx(32 + row_of(r, half), pos, G::SOUT)) = make_float2(o1[r], o1[r + 1]);
                            }
                        }
                    } else if (valid) {
#pragma unroll
                        for (int r = 0; r < 16; r += 2) {  // rows r, r+1 are adjacent output channels: one 8-byte store
                            *reinterpret_cast<float2*>(zout + pidx(row_of(r, half), pos, G::SOUT)) =
                                make_float2(relu(acc0[r]), relu(acc0[r + 1]));
                            *reinterpret_cast<float2*>(zout + pidx(32 + row_of(r, half), pos, G::SOUT)) =
                                make_float2(relu(acc1[r]), relu(acc1[r + 1]));
                        }
                    }
                } else {
                    // relu is an asm statement: written as `valid ? relu(x) : 0` every element became its own exec-masked
                    // branch region (16 per unit).  An AND with an all-ones / all-zeros mask selects without a branch -- and,
                    // unlike a 0/1 factor, also if a halo lane ever held a NaN.
                    const uint32_t keep = valid ? 0xffffffffu : 0u;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        psum[0][r] += __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, relu(acc0[r])) & keep);
                        psum[1][r] += __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, relu(acc1[r])) & keep);
                    }
                    if (act4 && valid) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            act4[row_of(r, half) * G::POUT + pos] = relu(acc0[r]) * pc.inv_out;
                            act4[(32 + row_of(r, half)) * G::POUT + pos] = relu(acc1[r]) * pc.inv_out;
                        }
                    }
                }
            };
            // this wave's last unit: the A operands are dead, so the next block's are fetched now and the
            // loads fly under the epilogue, the barrier and the next prologue
            if constexpr (!SPLIT) {
                if (N < 4 && t + NW >= n_tiles) {
                    __builtin_amdgcn_sched_barrier(0);  // not before the last MFMA has read the old operands
                    if constexpr (N < 4) load_pointwise(w, N + 1, lane, pwo);
                    __builtin_amdgcn_sched_barrier(0);
                    epilogue();
                    break;
                }
            }
            epilogue();
        } else {
            // VALU cross-check of the pointwise GEMM: each half sums its 32 input channels, halves are
            // combined with a lane exchange.
            const float* pw_w = w.pw_w + (N - 1) * CH * CH;
            float y[32];
#pragma unroll
            for (int s = 0; s < 32; s += 2) {
                TapPair tq;
                tap_pair_load(s >> 1, tq);
                DwPair wp;
                wts_pair_load(s >> 1, wp);
                y[s] = dw_eval(wp, tq, std::integral_constant<int, 0>{});
                y[s + 1] = dw_eval(wp, tq, std::integral_constant<int, 1>{});
            }
#pragma unroll 1
            for (int co = 0; co < CH; ++co) {
                float part = 0.f;
#pragma unroll
                for (int s = 0; s < 32; ++s) part = fmaf(pw_w[(16 * (s >> 3) + 8 * half + (s & 7)) * CH + co], y[s], part);
                const float tot = relu(part + __shfl_xor(part, 32, 64) + pwb[co]);
                if constexpr (N < 4) {
                    if (valid && half == 0) zout[pidx(co, pos, G::SOUT)] = tot;
                } else {
                    // pool: sum this tile's positions and accumulate into the wave's own scratch row
                    if (act4 && valid && half == 0) act4[co * G::POUT + pos] = tot;
                    float sum = (valid && half == 0) ? tot : 0.f;
#pragma unroll
                    for (int o = 16; o >= 1; o >>= 1) sum += __shfl_xor(sum, o, 64);
                    if (lane == 0) poolbuf[wv * CH + co] += sum;
                }
            }
        }
    }

    if constexpr (KSL) {
        using L = Leftover<N>;
        if (wv >= L::WAVE0 && wv < L::WAVE0 + 4) {
            leftover_partial_unit<N, NP>(lds, lane, wv - L::WAVE0, pwo.ring[0]);
            __builtin_amdgcn_sched_barrier(0);
            load_afrag(w, N + 1, 0, lane, pwo.ring[0]);  // the next block's first operands fly across the barrier
        }
    }
    if constexpr (N < 4) store_block_tables(lds, N + 1, tid, next_tables, pc.s_dwb, pc.s_pwb, pc.s_dww);
    if constexpr (PAIR && N <= 2) publish_wave_max(lds, N == 1 ? 2 : 0, wv, lane, stage_max);
    if constexpr (MFMA) {
        if constexpr (N < 4) {
            if (wv >= n_tiles && !(KSL && N == 1)) load_block_head(w, N + 1, lane, pwo);  // waves without a unit in this block
        } else {
            // reduce the pool partials over the positions held by each half-wave (DPP, no LDS round trips).  Step-major:
            // all 32 sums take a shift step before any takes the next, so a value is read by DPP well after it was written and
            // the VALU -> DPP wait states cost no s_nop (register-major the compiler padded every add: 132 s_nop per clip
            // and wavefront; as builtins it splits every add into v_mov_b32_dpp + v_add_f32).
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                float (&q)[16] = psum[ct];
                // sixteen sums per block, step-major and fused (v_add_f32_dpp reads its own destination shifted): a register is
                // read by DPP sixteen instructions after it was written, so no wait states are owed
                asm volatile("s_nop 4\n\t"  // also covers an EXEC write just before the block (5 wait states before DPP)
                    "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %8, %8, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %9, %9, %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %10, %10, %10 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %11, %11, %11 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %12, %12, %12 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %13, %13, %13 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %14, %14, %14 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %15, %15, %15 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %2, %2, %2 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %3, %3, %3 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %4, %4, %4 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %5, %5, %5 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %6, %6, %6 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %7, %7, %7 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %8, %8, %8 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %9, %9, %9 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %10, %10, %10 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %11, %11, %11 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %12, %12, %12 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %13, %13, %13 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %14, %14, %14 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %15, %15, %15 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %3, %3, %3 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %4, %4, %4 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %5, %5, %5 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %6, %6, %6 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %7, %7, %7 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %8, %8, %8 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %9, %9, %9 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %10, %10, %10 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %11, %11, %11 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %12, %12, %12 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %13, %13, %13 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %14, %14, %14 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %15, %15, %15 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %2, %2, %2 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %3, %3, %3 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %4, %4, %4 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %5, %5, %5 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %6, %6, %6 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %7, %7, %7 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %8, %8, %8 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %9, %9, %9 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %10, %10, %10 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %11, %11, %11 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %12, %12, %12 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %13, %13, %13 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %14, %14, %14 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %15, %15, %15 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                    "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                    "v_add_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                    "v_add_f32_dpp %3, %3, %3 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                    "v_add_f32_dpp %4, %4, %4 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                    "v_add_f32_dpp %5, %5, %5 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                    "v_add_f32_dpp %6, %6, %6 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                    "v_add_f32_dpp %7, %7, %7 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                    "v_add_f32_dpp %8, %8, %8 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                    "v_add_f32_dpp %9, %9, %9 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                    "v_add_f32_dpp %10, %10, %10 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                    "v_add_f32_dpp %11, %11, %11 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                    "v_add_f32_dpp %12, %12, %12 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                    "v_add_f32_dpp %13, %13, %13 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                    "v_add_f32_dpp %14, %14, %14 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                    "v_add_f32_dpp %15, %15, %15 row_bcast:15 row_mask:0xa bank_mask:0xf"
                    : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]), "+v"(q[8]), "+v"(q[9]), "+v"(q[10]), "+v"(q[11]), "+v"(q[12]), "+v"(q[13]), "+v"(q[14]), "+v"(q[15]));
            }
            if (col == 31) {
#pragma unroll
                for (int k = 0; k < 32; ++k) poolbuf[wv * CH + (k >> 4) * 32 + row_of(k & 15, half)] = psum[k >> 4][k & 15] * pc.inv_out;
            }
        }
    }
    (void)bias4;
}


// DIAG = false: the product instantiation -- no activation dump, no stamps (their pointers and loops cost
// registers and 5 KB of code even when unused).
// PRECONV: `feat` is not the MFCC map but conv1's output [B][64][47*3] (ReLU applied), computed by kws_conv1_general_kernel
// for a model with input_channels > 1 (models.py:125,135); the kernel then starts at block 1.
// STREAM: the streaming push in one launch (launch_dscnn_stream).  `feat` is the feature ring (its newest row is written
// here, through a pointer derived from `feat`, and never read), wavefront 0 computes the stream's new frame straight into the LDS feature map while the other
// wavefronts fetch the 98 older rows, and the hop counter sp.hops advances when the last workgroup is done.
// CLUSTER (with STREAM): sp.cluster workgroups share one stream's network by TIME TILES -- at 64 streams one workgroup per
// stream leaves three quarters of the CUs idle and the push latency is one clip's serial path through the kernel.  Workgroup
// (stream, tile) computes the rows of block 4's output that belong to its tile and, of every earlier stage, the rows those
// depend on (one more row per side and stage: ~4 conv1 rows of halo per side, recomputed, no exchange between workgroups);
// everything stays LDS-resident per tile.  Only the LAST tile needs the window's newest row, so only its wavefront 0 runs
// the one-frame front end.  The tiles' pooled partial sums meet in global memory; the workgroup that arrives last (a
// counter per stream) adds them in tile order, adds the ring term and runs fc + argmax.
template <int MODE, bool DIAG = true, bool PRECONV = false, bool STREAM = false, bool CLUSTER = false>
__global__ __launch_bounds__(NT) void kws_dscnn_fwd_kernel(DscnnWeights w, const float* __restrict__ feat, int B,
                                                           float* __restrict__ logits, int32_t* __restrict__ label,
                                                           float* __restrict__ act_arg,
                                                           unsigned long long* __restrict__ stamps_arg,
                                                           const int* __restrict__ ring_hops, StreamPush sp) {
    static_assert(!STREAM || (MODE >= 4 && !DIAG && !PRECONV), "the fused push exists for the product paths only");
    constexpr bool PAIR = MODE == 5;   // f16-pair arithmetic (kws_split_mfma.h): activations live in LDS in per-clip scaled units
    constexpr int NP = PAIR ? 2 : 3;
    BlockTables t1_pre{};              // PRECONV: block 1's tables between their fetch and their (PAIR: scaled) store
    static_assert(!CLUSTER || STREAM, "time-tile clusters exist for the streaming push only");
    float* const act = DIAG ? act_arg : nullptr;
    unsigned long long* const stamps = DIAG ? stamps_arg : nullptr;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr bool MFMA = MODE != 0;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;

    // One clip per workgroup (no persistent loop: hoisting the ~400 weight addresses out of a clip
    // loop costs more registers than the relaunch saves).
    const int cl_n = CLUSTER ? sp.cluster : 1;
    const int clip = CLUSTER ? (int)blockIdx.x / cl_n : (int)blockIdx.x;
    const int cl_tile = CLUSTER ? (int)blockIdx.x - clip * cl_n : 0;
    if (clip >= B) return;
    // Rows of each stage this workgroup computes (CLUSTER; otherwise everything).  Block 4's output rows [j0, j1) need block
    // 3's rows [j0 - 2, j1) (a 3-row stencil, and block 3's stored plane is block 4's input without its ring: row index - 1),
    // those need block 2's [j0 - 4, j1), block 1's [j0 - 6, j1), and block 1 reads conv1's rows [j0 - 7, j1 + 1) (no ring there).
    const int j0 = Blk<4>::H * cl_tile / cl_n, j1 = Blk<4>::H * (cl_tile + 1) / cl_n;
    auto rows = [&](int lo, int hi, int H, int W) { return PosRange{(lo < 0 ? 0 : lo) * W, (hi > H ? H : hi) * W}; };
    const PosRange rg4 = rows(j0, j1, Blk<4>::H, Blk<4>::W), rg3 = rows(j0 - 2, j1, Blk<3>::H, Blk<3>::W),
                   rg2 = rows(j0 - 4, j1, Blk<2>::H, Blk<2>::W), rg1 = rows(j0 - 6, j1, Blk<1>::H, Blk<1>::W),
                   rg0 = rows(j0 - 7, j1 + 1, C1_H, C1_W);
    // diagnostics only (stamps == nullptr in every product call): shader-clock stamps of thread 0 at the
    // phase boundaries, KWS_DSCNN_STAMPS per clip; [14] and [15] carry the 100 MHz real-time counter
    int n_stamp = 0;
    auto stamp = [&]() {
        if (stamps && tid == KWS_X_DSCNN_STAMP_TID) stamps[(size_t)clip * KWS_DSCNN_STAMPS + n_stamp] = __builtin_amdgcn_s_memtime();
        ++n_stamp;
    };
    if (stamps && tid == KWS_X_DSCNN_STAMP_TID) stamps[(size_t)clip * KWS_DSCNN_STAMPS + KWS_DSCNN_STAMPS - 2] = __builtin_amdgcn_s_memrealtime();
    stamp();  // 0: start

    constexpr bool SPLIT = MODE >= 4;
    PwOperands<MODE> wa;            // pointwise operands of the running block
    int hops_before = 0;            // STREAM: pushes before this one
    // PAIR: exponents of the clip's scales.  kx: features; ky[n]: block n's depthwise output (true units) * 2^ky[n] < 2^15;
    // sg[n]: the units stage n's accumulators and stored output are in (sg[0]: conv1), = ky[n] + the layer's weight exponent
    int kx = 0, ky[5] = {0, 0, 0, 0, 0}, sg[5] = {0, 0, 0, 0, 0};
    if constexpr (PRECONV) {
        static_assert(!PRECONV || MODE >= 4, "the pre-convolved entry exists for the product (split) path only");
        const float* z = feat + (size_t)clip * (CH * P0);
        fetch_block_tables(w, 1, tid, t1_pre);
        float zmax = 0.f;
        for (int i = tid; i < CH * P0; i += NT) {
            const float v = z[i];
            lds[OFF_Z0 + pidx(i / P0, i % P0, P0 + 2)] = v;
            zmax = fmaxf(zmax, v);
        }
        if (tid < CH) {
            lds[OFF_Z0 + pidx(tid, P0, P0 + 2)] = 0.f;
            lds[OFF_Z0 + pidx(tid, P0 + 1, P0 + 2)] = 0.f;
        }
        // PAIR: conv1's output arrives in true units (sg[0] = 0) and nothing bounds it in advance: its largest value is measured
        // here, block 1's scale and tables follow behind the barrier below (one more barrier than the single-channel path)
        if constexpr (PAIR)
            publish_wave_max(lds, 1, wv, lane, zmax);
        else
            store_block_tables(lds, 1, tid, t1_pre);
        if constexpr (Leftover<1>::HAS && (MODE == 4 || MODE == 5) && !CLUSTER)
            load_afrag(w, 1, wv >= 4 ? wv - 4 : 0, lane, wa.ring[0]);  // wavefronts 4-7: their k-block of block 1's leftover tile
        else
            load_block_head(w, 1, lane, wa);
        stamp();  // 1
    } else {
    // ---- phase 0: MFCC map -> zero-padded [103][14] in LDS, weight loads in flight --------------------
    // The feature map and block 1's tables are on the critical path of this phase: their loads are issued first
    // (vector memory returns in order), the conv1 operands behind them.
    float* featp = lds + OFF_FEAT;
    const float* f = feat + (size_t)clip * (IN_T * IN_F);
    constexpr int FV = (IN_T * IN_F + NT - 1) / NT;
    float fv[FV];
    // streaming: the feature map is a ring of IN_T frames; after `hops` pushes the newest frame sits in row
    // (hops - K) mod IN_T, K = ceil(frame_len / frame_step) hops per frame, and the window starts one row after it
    int row_new = -1;  // STREAM: window row the new frame takes (-1: none yet)
    int head;
    if constexpr (STREAM) {
        hops_before = *sp.hops;
        // K = hops a frame spans = ceil(frame_len / frame_step) (3 for 400 / 160): after h pushes the newest frame is h - K
        // and the window starts one row after it; here h = hops_before + 1
        const int K = (sp.p.frame_len + sp.p.frame_step - 1) / sp.p.frame_step;
        head = (((hops_before + 2 - K) % IN_T) + IN_T) % IN_T;  // what the two-launch path derives from the advanced counter
        const int step = sp.p.frame_step;
        const long f_start = (long)step * hops_before + step - (long)((sp.p.frame_len + step - 1) / step) * step;
        if (f_start >= 0) row_new = (int)((((f_start / step) - head) % IN_T + IN_T) % IN_T);
    } else {
        head = ring_hops ? (((*ring_hops + 1 - sp.frames_lag) % IN_T) + IN_T) % IN_T : 0;
    }
#pragma unroll
    for (int k = 0; k < FV; ++k) {
        const int i = tid + k * NT;
        fv[k] = (i < IN_T * IN_F && i / IN_F != row_new) ? f[((i / IN_F + head) % IN_T) * IN_F + i % IN_F] : 0.f;
    }
    BlockTables t1;
    fetch_block_tables(w, 1, tid, t1);
    __builtin_amdgcn_sched_barrier(0);
    float a1[SPLIT ? 1 : 50];       // conv1 weights of this wave's output-channel tile (f32 MFMA A operands)
    uintx4 c1f[SPLIT ? 7 : 1][NP];  // the same as bf16 pieces / f16 pairs (split paths)
    if constexpr (SPLIT) {
        const uintx4* src = reinterpret_cast<const uintx4*>(PAIR ? w.c1_pair : w.c1_split) + (size_t)(wv & 1) * (7 * NP * 64) + lane;
#pragma unroll
        for (int kb = 0; kb < 7; ++kb)
#pragma unroll
            for (int p = 0; p < NP; ++p) c1f[kb][p] = src[(kb * NP + p) * 64];
    } else if constexpr (MFMA) {
        const int half = lane >> 5, col = lane & 31, ct = wv & 1;
#pragma unroll
        for (int s = 0; s < 50; ++s) a1[s] = w.c1_w[(2 * s + half) * CH + ct * 32 + col];
        load_pointwise(w, 1, lane, wa);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PAIR) {  // the clip's largest |feature| (the row the streaming front end adds is looked at after the barrier)
        float m = 0.f;
#pragma unroll
        for (int k = 0; k < FV; ++k) m = fmaxf(m, fabsf(fv[k]));
        publish_wave_max(lds, 0, wv, lane, m);
    }
    if constexpr (STREAM) {
        // One barrier: the zero fill touches only the padding, the scatter only the interior, and the new frame's row is
        // written by wavefront 0 alone, straight from the cepstrum registers of its one-frame front end (scratch and tables
        // in the region block 2 will overwrite much later).
#ifdef KWS_X_STREAM_NO_FRAME  // timing experiment (wrong results): the push without the one-frame front end on any tile's path
        if (false)
#else
        if (wv == 0 && cl_tile == cl_n - 1)  // first: its sample and table loads join the feature loads already in flight
#endif
            stream_frame_wave(sp.p, sp.t, sp.hop, clip, clip, false, sp.pcm_ring, sp.ring_len, const_cast<float*>(feat), hops_before,
                              reinterpret_cast<unsigned char*>(lds + OFF_Z2), lane,
                              row_new >= 0 ? featp + (row_new + 2) * FEAT_W + 2 : nullptr, sp.refine_ctr);
        for (int i = tid; i < FEAT_H * FEAT_W; i += NT) {
            const int r = i / FEAT_W - 2, c = i % FEAT_W - 2;
            if (!((unsigned)r < (unsigned)IN_T && (unsigned)c < (unsigned)IN_F)) featp[i] = 0.f;
        }
        if constexpr (!PAIR) store_block_tables(lds, 1, tid, t1);
#pragma unroll
        for (int k = 0; k < FV; ++k) {
            const int i = tid + k * NT;
            if (i < IN_T * IN_F && i / IN_F != row_new) featp[(i / IN_F + 2) * FEAT_W + (i % IN_F) + 2] = fv[k];
        }
        __syncthreads();
    } else {
    // one barrier: the zero fill touches only the padding cells, the scatter only the interior
    for (int i = tid; i < FEAT_H * FEAT_W; i += NT) {
        const int r = i / FEAT_W - 2, c = i % FEAT_W - 2;
        if (!((unsigned)r < (unsigned)IN_T && (unsigned)c < (unsigned)IN_F)) featp[i] = 0.f;
    }
    if constexpr (!PAIR) store_block_tables(lds, 1, tid, t1);
#pragma unroll
    for (int k = 0; k < FV; ++k) {
        const int i = tid + k * NT;
        if (i < IN_T * IN_F) featp[(i / IN_F + 2) * FEAT_W + (i % IN_F) + 2] = fv[k];
    }
    __syncthreads();
    }
    if constexpr (PAIR) {
        // Scales of conv1 and block 1, from the clip's largest |feature| Mx and bounds that hold for any input (kws_internal.h):
        // features * 2^kx < 2^15; conv1's accumulators and stored output are in units 2^sg0; block 1's depthwise output
        // (true units) is below dw_abs (c1_abs Mx + c1_bmax) + dw_bmax, which fixes its operand scale 2^ky1 and its units.
        float mxf = read_stage_max(lds, 0, 1);
        if constexpr (STREAM) {
            // the window's newest row exists only in the workgroup whose wavefront 0 computed it (the last time tile); in the
            // others its cells are never written (their ranges stop short of it)
            if (row_new >= 0 && cl_tile == cl_n - 1)
                for (int c = 0; c < IN_F; ++c) mxf = fmaxf(mxf, fabsf(featp[(row_new + 2) * FEAT_W + 2 + c]));
        }
        kx = pow2_exp_for(mxf);
        const float bz0 = (w.c1_abs * mxf + w.c1_bmax) * 1.001f;
        cap_units(kx, sg[0], w.k_c1, bz0);
        const float by1 = (w.dw_abs[0] * bz0 + w.dw_bmax[0]) * 1.001f;
        ky[1] = pow2_exp_for(by1);
        cap_units(ky[1], sg[1], w.k_pw[0], (w.pw_abs[0] * by1 + w.pw_bmax[0]) * 1.001f);
        store_block_tables(lds, 1, tid, t1, pow2f(ky[1]), pow2f(sg[1]), pow2f(ky[1] - sg[0]));  // (read in block 1, behind conv1's barrier)
        conv1_build_windows(lds, tid, pow2f(kx));                          // conv1's operands, split once
        __syncthreads();
    }
    stamp();  // 1: features staged

    if constexpr (SPLIT) {
        conv1_phase_split<CLUSTER, NP>(w, lds, tid, c1f, rg0, pow2f(kx), pow2f(sg[0]));
        // the conv1 operands are dead: block 1's first fly across the barrier (wavefronts 4-7: their k-block of the leftover tile)
        if constexpr (Leftover<1>::HAS && (MODE == 4 || MODE == 5) && !CLUSTER)
            load_afrag(w, 1, wv >= 4 ? wv - 4 : 0, lane, wa.ring[0]);
        else
            load_block_head(w, 1, lane, wa);
    } else {
        conv1_phase<MFMA>(w, lds, tid, a1);
    }
    }
    stamp();  // 2: conv1 units of wave 0 done
    __syncthreads();
    stamp();  // 3: conv1 barrier
    float* a = act ? act + (size_t)clip * KWS_ACT_FLOATS_PER_CLIP : nullptr;
    constexpr bool KSL_ON = (MODE == 4 || MODE == 5) && !CLUSTER;  // the leftover tiles of blocks 1 / 2 are K-split
    // PAIR: the planes hold activations * 2^sg[n]; the dumps (diagnostics) go out in true units
    if (a) {
        const float u = PAIR ? pow2f(-sg[0]) : 1.f;
        for (int i = tid; i < CH * P0; i += NT) a[i] = lds[OFF_Z0 + pidx(i / P0, i % P0, P0 + 2)] * u;
        a += CH * P0;
    }
    PairCtx pc;
    if constexpr (PAIR && PRECONV) {
        const float mz0 = read_stage_max(lds, 1, 1);  // true units
        const float by1 = (w.dw_abs[0] * mz0 + w.dw_bmax[0]) * 1.001f;
        ky[1] = pow2_exp_for(by1);
        cap_units(ky[1], sg[1], w.k_pw[0], (w.pw_abs[0] * by1 + w.pw_bmax[0]) * 1.001f);
        store_block_tables(lds, 1, tid, t1_pre, pow2f(ky[1]), pow2f(sg[1]), pow2f(ky[1]));
        __syncthreads();
    }
    if constexpr (PAIR) {
        // conv1's largest output is known now: it bounds block 1's output, which fixes block 2's operand scale and units --
        // two layers ahead, so that block 2's tables can be stored (scaled) while block 1 runs
        const float mz = read_stage_max(lds, 1, 1) * pow2f(-sg[0]);
        const float bz = (w.pw_abs[0] * ((w.dw_abs[0] * mz + w.dw_bmax[0]) * 1.001f) + w.pw_bmax[0]) * 1.001f;
        const float by = (w.dw_abs[1] * bz + w.dw_bmax[1]) * 1.001f;
        ky[2] = pow2_exp_for(by);
        cap_units(ky[2], sg[2], w.k_pw[1], (w.pw_abs[1] * by + w.pw_bmax[1]) * 1.001f);
        pc.s_dww = pow2f(ky[2] - sg[1]);
        pc.s_dwb = pow2f(ky[2]);
        pc.s_pwb = pow2f(sg[2]);
    }
    block_phase<1, MODE, CLUSTER>(w, lds, tid, wa, nullptr, rg1, pc);
    stamp();  // 4: block 1 units of wave 0 done
    __syncthreads();
    if constexpr (Leftover<1>::HAS && KSL_ON) {
        leftover_combine<1, PAIR>(lds, tid);
        __syncthreads();
    }
    stamp();  // 5: block 1 barrier
    if (a) {
        const float u = PAIR ? pow2f(-sg[1]) : 1.f;
        for (int i = tid; i < CH * Blk<1>::POUT; i += NT)
            a[i] = lds[OFF_Z1 + pidx(i / Blk<1>::POUT, i % Blk<1>::POUT, Blk<1>::SOUT)] * u;
        a += CH * Blk<1>::POUT;
    }
    if constexpr (PAIR) {
        // block 2 reads block 1's interior and its ring (relu(bias) <= pw_bmax)
        const float mz = fmaxf(read_stage_max(lds, 2, (Leftover<1>::HAS && KSL_ON) ? 2 : 1) * pow2f(-sg[1]), w.pw_bmax[0]);
        const float bz = (w.pw_abs[1] * ((w.dw_abs[1] * mz + w.dw_bmax[1]) * 1.001f) + w.pw_bmax[1]) * 1.001f;
        const float by = (w.dw_abs[2] * bz + w.dw_bmax[2]) * 1.001f;
        ky[3] = pow2_exp_for(by);
        cap_units(ky[3], sg[3], w.k_pw[2], (w.pw_abs[2] * by + w.pw_bmax[2]) * 1.001f);
        pc.s_dww = pow2f(ky[3] - sg[2]);
        pc.s_dwb = pow2f(ky[3]);
        pc.s_pwb = pow2f(sg[3]);
    }
    block_phase<2, MODE, CLUSTER>(w, lds, tid, wa, nullptr, rg2, pc);
    stamp();  // 6
    __syncthreads();
    if constexpr (Leftover<2>::HAS && KSL_ON) {
        leftover_combine<2, PAIR>(lds, tid);
        __syncthreads();
    }
    stamp();  // 7
    if (a) {
        const float u = PAIR ? pow2f(-sg[2]) : 1.f;
        for (int i = tid; i < CH * Blk<2>::POUT; i += NT)
            a[i] = lds[OFF_Z2 + pidx(i / Blk<2>::POUT, i % Blk<2>::POUT, Blk<2>::SOUT)] * u;
        a += CH * Blk<2>::POUT;
    }
    if constexpr (PAIR) {
        const float mz = fmaxf(read_stage_max(lds, 0, (Leftover<2>::HAS && KSL_ON) ? 2 : 1) * pow2f(-sg[2]), w.pw_bmax[1]);
        const float bz = (w.pw_abs[2] * ((w.dw_abs[2] * mz + w.dw_bmax[2]) * 1.001f) + w.pw_bmax[2]) * 1.001f;
        const float by = (w.dw_abs[3] * bz + w.dw_bmax[3]) * 1.001f;
        ky[4] = pow2_exp_for(by);
        cap_units(ky[4], sg[4], w.k_pw[3], (w.pw_abs[3] * by + w.pw_bmax[3]) * 1.001f);
        pc.s_dww = pow2f(ky[4] - sg[3]);
        pc.s_dwb = pow2f(ky[4]);
        pc.s_pwb = pow2f(sg[4]);
    }
    block_phase<3, MODE, CLUSTER>(w, lds, tid, wa, nullptr, rg3, pc);
    stamp();  // 8
    __syncthreads();
    stamp();  // 9
    if (a) {
        const float u = PAIR ? pow2f(-sg[3]) : 1.f;
        for (int i = tid; i < CH * Blk<3>::POUT; i += NT)
            a[i] = lds[OFF_Z3 + pidx(i / Blk<3>::POUT, i % Blk<3>::POUT, Blk<3>::SOUT)] * u;
        a += CH * Blk<3>::POUT;
    }
    if constexpr (PAIR) {
        pc.inv_out = pow2f(-sg[4]);
    }
    block_phase<4, MODE, CLUSTER>(w, lds, tid, wa, a ? a + CH : nullptr, rg4, pc);  // block 4's output follows the pooled means
    stamp();  // 10
    // The classifier row of lane c (wavefront 0) and the ring bias of channel tid are requested BEFORE the barrier:
    // their L2 round trips pass while the workgroup waits for its slowest wavefront, instead of sitting exposed at
    // the very end of the clip (the CU cannot take its next clip before this wavefront is done).
    const int C = w.num_classes;
    float4 fcw[CH / 4];
    float fcb = 0.f, ring_b = 0.f;
    if (wv == 0 && lane < C) {
        const float4* wr = reinterpret_cast<const float4*>(w.fc_w + lane * CH);
#pragma unroll
        for (int c = 0; c < CH / 4; ++c) fcw[c] = wr[c];
        fcb = w.fc_b[lane];
    }
    if (tid < CH) ring_b = w.pw_b[3 * CH + tid];
    __syncthreads();
    stamp();  // 11

    // ---- global average pool over 55 x 11 = 477 interior + 128 ring positions ---------------------
    float* pooled = lds + OFF_POOLED;
    constexpr float RING_N = 55.f * 11.f - 53.f * 9.f;  // 128
    bool run_fc = true;  // workgroup-uniform
    if constexpr (CLUSTER) {
        // this tile's partial sums go to global memory (agent-scope stores: the other tiles' workgroups sit on other XCDs, whose
        // L2s are not coherent with this one); the workgroup of the stream that arrives last combines them in tile order
        float* part = sp.cl_part + ((size_t)clip * cl_n + cl_tile) * CH;
        if (tid < CH) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < NW; ++k) s += lds[OFF_POOLBUF + k * CH + tid];
            __hip_atomic_store(part + tid, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // No fence: an agent-scope release writes the XCD's whole L2 back -- measured 0.33 us per workgroup, serialised
        // across the chip (256 workgroups: 88 us).  The partial sums are agent-scope atomic stores (write-through to the
        // coherence point); once they are acknowledged (vmcnt 0) any XCD's agent-scope load sees them, and the counter is
        // bumped only after the whole workgroup has passed that wait.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* last_flag = reinterpret_cast<int*>(lds + OFF_POOLED + CH);
        if (tid == 0) {
            const int arrived = atomicAdd(&sp.cl_count[clip], 1);
            *last_flag = arrived == cl_n - 1;
            if (arrived == cl_n - 1) sp.cl_count[clip] = 0;  // for the next push (published by the kernel boundary)
        }
        __syncthreads();
        run_fc = *last_flag != 0;
        if (run_fc) {
            if (tid < CH) {  // agent-scope loads: never served from this XCD's (non-coherent) L2
                float s = 0.f;
                for (int t = 0; t < cl_n; ++t)
                    s += __hip_atomic_load(sp.cl_part + ((size_t)clip * cl_n + t) * CH + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pooled[tid] = fmaf(RING_N, relu(ring_b), s) * (1.0f / (55.f * 11.f));
            }
        }
        __syncthreads();
    } else {
    if (tid < CH) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NW; ++k) s += lds[OFF_POOLBUF + k * CH + tid];
        s = fmaf(RING_N, relu(ring_b), s) * (1.0f / (55.f * 11.f));
        pooled[tid] = s;
        if (a) a[tid] = s;
    }
    __syncthreads();
    }

    // ---- Linear(64 -> C) + argmax (first maximum wins) on wavefront 0 ------------------------------
    if (run_fc && wv == 0) {
        float v = -INFINITY;
        if (lane < C) {
            float acc = fcb;
            const float4* pl = reinterpret_cast<const float4*>(pooled);
#pragma unroll
            for (int c = 0; c < CH / 4; ++c) {
                const float4 a4 = fcw[c], p4 = pl[c];
                acc = fmaf(a4.x, p4.x, acc);
                acc = fmaf(a4.y, p4.y, acc);
                acc = fmaf(a4.z, p4.z, acc);
                acc = fmaf(a4.w, p4.w, acc);
            }
            logits[(size_t)clip * C + lane] = acc;
            if constexpr (STREAM) {  // zero-copy delivery: the host's pinned copy, written through (system scope), see below
                if (sp.h_logits) __hip_atomic_store(sp.h_logits + (size_t)clip * C + lane, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            v = acc;
        }
        // argmax, first maximum wins: wave maximum by DPP (no LDS round trips; six dependent __shfl_xor rounds through
        // ds_bpermute were 1.4 k of the 2.6 k cycles this tail took), then the lowest lane that holds it
        float m = v;
        m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, m), __builtin_bit_cast(int, m), 0x111, 0xf, 0xf, false)));
        m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, m), __builtin_bit_cast(int, m), 0x112, 0xf, 0xf, false)));
        m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, m), __builtin_bit_cast(int, m), 0x114, 0xf, 0xf, false)));
        m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, m), __builtin_bit_cast(int, m), 0x118, 0xf, 0xf, false)));
        m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, m), __builtin_bit_cast(int, m), 0x142, 0xa, 0xf, false)));
        m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, m), __builtin_bit_cast(int, m), 0x143, 0xc, 0xf, false)));
        const float vmax = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m), 63));
        const unsigned long long holders = __ballot(lane < C && v == vmax);
        const int idx = holders ? __ffsll(holders) - 1 : 0;  // all-NaN logits: label 0
        if (label && lane == 0) label[clip] = idx;
        if constexpr (STREAM) {
            if (sp.h_label && lane == 0) __hip_atomic_store(sp.h_label + clip, idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if constexpr (STREAM) {
        // hop counter: every workgroup read sp.hops[0] in its prologue; the one that finishes last advances it
        // (sp.hops[1] counts finished workgroups).  The kernel boundary publishes it to the next push.
        // Zero-copy result delivery (kws_stream_host_results): logits and labels also went to pinned host memory as
        // system-scope stores; a workgroup bumps the counter only after its own have been acknowledged (vmcnt 0: they are
        // on their way over PCIe, in order), and the last one raises the host's flag behind them -- posted writes of one
        // device keep their order, so the host that sees the flag sees every stream's results.  No fence (they write whole
        // L2s back on this part).
        // Only the workgroup that ran a stream's fc takes part (with time tiles it is the last of the stream's workgroups to
        // arrive, so every tile of the stream is past its reads of the counter): one add per STREAM to this one address, not
        // one per workgroup -- adds from eight XCDs to one address queue up at the coherence point.
        int finishers = (int)gridDim.x;
        if constexpr (CLUSTER) finishers /= cl_n;
        if (sp.h_flag) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (run_fc && tid == 0 && atomicAdd(&sp.hops[1], 1) == finishers - 1) {
            sp.hops[1] = 0;
            sp.hops[0] = hops_before + 1;
            if (sp.h_flag) __hip_atomic_store(sp.h_flag, hops_before + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    stamp();  // 12: pool + fc + argmax done
    if (stamps && tid == KWS_X_DSCNN_STAMP_TID) stamps[(size_t)clip * KWS_DSCNN_STAMPS + KWS_DSCNN_STAMPS - 1] = __builtin_amdgcn_s_memrealtime();
}

}  // namespace

// The kernel needs the CU's whole 160 KiB of LDS as dynamic shared memory: opt in once per device.
hipError_t dscnn_init_device() {
    const int lds = LDS_FLOATS * (int)sizeof(float);
    const void* kernels[] = {reinterpret_cast<const void*>(kws_dscnn_fwd_kernel<0>), reinterpret_cast<const void*>(kws_dscnn_fwd_kernel<1>),
                             reinterpret_cast<const void*>(kws_dscnn_fwd_kernel<2>), reinterpret_cast<const void*>(kws_dscnn_fwd_kernel<3>),
                             reinterpret_cast<const void*>(kws_dscnn_fwd_kernel<4>), reinterpret_cast<const void*>(kws_dscnn_fwd_kernel<6>),
                             reinterpret_cast<const void*>(kws_dscnn_fwd_kernel<4, false>),
                             reinterpret_cast<const void*>(kws_dscnn_fwd_kernel<4, false, true>),
                             reinterpret_cast<const void*>(kws_dscnn_fwd_kernel<4, false, false, true>),
                             reinterpret_cast<const void*>(kws_dscnn_fwd_kernel<4, false, false, true, true>),
                             reinterpret_cast<const void*>(kws_dscnn_fwd_kernel<5>), reinterpret_cast<const void*>(kws_dscnn_fwd_kernel<5, false>),
                             reinterpret_cast<const void*>(kws_dscnn_fwd_kernel<5, false, true>),
                             reinterpret_cast<const void*>(kws_dscnn_fwd_kernel<5, false, false, true>),
                             reinterpret_cast<const void*>(kws_dscnn_fwd_kernel<5, false, false, true, true>)};
    for (const void* k : kernels) {
        hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// conv1 for input_channels > 1 (reference kws/libs/models.py:125,135: Conv2d(input_channels, 64, 10, stride 2, padding 2)):
// one 512-thread workgroup per clip; thread (co = tid & 63, group g = tid >> 6) owns output channel co at positions
// g, g + 8, ...; per input channel the zero-padded plane goes through LDS (every lane of a wavefront reads the same
// address: a broadcast) and the 100 taps come from a [ci][tap][co] weight image (coalesced).  ReLU(bias + sum) -> [64][141].
namespace {
__global__ __launch_bounds__(NT) void kws_conv1_general_kernel(const float* __restrict__ x, int C_in, const float* __restrict__ wt,
                                                               const float* __restrict__ bias, float* __restrict__ out) {
    __shared__ float plane[FEAT_H * FEAT_W];
    const int tid = threadIdx.x, co = tid & 63, g = tid >> 6;
    constexpr int PER = (P0 + NW - 1) / NW;  // positions per thread
    float acc[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) acc[k] = 0.f;
    const float* xc = x + (size_t)blockIdx.x * C_in * (IN_T * IN_F);
    for (int ci = 0; ci < C_in; ++ci) {
        __syncthreads();
        for (int i = tid; i < FEAT_H * FEAT_W; i += NT) {
            const int r = i / FEAT_W - 2, cidx = i % FEAT_W - 2;
            plane[i] = ((unsigned)r < (unsigned)IN_T && (unsigned)cidx < (unsigned)IN_F) ? xc[(size_t)ci * (IN_T * IN_F) + r * IN_F + cidx] : 0.f;
        }
        __syncthreads();
        const float* wc = wt + (size_t)ci * (C1_K * C1_K) * CH + co;
        for (int tap = 0; tap < C1_K * C1_K; ++tap) {
            const float wv_ = wc[(size_t)tap * CH];
            const int kh = tap / C1_K, kw = tap % C1_K;
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int pos = g + NW * k;
                if (pos < P0) acc[k] = fmaf(wv_, plane[(2 * (pos / C1_W) + kh) * FEAT_W + 2 * (pos % C1_W) + kw], acc[k]);
            }
        }
    }
    float* o = out + (size_t)blockIdx.x * (CH * P0) + (size_t)co * P0;
    const float bv = bias[co];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int pos = g + NW * k;
        if (pos < P0) o[pos] = relu(acc[k] + bv);
    }
}
}  // namespace

hipError_t launch_conv1_general(hipStream_t s, const float* d_x, int B, int C_in, const float* d_wt, const float* d_bias, float* d_out) {
    hipLaunchKernelGGL(kws_conv1_general_kernel, dim3(B), dim3(NT), 0, s, d_x, C_in, d_wt, d_bias, d_out);
    return hipGetLastError();
}

hipError_t launch_dscnn_stream(hipStream_t s, const DscnnWeights& w, const StreamPush& sp, float* d_feat_ring, int n_streams,
                               float* d_logits, int32_t* d_label, bool pair) {
    static_assert(sizeof(float) * (size_t)(LDS_FLOATS - OFF_Z2) >= 16 * 1024 + STREAM_F64_BYTES, "room for the one-frame front end's tables and scratch");
    const size_t lds = LDS_FLOATS * sizeof(float);
    if (sp.cluster > 1) {
        if (pair)
            hipLaunchKernelGGL((kws_dscnn_fwd_kernel<5, false, false, true, true>), dim3(n_streams * sp.cluster), dim3(NT), lds, s, w, d_feat_ring,
                               n_streams, d_logits, d_label, nullptr, nullptr, nullptr, sp);
        else
            hipLaunchKernelGGL((kws_dscnn_fwd_kernel<4, false, false, true, true>), dim3(n_streams * sp.cluster), dim3(NT), lds, s, w, d_feat_ring,
                               n_streams, d_logits, d_label, nullptr, nullptr, nullptr, sp);
    } else {
        if (pair)
            hipLaunchKernelGGL((kws_dscnn_fwd_kernel<5, false, false, true>), dim3(n_streams), dim3(NT), lds, s, w, d_feat_ring, n_streams, d_logits,
                               d_label, nullptr, nullptr, nullptr, sp);
        else
            hipLaunchKernelGGL((kws_dscnn_fwd_kernel<4, false, false, true>), dim3(n_streams), dim3(NT), lds, s, w, d_feat_ring, n_streams, d_logits,
                               d_label, nullptr, nullptr, nullptr, sp);
    }
    return hipGetLastError();
}

hipError_t launch_dscnn(hipStream_t s, const DscnnWeights& w, const float* d_feat, int B, float* d_logits,
                        int32_t* d_label, float* d_act, int mode, unsigned long long* d_stamps, const int* d_ring_hops,
                        bool preconv, int frames_lag) {
    const size_t lds = LDS_FLOATS * sizeof(float);
    StreamPush lag{};  // the two-launch streaming route: only the hops-per-frame count travels (the window's first row)
    lag.frames_lag = frames_lag;
    const int grid = B;  // one clip per workgroup; one workgroup per CU (160 KiB LDS)
    if (preconv) {  // d_feat = conv1 output of a multi-channel model (kws_conv1_general_kernel): product paths only
        if (mode == 5)
            hipLaunchKernelGGL((kws_dscnn_fwd_kernel<5, false, true>), dim3(grid), dim3(NT), lds, s, w, d_feat, B, d_logits, d_label, nullptr, nullptr, nullptr, StreamPush{});
        else
            hipLaunchKernelGGL((kws_dscnn_fwd_kernel<4, false, true>), dim3(grid), dim3(NT), lds, s, w, d_feat, B, d_logits, d_label, nullptr, nullptr, nullptr, StreamPush{});
        return hipGetLastError();
    }
    // mode: 0 = VALU cross-check of the GEMMs, 1 = product path, 2 / 3 = timing ablations (matrix core only /
    // stencil only; wrong results by construction, reachable only through the diagnostics entry point)
    switch (mode) {
        case 0: hipLaunchKernelGGL(kws_dscnn_fwd_kernel<0>, dim3(grid), dim3(NT), lds, s, w, d_feat, B, d_logits, d_label, d_act, d_stamps, d_ring_hops, lag); break;
        case 2: hipLaunchKernelGGL(kws_dscnn_fwd_kernel<2>, dim3(grid), dim3(NT), lds, s, w, d_feat, B, d_logits, d_label, d_act, d_stamps, d_ring_hops, lag); break;
        case 3: hipLaunchKernelGGL(kws_dscnn_fwd_kernel<3>, dim3(grid), dim3(NT), lds, s, w, d_feat, B, d_logits, d_label, d_act, d_stamps, d_ring_hops, lag); break;
        case 4:
            if (d_act || d_stamps)
                hipLaunchKernelGGL((kws_dscnn_fwd_kernel<4, true>), dim3(grid), dim3(NT), lds, s, w, d_feat, B, d_logits, d_label, d_act, d_stamps, d_ring_hops, lag);
            else
                hipLaunchKernelGGL((kws_dscnn_fwd_kernel<4, false>), dim3(grid), dim3(NT), lds, s, w, d_feat, B, d_logits, d_label, d_act, d_stamps, d_ring_hops, lag);
            break;
        case 5:
            if (d_act || d_stamps)
                hipLaunchKernelGGL((kws_dscnn_fwd_kernel<5, true>), dim3(grid), dim3(NT), lds, s, w, d_feat, B, d_logits, d_label, d_act, d_stamps, d_ring_hops, lag);
            else
                hipLaunchKernelGGL((kws_dscnn_fwd_kernel<5, false>), dim3(grid), dim3(NT), lds, s, w, d_feat, B, d_logits, d_label, d_act, d_stamps, d_ring_hops, lag);
            break;
        case 6: hipLaunchKernelGGL(kws_dscnn_fwd_kernel<6>, dim3(grid), dim3(NT), lds, s, w, d_feat, B, d_logits, d_label, d_act, d_stamps, d_ring_hops, lag); break;
        default: hipLaunchKernelGGL(kws_dscnn_fwd_kernel<1>, dim3(grid), dim3(NT), lds, s, w, d_feat, B, d_logits, d_label, d_act, d_stamps, d_ring_hops, lag); break;
    }
    return hipGetLastError();
}


// ------------------------------------------------------------------------------------------------
// Posteriors (nothing in the reference: its scripts take argmax of the logits).
namespace {

// one thread per clip / stream: C <= 64 values, the work is launch latency, not arithmetic
__device__ __forceinline__ void softmax_row(const float* __restrict__ z, int C, float* __restrict__ p) {
    float m = z[0];
    for (int i = 1; i < C; ++i) m = fmaxf(m, z[i]);
    float sum = 0.f;
    for (int i = 0; i < C; ++i) {
        const float e = expf(z[i] - m);
        p[i] = e;
        sum += e;
    }
    const float inv = 1.0f / sum;
    for (int i = 0; i < C; ++i) p[i] *= inv;
}

__global__ void kws_softmax_f32_kernel(const float* __restrict__ logits, int B, int C, float* __restrict__ prob) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) softmax_row(logits + (size_t)b * C, C, prob + (size_t)b * C);
}

// Moving average of the last `window` posterior vectors per stream (ring [S][window][C], running sum [S][C]),
// then argmax of the smoothed vector (first maximum wins).  count = hops smoothed so far, before this one.
// The running sum is updated incrementally (sum += p - oldest) and REBUILT from the ring every `window` hops (when the
// write slot wraps to 0), so its float32 rounding error is bounded by one window's worth of updates instead of growing
// over the life of a stream (10 ms hops = 8.6 M updates a day).
__global__ void kws_smooth_posteriors_kernel(const float* __restrict__ logits, int S, int C, int window,
                                             float* __restrict__ ring, float* __restrict__ sum, int* __restrict__ count_ptr,
                                             float* __restrict__ smoothed, int32_t* __restrict__ label) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    const int count = count_ptr[0];
    // every workgroup has read the hop count; the one that finishes last advances it (count_ptr[1] = done counter)
    auto finish = [&]() {
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            if (atomicAdd(&count_ptr[1], 1) == (int)gridDim.x - 1) {
                count_ptr[1] = 0;
                count_ptr[0] = count + 1;
            }
        }
    };
    if (s >= S) {
        finish();
        return;
    }
    const int slot = count % window;
    float p[MAX_CLASSES];
    softmax_row(logits + (size_t)s * C, C, p);
    float* r = ring + ((size_t)s * window + slot) * C;
    float* acc = sum + (size_t)s * C;
    const float inv = 1.0f / (float)((count + 1 < window) ? count + 1 : window);
    float best = -1.f;
    int arg = 0;
    const bool rebuild = count >= window && slot == 0;
    for (int i = 0; i < C; ++i) {
        const float old = count >= window ? r[i] : 0.f;
        r[i] = p[i];
        float a;
        if (rebuild) {
            a = 0.f;
            for (int k = 0; k < window; ++k) a += ring[((size_t)s * window + k) * C + i];
        } else {
            a = acc[i] + (p[i] - old);
        }
        acc[i] = a;
        const float v = a * inv;
        smoothed[(size_t)s * C + i] = v;
        if (v > best) {
            best = v;
            arg = i;
        }
    }
    if (label) label[s] = arg;
    finish();
}

}  // namespace

hipError_t launch_softmax(hipStream_t s, const float* d_logits, int B, int C, float* d_prob) {
    hipLaunchKernelGGL(kws_softmax_f32_kernel, dim3((B + 255) / 256), dim3(256), 0, s, d_logits, B, C, d_prob);
    return hipGetLastError();
}

// Energy endpointer per stream (the gate that replaces webrtcvad in the reference's live loop,
// kws/inference/inference_local.py:131-166 -- same hysteresis, at hop granularity).  The newest frame's log energy
// (cepstrum 0 with appendEnergy) above the threshold marks the hop voiced; an utterance OPENS when more than 80 % of the
// last `on_window` hops are voiced (:151) and CLOSES when more than 90 % of the last `off_window` hops are unvoiced
// (:161); hops before the stream began count as unvoiced (the reference's rings start as zeros).  One thread per stream.
// state[s] = triggered | event << 1, event 1 = opened at this hop, 2 = closed at this hop.
__global__ void kws_stream_vad_kernel(const float* __restrict__ feat_ring, const int* __restrict__ hops_ptr, int n_streams,
                                      int num_frames, int numcep, int frames_lag, float threshold, int on_window, int off_window,
                                      unsigned char* __restrict__ flags, int* __restrict__ cursor_trig, int32_t* __restrict__ state) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_streams) return;
    const int hops = *hops_ptr;  // already advanced by the push this call follows: the newest frame is hops - K (K hops per frame)
    if (hops < frames_lag) {     // no complete frame yet
        state[s] = 0;
        return;
    }
    const float c0 = feat_ring[((size_t)s * num_frames + (hops - frames_lag) % num_frames) * numcep];
    unsigned char* fl = flags + (size_t)s * off_window;
    const int cur = cursor_trig[2 * s];
    int trig = cursor_trig[2 * s + 1];
    fl[cur % off_window] = c0 > threshold ? 1 : 0;
    int n_on = 0, n_all = 0;
    for (int k = 0; k < off_window; ++k) {  // k hops back from the newest
        const int v = k <= cur ? fl[(cur - k) % off_window] : 0;
        n_all += v;
        if (k < on_window) n_on += v;
    }
    int event = 0;
    if (!trig) {
        if (10 * n_on > 8 * on_window) trig = 1, event = 1;
    } else if (10 * (off_window - n_all) > 9 * off_window) {
        trig = 0, event = 2;
    }
    cursor_trig[2 * s] = cur + 1;
    cursor_trig[2 * s + 1] = trig;
    state[s] = trig | (event << 1);
}

hipError_t launch_stream_vad(hipStream_t s, const float* d_feat_ring, const int* d_hops, int n_streams, int num_frames, int numcep,
                             int frames_lag, float threshold, int on_window, int off_window, unsigned char* d_flags, int* d_cursor_trig,
                             int32_t* d_state) {
    hipLaunchKernelGGL(kws_stream_vad_kernel, dim3((n_streams + 63) / 64), dim3(64), 0, s, d_feat_ring, d_hops, n_streams,
                       num_frames, numcep, frames_lag, threshold, on_window, off_window, d_flags, d_cursor_trig, d_state);
    return hipGetLastError();
}

hipError_t launch_smooth_posteriors(hipStream_t s, const float* d_logits, int S, int C, int window, float* d_ring,
                                    float* d_sum, int* d_count, float* d_smoothed, int32_t* d_label) {
    hipLaunchKernelGGL(kws_smooth_posteriors_kernel, dim3((S + 63) / 64), dim3(64), 0, s, d_logits, S, C, window, d_ring, d_sum,
                       d_count, d_smoothed, d_label);
    return hipGetLastError();
}

}  // namespace kws
