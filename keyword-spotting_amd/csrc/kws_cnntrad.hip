// cnn-trad-fpool3 forward for gfx950 (MI355X) -- a build-defined member of the model zoo (SURVEY.md section 8 f-4;
// the reference only names it, test.py:80).  Sainath & Parada's "cnn-trad-fpool3" on the reference's [1,99,10] MFCC
// map with SAME padding:
//   conv1  1->64, 20 (time) x 8 (freq), pad 9/10 x 3/4, ReLU          -> 64 x 99 x 10
//   max-pool 1 x 3 over frequency, stride 3 (floor)                    -> 64 x 99 x 3
//   conv2  64->64, 10 x 4, pad 4/5 x 1/2, ReLU                          -> 64 x 99 x 3
//   flatten (channel-major, 19 008) -> Linear 32 -> Linear 128 + ReLU -> Linear C ; argmax (first maximum wins)
// 59.4 M multiply-adds per clip, 82 % of them in conv2.
//
// Kernel A (one 512-thread workgroup per clip): both convolutions as implicit GEMMs on the bf16 matrix pipe with
// the exact three-way bf16 split of kws_split_mfma.h (f32-grade results).  conv1 gathers its f32 im2col operand
// from the zero-padded map in LDS and splits it on the fly; its ReLU'd, frequency-pooled output is written to LDS
// ALREADY split, as three bf16 planes [position][input channel], so that conv2's B operand -- eight consecutive
// input channels of one input position -- is a single aligned ds_read_b128 per piece with no VALU work in the
// loop.  Taps that fall into the padding read a zero region.  conv2's pre-split weights (983 KB) stream from L2.
// Kernel B: the dense tail, 16 clips per workgroup; its 19008 -> 32 layer is a GEMM on the same matrix-pipe path.
#include <type_traits>

#include "kws_internal.h"
#include "kws_split_mfma.h"

namespace kws {
namespace {

constexpr int CT_T = 99, CT_F = 10, CT_FP = 3, CT_P2 = CT_T * CT_FP;  // 297 pooled positions
constexpr int CT_K1H = 20, CT_K1W = 8, CT_K2H = 10, CT_K2W = 4;
constexpr int XP_H = CT_T + CT_K1H - 1, XP_W = CT_F + CT_K1W - 1;      // 118 x 17 zero-padded conv1 input
constexpr int XP_BYTES = 8192;                                          // >= 118*17*4, keeps the planes 16-byte aligned
constexpr int PL_STRIDE = CH * 2 + 16;                                  // bytes per position: 128 of channels + 16 of padding, so
                                                                        // that the 16 lanes of a ds_read_b128 group (consecutive
                                                                        // positions) fall on 16 different 16-byte bank groups
                                                                        // (at 128 they collide eight ways: measured LDS-bound)
// Taps that fall into the padding read zeros.  A single zero row would sit on the banks of one of the valid lanes
// of the same instruction (two-way conflict on three of four taps: 27 % of conv2's LDS cycles by the counters);
// instead every plane ends in a 256-byte-aligned zero region and a padding lane reads it at the offset its natural
// address has modulo 256 -- the bank slot that lane would have used anyway, which no valid lane touches.
constexpr int PL_ZERO_OFF = (CT_P2 * PL_STRIDE + 255) / 256 * 256;      // 43 008
constexpr int PL_ZERO_BYTES = 256 + 3 * 32 + 32;                        // natural offset mod 256, + channel block, + read width
constexpr int PLANE_BYTES = PL_ZERO_OFF + PL_ZERO_BYTES;                // one bf16 piece plane [pos][64 cin + pad] + zero region
constexpr int CT_LDS_BYTES = XP_BYTES + 3 * PLANE_BYTES;                // 138 368
constexpr int WIN_PLANE_BYTES = XP_H * CT_F * 16;                       // 18 880: f16-pair arithmetic, conv1's input windows per piece
constexpr int CT_NW = 8, CT_NT = CT_NW * 64;
constexpr int C1_TILES = CT_T / 3;                                      // 33 tiles of 3 time rows x 10 bins (30 of 32 columns)
constexpr int C2_TILES = (CT_P2 + 31) / 32;                             // 10
static_assert(CT_T % 3 == 0, "conv1 tiles hold whole time rows");
static_assert(C2_TILES == 10 && CT_NW == 8, "conv2 tile groups {3,3,2,2} x 2 channel tiles assume 10 tiles on 8 wavefronts");
static_assert(XP_BYTES % 256 == 0 && PL_STRIDE % 16 == 0, "the zero regions rely on 256-byte bank periodicity");
static_assert(XP_H * XP_W * 4 <= XP_BYTES, "padded input does not fit its LDS slot");

__device__ __forceinline__ float lane_up(float v) { return from_lane_above(v); }

// ---- f16-pair arithmetic (H2 = true, the default; kws_set_cnn_trad_math) ----------------------------------------
// Every f32 GEMM operand v, first scaled by a power of two s into f16's range, is written as hi + lo' * 2^-11 with
// hi = f16(v s) (round to nearest) and lo' = f16((v s - hi) * 2^11): 22 significant bits.  f16 x f16 products are exact in
// the matrix core's f32 accumulate; hi*hi goes to one accumulator, hi*lo' + lo'*hi to a second one that is scaled by
// 2^-11 at the end (the dropped lo*lo term is <= 2^-22 of the product).  THREE v_mfma_f32_32x32x16_f16 per f32 k-block
// instead of the six bf16 ones of the three-way split, two operand pieces to load instead of three -- this model is
// bound by the matrix pipe, so that is the lever -- for logits that differ from float64 by what torch's own f32 forward
// differs by (tests/test_cnntrad_f16_pair_sim.py).  Range: weights are scaled per layer at load time so that max |w| s < 2^15;
// activations per CLIP, from rigorous bounds known before the values exist: M0 = max |feature| of the clip (measured
// while staging), |conv1 out| <= max_c sum|w1[c]| M0 + max|b1|, |conv2 out| <= max_c sum|w2[c]| * that + max|b2|.  The
// bounds are loose by 2^4..2^7 per layer; f16 keeps 11 bits down to 2^-14 and the scaled values top out below 2^15, so a
// value keeps full relative precision down to 2^-29 of its layer's bound and an absolute 2^-40 of it below that: no
// overflow for any input, no precision cliff.  All scalings are by powers of two (exact).
__device__ __forceinline__ void split_pair(float even, float odd, uint32_t& hi, uint32_t& lo) {
    const halfx2 h = {(_Float16)even, (_Float16)odd};
    const halfx2 l = {(_Float16)((even - (float)h[0]) * 2048.f), (_Float16)((odd - (float)h[1]) * 2048.f)};
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, l);
}
__device__ __forceinline__ void split2(const float (&y)[8], uintx4& hi, uintx4& lo) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t h, l;
        split_pair(y[2 * i], y[2 * i + 1], h, l);
        hi[i] = h;
        lo[i] = l;
    }
}
constexpr float LO_UNSCALE = 1.f / 2048.f;
template <bool H2>
__global__ __launch_bounds__(CT_NT) void kws_cnntrad_conv_kernel(CnnTradWeights w, const float* __restrict__ feat, int B,
                                                                 float* __restrict__ conv_out, float* __restrict__ clip_scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ uint32_t m0_bits;
    float* xp = reinterpret_cast<float*>(smem);
    unsigned char* planes = smem + XP_BYTES;
    constexpr int NP = H2 ? 2 : 3;  // operand pieces
    unsigned char* win = planes + NP * PLANE_BYTES;  // H2: conv1's pre-split input windows, [piece][padded row][start column][8 x f16]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, half = lane >> 5, col = lane & 31;
    const int clip = blockIdx.x;
    if (clip >= B) return;

    // ---- stage: zero-padded input map, zero regions of the pooled planes ---------------------------------
    for (int i = tid; i < XP_H * XP_W; i += CT_NT) xp[i] = 0.f;
    if (tid < NP * PL_ZERO_BYTES / 4)
        reinterpret_cast<uint32_t*>(planes + (tid / (PL_ZERO_BYTES / 4)) * PLANE_BYTES + PL_ZERO_OFF)[tid % (PL_ZERO_BYTES / 4)] = 0u;
    if (tid == 0) m0_bits = 0u;
    __syncthreads();
    // H2: per-clip scales.  post1 / post2 take a layer's accumulators back to true units, s1 maps the pooled conv1 output
    // into f16's range (see the note at split_pair)
    float post1 = 1.f, post2 = 1.f, s1 = 1.f;
    if constexpr (H2) {
        float v[2] = {0.f, 0.f}, mx = 0.f;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int i = tid + n * CT_NT;
            if (i < CT_T * CT_F) {
                v[n] = feat[(size_t)clip * (CT_T * CT_F) + i];
                mx = fmaxf(mx, fabsf(v[n]));
            }
        }
#pragma unroll
        for (int off = 32; off; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        if (lane == 0) atomicMax(&m0_bits, __builtin_bit_cast(uint32_t, mx));  // non-negative floats order like their bits
        __syncthreads();
        const float m0 = __builtin_bit_cast(float, m0_bits);
        const int k0 = pow2_exp_for(m0);
        const float bound1 = (w.w1_abs * m0 + w.b1_max) * 1.001f;
        const float bound2 = (w.w2_abs * bound1 + w.b2_max) * 1.001f;
        const int k1 = pow2_exp_for(bound1), k2 = pow2_exp_for(bound2);
        post1 = pow2f(-k0) * w.inv_sw1;
        s1 = pow2f(k1);
        post2 = pow2f(-k1) * w.inv_sw2;
        if (tid == 0) clip_scale[clip] = pow2f(k2);
        const float s0 = pow2f(k0);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int i = tid + n * CT_NT;
            if (i < CT_T * CT_F) xp[(i / CT_F + 9) * XP_W + i % CT_F + 3] = v[n] * s0;
        }
        __syncthreads();
        // conv1's B operands, split ONCE: window (padded row r, start column f) = the eight consecutive scaled inputs a lane
        // needs for one kernel row, as one 16-byte hi and one 16-byte lo' fragment.  (Gathered and split per use it was
        // 8 LDS reads + ~30 VALU instructions per three MFMAs, and conv1 took more than a third of the kernel.)
        for (int i = tid; i < XP_H * CT_F; i += CT_NT) {
            const float* src = xp + (i / CT_F) * XP_W + i % CT_F;
            const float y[8] = {src[0], src[1], src[2], src[3], src[4], src[5], src[6], src[7]};
            uintx4 hi, lo;
            split2(y, hi, lo);
            *reinterpret_cast<uintx4*>(win + i * 16) = hi;
            *reinterpret_cast<uintx4*>(win + i * 16 + WIN_PLANE_BYTES) = lo;
        }
    } else {
        for (int i = tid; i < CT_T * CT_F; i += CT_NT)
            xp[(i / CT_F + 9) * XP_W + i % CT_F + 3] = feat[(size_t)clip * (CT_T * CT_F) + i];
    }
    __syncthreads();

    // ---- conv1 + ReLU + frequency max-pool -> pre-split planes ------------------------------------------
    // A tile = 3 time rows x 10 bins in columns 0..29.  k-block kb covers kernel rows 2kb (lanes 0..31) and 2kb+1
    // (lanes 32..63), all 8 kernel columns: the lane's eight B elements are consecutive floats of one padded row.
    // Wavefront w owns channel tile w & 1 and keeps that tile's 30 pre-split A fragments (120 registers) for the
    // whole phase -- streamed per tile they would cost 2 MB of L1 traffic per clip -- and walks tiles w>>1, +4, ...
    {
        const int ct = wv & 1;
        uintx4 af[CT_K1H / 2][NP];  // [k-block][piece]
        {
            const uintx4* asrc = reinterpret_cast<const uintx4*>(H2 ? w.c1_h2 : w.c1_split) + ct * (NP * 64) + lane;
#pragma unroll
            for (int kb = 0; kb < CT_K1H / 2; ++kb)
#pragma unroll
                for (int pc = 0; pc < NP; ++pc) af[kb][pc] = asrc[(kb * 2 * NP + pc) * 64];
        }
        float bias[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bias[r] = w.c1_b[ct * 32 + row_of(r, half)] * s1;  // H2: in the planes' scale (s1 = 1 otherwise)
        const float ps1 = post1 * s1;
        const int cc = col < 30 ? col : 29;
        const int tr = cc / CT_F, f = cc % CT_F;
        const bool owner = col < 30 && f % 3 == 0 && f < 9;
#ifdef KWS_X_CT_NO_CONV1  // timing ablation: no conv1 (the planes keep whatever LDS held)
        for (int u = C1_TILES; u < C1_TILES; u += CT_NW / 2) {
#else
        for (int u = wv >> 1; u < C1_TILES; u += CT_NW / 2) {
#endif
            const int t = 3 * u + tr;
            const float* base = xp + (t + half) * XP_W + f;
            floatx16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, acc2 = acc;  // two chains, summed below
            if constexpr (H2) {
                // B operand = the pre-split window (padded row t + 2kb + half, start column f): two aligned ds_read_b128 per
                // k-block, no VALU work; fetched two k-blocks ahead.  acc = hi * hi, acc2 = the cross terms (units of 2^-11)
                const unsigned char* wb = win + ((t + half) * CT_F + f) * 16;
                uintx4 bq[3][2];
                auto wload = [&](int kb, uintx4 (&d)[2]) {
                    d[0] = *reinterpret_cast<const uintx4*>(wb + kb * (2 * CT_F * 16));
                    d[1] = *reinterpret_cast<const uintx4*>(wb + kb * (2 * CT_F * 16) + WIN_PLANE_BYTES);
                };
                wload(0, bq[0]);
                wload(1, bq[1]);
#pragma unroll
                for (int kb = 0; kb < CT_K1H / 2; ++kb) {
                    if (kb + 2 < CT_K1H / 2) wload(kb + 2, bq[(kb + 2) % 3]);
#ifdef KWS_X_CT_NO_C1_MFMA  // timing ablation: conv1's loads and epilogue without the matrix instructions
                    acc[kb] += __builtin_bit_cast(float, bq[kb % 3][0][0] ^ bq[kb % 3][1][1] ^ af[kb][0][0] ^ af[kb][NP - 1][1]);
#else
                    acc2 = mfma_f16(af[kb][0], bq[kb % 3][1], acc2);
                    acc = mfma_f16(af[kb][0], bq[kb % 3][0], acc);
                    acc2 = mfma_f16(af[kb][NP - 1], bq[kb % 3][0], acc2);
#endif
                }
            } else {
                float y[2][8];
                auto gather = [&](int kb, float (&dst)[8]) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) dst[j] = base[kb * 2 * XP_W + j];
                };
                gather(0, y[0]);
                gather(1, y[1]);
#pragma unroll
                for (int kb = 0; kb < CT_K1H / 2; ++kb) {
                    const int cur = kb & 1;
                    uintx4 bh, bm, bl;
                    split3(y[cur], bh, bm, bl);
                    if (kb + 2 < CT_K1H / 2) gather(kb + 2, y[cur]);
                    // six piece products, smallest first
                    acc = mfma_bf16(af[kb][NP - 1], bh, acc);
                    acc2 = mfma_bf16(af[kb][0], bl, acc2);
                    acc = mfma_bf16(af[kb][1], bm, acc);
                    acc2 = mfma_bf16(af[kb][1], bh, acc2);
                    acc = mfma_bf16(af[kb][0], bm, acc);
                    acc2 = mfma_bf16(af[kb][0], bh, acc2);
                }
            }
            if constexpr (!H2) acc += acc2;
            // bias, ReLU, max over bins (f, f+1, f+2) via two lane shifts; lanes with f in {0,3,6} own a pooled value
            const int p = t * CT_FP + f / 3;
#ifdef KWS_X_CT_NO_C1_EPILOGUE  // timing ablation: conv1 without its epilogue (one store keeps the accumulators alive)
            if (acc[0] + acc[5] + acc[10] + acc[15] == 12345.f) planes[p] = 1;
            continue;
#endif
            if constexpr (H2) {
                // relu(x * post + b) * s1 is monotone in x, so the pool runs on the raw sums (two DPP maxima per value) and the
                // affine map, with s1 folded in (powers of two: exact), once on the maximum -- same bits as pooling afterwards
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    float m[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float v = acc[r + e] + acc2[r + e] * LO_UNSCALE;
                        const float a = fmaxf(v, lane_up(v));
                        m[e] = relu(fmaxf(a, lane_up(a)) * ps1 + bias[r + e]);
                    }
                    if (owner) {  // channels co, co+1 (co even) as one dword per piece
                        const int co = ct * 32 + row_of(r, half);
                        uint32_t* dst = reinterpret_cast<uint32_t*>(planes + p * PL_STRIDE + co * 2);
                        uint32_t hi, lo;
                        split_pair(m[0], m[1], hi, lo);
                        dst[0] = hi;
                        dst[PLANE_BYTES / 4] = lo;
                    }
                }
                continue;
            }
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                float m[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float v = relu(acc[r + e] + bias[r + e]);
                    const float a = lane_up(v), b = lane_up(a);
                    m[e] = fmaxf(v, fmaxf(a, b));
                }
                if (owner) {  // channels co, co+1 (co even) as one dword per piece
                    const int co = ct * 32 + row_of(r, half);
                    uint32_t* dst = reinterpret_cast<uint32_t*>(planes + p * PL_STRIDE + co * 2);
                    const float r0 = m[0] - top16(m[0]), r1 = m[1] - top16(m[1]);
                    dst[0] = pack_top16(m[0], m[1]);
                    dst[PLANE_BYTES / 4] = pack_top16(r0, r1);
                    dst[(NP - 1) * PLANE_BYTES / 4] = pack_top16(r0 - top16(r0), r1 - top16(r1));
                }
            }
        }
    }
    __syncthreads();

    // ---- conv2 + ReLU -> HBM ------------------------------------------------------------------------------
    // K = 2560 ordered (kh, kw, cin): k-block (kk = kh*4 + kw, cb) covers input channels 16cb..16cb+15 of the input
    // position (t + kh - 4, fp + kw - 1); lane (column, half) reads its eight channels 16cb + 8half .. +7 of each
    // piece with one ds_read_b128.  One unit per wavefront, a single round: wavefront w takes channel tile w & 1 and
    // a group of position tiles -- {0,1,2}, {3,4,5}, {6,7}, {8,9} for w >> 1 = 0..3 -- so that the two wavefronts of
    // a SIMD (w, w+4) carry 3 + 2 tiles, the same on every SIMD, and the A operands (983 KB of pre-split weights
    // streaming from L2 through L1, which would otherwise be as busy as the matrix pipe) are fetched once per
    // k-block for all of the wavefront's tiles.
    auto conv2_unit = [&](auto ntile_tag, int tile0) {
        constexpr int NTILE = decltype(ntile_tag)::value;
        constexpr int NACC = H2 ? 2 : 1;       // H2: [0] = hi * hi, [1] = cross terms in units of 2^-11
        constexpr int NPROD = H2 ? 3 : 6;      // piece products per k-block
        const int ct = wv & 1;
        int p[NTILE], t[NTILE], fp[NTILE];
        bool pvalid[NTILE];
        floatx16 acc[NTILE][NACC];
#pragma unroll
        for (int i = 0; i < NTILE; ++i) {
            p[i] = (tile0 + i) * 32 + col;
            pvalid[i] = p[i] < CT_P2;
            t[i] = p[i] / CT_FP;  // columns past the map keep their natural geometry: they read zeros from their own bank slot
            fp[i] = p[i] % CT_FP;
#pragma unroll
            for (int a = 0; a < NACC; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][a][r] = 0.f;
        }
        const uintx4* asrc = reinterpret_cast<const uintx4*>(H2 ? w.c2_h2 : w.c2_split) + ct * (NP * 64) + lane;
        // Pipeline at k-block granularity: the A operands of (kk+1, cb) are requested right after those of (kk, cb)
        // have been used (four k-blocks ahead of their use, an L2 round trip); the B operands of the next k-block
        // are read from LDS one k-block ahead.
        uintx4 af[4][NP];         // [cb][piece] of this wavefront's channel tile
        uintx4 bf[2][NTILE][NP];  // [parity of the k-block][tile][piece]
        auto a_index = [&](int kk, int cb, int pc2) { return (((size_t)kk * 4 + cb) * 2 * NP + pc2) * 64; };
        auto a_load = [&](int kk, int cb) {
#pragma unroll
            for (int pc2 = 0; pc2 < NP; ++pc2) af[cb][pc2] = asrc[a_index(kk, cb, pc2)];
        };
        auto b_addr = [&](int kk, int i) -> const unsigned char* {
            const int kh = kk >> 2, kw = kk & 3;
            const int tin = t[i] + kh - 4, fin = fp[i] + kw - 1;
            const bool ok = pvalid[i] && (unsigned)tin < (unsigned)CT_T && (unsigned)fin < (unsigned)CT_FP;
            const int natural = (tin * CT_FP + fin) * PL_STRIDE + half * 16;  // may lie outside the plane
            return planes + (ok ? natural : PL_ZERO_OFF + (natural & 255));
        };
        auto b_load = [&](const unsigned char* const (&ba)[NTILE], int cb, uintx4 (&dst)[NTILE][NP]) {
#pragma unroll
            for (int i = 0; i < NTILE; ++i)
#pragma unroll
                for (int pc2 = 0; pc2 < NP; ++pc2) dst[i][pc2] = *reinterpret_cast<const uintx4*>(ba[i] + cb * 32 + pc2 * PLANE_BYTES);
        };
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) a_load(0, cb);
        const unsigned char* ba[NTILE];
#pragma unroll
        for (int i = 0; i < NTILE; ++i) ba[i] = b_addr(0, i);
        b_load(ba, 0, bf[0]);
#ifdef KWS_X_CT_NO_CONV2  // timing ablation: no conv2 loop
        for (int kk = CT_K2H * CT_K2W; kk < CT_K2H * CT_K2W; ++kk) {
#else
        for (int kk = 0; kk < CT_K2H * CT_K2W; ++kk) {
#endif
            const unsigned char* ba_next[NTILE];
#pragma unroll
            for (int i = 0; i < NTILE; ++i) ba_next[i] = b_addr(kk + 1 < CT_K2H * CT_K2W ? kk + 1 : kk, i);
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                const int cur = cb & 1;
                const bool more = kk + 1 < CT_K2H * CT_K2W;
                // The loads of the next k-block are spread between the MFMAs (one load behind each matrix
                // instruction, pinned with scheduling barriers): issued in a lump after them, the matrix pipe drains
                // while the wavefront works through a dozen memory instructions.  An A piece is re-requested for
                // (kk+1, cb) right behind its last use.
#pragma unroll
                for (int q = 0; q < NPROD; ++q) {  // piece products, smallest first, tiles interleaved
                    // three-way bf16: (lo,hi) (hi,lo) (mid,mid) (mid,hi) (hi,mid) (hi,hi); f16 pair: (hi,lo') (hi,hi) (lo',hi)
                    const int pa = H2 ? (q == 2 ? 1 : 0) : (q == 0 ? 2 : (q == 2 || q == 3) ? 1 : 0);
                    const int pb = H2 ? (q == 0 ? 1 : 0) : ((q == 0 || q == 3 || q == 5) ? 0 : (q == 1 ? 2 : 1));
                    const int dst = H2 ? (q == 1 ? 0 : 1) : 0;
#pragma unroll
                    for (int i = 0; i < NTILE; ++i) {
                        if constexpr (H2)
                            acc[i][dst] = mfma_f16(af[cb][pa], bf[cur][i][pb], acc[i][dst]);
                        else
                            acc[i][dst] = mfma_bf16(af[cb][pa], bf[cur][i][pb], acc[i][dst]);
                        __builtin_amdgcn_sched_barrier(0);
                        const int n = q * NTILE + i;  // one B load (tile n / NP, piece n % NP) of the next k-block per MFMA
#ifdef KWS_X_CT_NO_BLOAD  // timing ablation: B operands are not refreshed
                        if (n < 0) {
#else
                        if (n < NP * NTILE) {
#endif
                            const unsigned char* src = (cb < 3 ? ba[n / NP] + (cb + 1) * 32 : ba_next[n / NP]) + (n % NP) * PLANE_BYTES;
                            bf[cur ^ 1][n / NP][n % NP] = *reinterpret_cast<const uintx4*>(src);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    // last use of an A piece: three-way lo after product 0, mid after 3, hi after 5; pair hi after 1, lo' after 2
                    const int done = H2 ? (q == 1 ? 0 : (q == 2 ? 1 : -1)) : (q == 0 ? 2 : (q == 3 ? 1 : (q == 5 ? 0 : -1)));
#ifdef KWS_X_CT_NO_ALOAD  // timing ablation: A operands are not refreshed
                    if (false) {
#else
                    if (more && done >= 0) {
#endif
                        af[cb][done] = asrc[a_index(kk + 1, cb, done)];
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < NTILE; ++i) ba[i] = ba_next[i];
        }
#pragma unroll
        for (int i = 0; i < NTILE; ++i) {
            if (pvalid[i]) {
                float* o = conv_out + (size_t)clip * (CH * CT_P2) + p[i];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = ct * 32 + row_of(r, half);
                    float v = acc[i][0][r];
                    if constexpr (H2) v = (v + acc[i][NACC - 1][r] * LO_UNSCALE) * post2;
                    o[co * CT_P2] = relu(v + w.c2_b[co]);
                }
            }
        }
    };
    {
        const int g = wv >> 1;
        if (g < 2)
            conv2_unit(std::integral_constant<int, 3>{}, 3 * g);
        else
            conv2_unit(std::integral_constant<int, 2>{}, 6 + 2 * (g - 2));
    }
}

// Kernel B: Linear(19008 -> 32) ; Linear(32 -> 128) + ReLU ; Linear(128 -> C) ; argmax.
// The first layer is a [clips x 19008] x [19008 x 32] GEMM: 16 clips per workgroup (in the 32 MFMA rows), D[clip][output] on the bf16
// matrix pipe with the exact split (activations split on the fly, weights pre-split as B operands), K divided
// among the 12 wavefronts of the workgroup (99 k-blocks of 16 each) and the partial sums combined through LDS.
// It reads the 76 KB per clip that kernel A wrote: HBM-bound, so loads run three k-blocks ahead in every wave.
constexpr int CT_FLAT = CH * CT_P2;  // 19008
constexpr int CT_LIN = 32, CT_DNN = 128;
constexpr int DN_WAVES = 12, DN_KB = CT_FLAT / 16 / DN_WAVES;  // 99 k-blocks per wavefront
constexpr int DN_CLIPS = 16;  // clips per workgroup: 4096 clips = 256 workgroups, one per CU (32 leaves half the CUs idle: 2.31 vs 2.27 ms for the model; 8 is slower again, 2.36)
static_assert(DN_KB * DN_WAVES * 16 == CT_FLAT, "K must divide evenly among the wavefronts");
template <bool H2>
__global__ __launch_bounds__(DN_WAVES * 64) void kws_cnntrad_dense_kernel(CnnTradWeights w, const float* __restrict__ conv_out,
                                                                          const float* __restrict__ clip_scale, int B,
                                                                          float* __restrict__ logits, int32_t* __restrict__ label) {
    constexpr int NP = H2 ? 2 : 3;
    __shared__ float part[DN_WAVES][32 * 32];  // partial D of every wavefront, [clip][output]
    __shared__ float h1[32][CT_LIN];
    __shared__ float h2[32][CT_DNN];
    __shared__ float lg[32][MAX_CLASSES];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, half = lane >> 5, col = lane & 31;
    const int clip0 = blockIdx.x * DN_CLIPS;
    {
        // A operand: lane (row = clip slot col, half) holds x[clip][16kb + 8half .. +7]
        // the 32 MFMA rows hold DN_CLIPS clips (the upper rows repeat them: same addresses, L1 hits, results unused)
        const int aslot = col % DN_CLIPS;
        const int aclip = clip0 + aslot < B ? clip0 + aslot : B - 1;
        const float4* xa = reinterpret_cast<const float4*>(conv_out + (size_t)aclip * CT_FLAT + (size_t)wv * DN_KB * 16 + 8 * half);
        // B operand: pre-split weights [kb][piece][lane]
        const uintx4* wb = reinterpret_cast<const uintx4*>(H2 ? w.lin_h2 : w.lin_split) + (size_t)wv * DN_KB * NP * 64 + lane;
        floatx16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, accx = acc;
        const float s2 = H2 ? clip_scale[aclip] : 1.f;  // H2: the clip's power-of-two scale into f16's range (written by the conv kernel)
        constexpr int DEPTH = 3;
        float4 xr[DEPTH][2];
        uintx4 wr[DEPTH][NP];
        auto load = [&](int kb, int slot) {
            xr[slot][0] = xa[kb * 4];
            xr[slot][1] = xa[kb * 4 + 1];
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) wr[slot][pc] = wb[(kb * NP + pc) * 64];
        };
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) load(d, d);
        for (int kb0 = 0; kb0 < DN_KB; kb0 += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int kb = kb0 + d;
                if constexpr (H2) {
                    const float y[8] = {xr[d][0].x * s2, xr[d][0].y * s2, xr[d][0].z * s2, xr[d][0].w * s2,
                                        xr[d][1].x * s2, xr[d][1].y * s2, xr[d][1].z * s2, xr[d][1].w * s2};
                    uintx4 ah, al;
                    split2(y, ah, al);
                    const uintx4 bh = wr[d][0], bl = wr[d][NP - 1];
                    if (kb + DEPTH < DN_KB) load(kb + DEPTH, d);
                    accx = mfma_f16(ah, bl, accx);
                    acc = mfma_f16(ah, bh, acc);
                    accx = mfma_f16(al, bh, accx);
                } else {
                    const float y[8] = {xr[d][0].x, xr[d][0].y, xr[d][0].z, xr[d][0].w, xr[d][1].x, xr[d][1].y, xr[d][1].z, xr[d][1].w};
                    uintx4 ah, am, al;
                    split3(y, ah, am, al);
                    const uintx4 bh = wr[d][0], bm = wr[d][1], bl = wr[d][NP - 1];
                    if (kb + DEPTH < DN_KB) load(kb + DEPTH, d);
                    acc = mfma_bf16(al, bh, acc);
                    acc = mfma_bf16(ah, bl, acc);
                    acc = mfma_bf16(am, bm, acc);
                    acc = mfma_bf16(am, bh, acc);
                    acc = mfma_bf16(ah, bm, acc);
                    acc = mfma_bf16(ah, bh, acc);
                }
            }
        }
        if constexpr (H2) acc += accx * LO_UNSCALE;
#pragma unroll
        for (int r = 0; r < 16; ++r) part[wv][row_of(r, half) * 32 + col] = acc[r];  // D row = clip slot, column = output
    }
    __syncthreads();
    for (int i = tid; i < DN_CLIPS * CT_LIN; i += DN_WAVES * 64) {
        float a = H2 ? 0.f : w.lin_b[i & 31];
#pragma unroll
        for (int k = 0; k < DN_WAVES; ++k) a += part[k][i];
        if constexpr (H2) {  // back to true units: the clip's scale and the layer's weight scale are powers of two
            const int cl = clip0 + (i >> 5) < B ? clip0 + (i >> 5) : B - 1;
            a = a * (w.inv_swl / clip_scale[cl]) + w.lin_b[i & 31];
        }
        h1[i >> 5][i & 31] = a;
    }
    __syncthreads();
    for (int i = tid; i < DN_CLIPS * CT_DNN; i += DN_WAVES * 64) {
        const int s = i / CT_DNN, j = i % CT_DNN;
        float a = w.dnn_b[j];
        for (int k = 0; k < CT_LIN; ++k) a = fmaf(h1[s][k], w.dnn_w[j * CT_LIN + k], a);
        h2[s][j] = a > 0.f ? a : 0.f;
    }
    __syncthreads();
    const int C = w.num_classes;
    for (int i = tid; i < DN_CLIPS * C; i += DN_WAVES * 64) {
        const int s = i / C, c = i % C;
        float a = w.fc_b[c];
        for (int k = 0; k < CT_DNN; ++k) a = fmaf(h2[s][k], w.fc_w[c * CT_DNN + k], a);
        lg[s][c] = a;
        if (clip0 + s < B) logits[(size_t)(clip0 + s) * C + c] = a;
    }
    __syncthreads();
    if (tid < DN_CLIPS && clip0 + tid < B && label) {
        int arg = 0;
        float best = lg[tid][0];
        for (int c = 1; c < C; ++c)
            if (lg[tid][c] > best) {
                best = lg[tid][c];
                arg = c;
            }
        label[clip0 + tid] = arg;
    }
}

}  // namespace

constexpr int CT_LDS_BYTES_H2 = XP_BYTES + 2 * PLANE_BYTES + 2 * WIN_PLANE_BYTES;  // 132 800

hipError_t cnntrad_init_device() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kws_cnntrad_conv_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       CT_LDS_BYTES);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kws_cnntrad_conv_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               CT_LDS_BYTES_H2);
}

// d_conv_ws: B * 19008 floats of conv2 output, then B floats of per-clip scales (f16-pair arithmetic only)
hipError_t launch_cnntrad_conv(hipStream_t s, const CnnTradWeights& w, const float* d_feat, int B, float* d_conv_ws, bool f16_pair) {
    float* scale = d_conv_ws + (size_t)B * CT_FLAT;
    if (f16_pair)
        hipLaunchKernelGGL(kws_cnntrad_conv_kernel<true>, dim3(B), dim3(CT_NT), CT_LDS_BYTES_H2, s, w, d_feat, B, d_conv_ws, scale);
    else
        hipLaunchKernelGGL(kws_cnntrad_conv_kernel<false>, dim3(B), dim3(CT_NT), CT_LDS_BYTES, s, w, d_feat, B, d_conv_ws, scale);
    return hipGetLastError();
}

hipError_t launch_cnntrad_dense(hipStream_t s, const CnnTradWeights& w, const float* d_conv_ws, int B, float* d_logits,
                                int32_t* d_label, bool f16_pair) {
    const float* scale = d_conv_ws + (size_t)B * CT_FLAT;
    const dim3 grid((B + DN_CLIPS - 1) / DN_CLIPS), block(DN_WAVES * 64);
    if (f16_pair)
        hipLaunchKernelGGL(kws_cnntrad_dense_kernel<true>, grid, block, 0, s, w, d_conv_ws, scale, B, d_logits, d_label);
    else
        hipLaunchKernelGGL(kws_cnntrad_dense_kernel<false>, grid, block, 0, s, w, d_conv_ws, scale, B, d_logits, d_label);
    return hipGetLastError();
}

}  // namespace kws
