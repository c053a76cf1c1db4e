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
constexpr int CT_NW = 8, CT_NT = CT_NW * 64;
constexpr int C1_TILES = CT_T / 3;                                      // 33 tiles of 3 time rows x 10 bins (30 of 32 columns)
constexpr int C2_TILES = (CT_P2 + 31) / 32;                             // 10
static_assert(CT_T % 3 == 0, "conv1 tiles hold whole time rows");
static_assert(C2_TILES == 10 && CT_NW == 8, "conv2 tile groups {3,3,2,2} x 2 channel tiles assume 10 tiles on 8 wavefronts");
static_assert(XP_BYTES % 256 == 0 && PL_STRIDE % 16 == 0, "the zero regions rely on 256-byte bank periodicity");
static_assert(XP_H * XP_W * 4 <= XP_BYTES, "padded input does not fit its LDS slot");

__device__ __forceinline__ float lane_up(float v) { return from_lane_above(v); }

__global__ __launch_bounds__(CT_NT) void kws_cnntrad_conv_kernel(CnnTradWeights w, const float* __restrict__ feat, int B,
                                                                 float* __restrict__ conv_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* xp = reinterpret_cast<float*>(smem);
    unsigned char* planes = smem + XP_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, half = lane >> 5, col = lane & 31;
    const int clip = blockIdx.x;
    if (clip >= B) return;

    // ---- stage: zero-padded input map, zero regions of the pooled planes ---------------------------------
    for (int i = tid; i < XP_H * XP_W; i += CT_NT) xp[i] = 0.f;
    if (tid < 3 * PL_ZERO_BYTES / 4)
        reinterpret_cast<uint32_t*>(planes + (tid / (PL_ZERO_BYTES / 4)) * PLANE_BYTES + PL_ZERO_OFF)[tid % (PL_ZERO_BYTES / 4)] = 0u;
    __syncthreads();
    for (int i = tid; i < CT_T * CT_F; i += CT_NT)
        xp[(i / CT_F + 9) * XP_W + i % CT_F + 3] = feat[(size_t)clip * (CT_T * CT_F) + i];
    __syncthreads();

    // ---- conv1 + ReLU + frequency max-pool -> pre-split planes ------------------------------------------
    // A tile = 3 time rows x 10 bins in columns 0..29.  k-block kb covers kernel rows 2kb (lanes 0..31) and 2kb+1
    // (lanes 32..63), all 8 kernel columns: the lane's eight B elements are consecutive floats of one padded row.
    // Wavefront w owns channel tile w & 1 and keeps that tile's 30 pre-split A fragments (120 registers) for the
    // whole phase -- streamed per tile they would cost 2 MB of L1 traffic per clip -- and walks tiles w>>1, +4, ...
    {
        const int ct = wv & 1;
        uintx4 af[CT_K1H / 2][3];  // [k-block][piece]
        {
            const uintx4* asrc = reinterpret_cast<const uintx4*>(w.c1_split) + ct * (3 * 64) + lane;
#pragma unroll
            for (int kb = 0; kb < CT_K1H / 2; ++kb)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) af[kb][pc] = asrc[(kb * 2 * 3 + pc) * 64];
        }
        float bias[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bias[r] = w.c1_b[ct * 32 + row_of(r, half)];
        const int cc = col < 30 ? col : 29;
        const int tr = cc / CT_F, f = cc % CT_F;
        const bool owner = col < 30 && f % 3 == 0 && f < 9;
        for (int u = wv >> 1; u < C1_TILES; u += CT_NW / 2) {
            const int t = 3 * u + tr;
            const float* base = xp + (t + half) * XP_W + f;
            floatx16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, acc2 = acc;  // two chains, summed below
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
                acc = mfma_bf16(af[kb][2], bh, acc);
                acc2 = mfma_bf16(af[kb][0], bl, acc2);
                acc = mfma_bf16(af[kb][1], bm, acc);
                acc2 = mfma_bf16(af[kb][1], bh, acc2);
                acc = mfma_bf16(af[kb][0], bm, acc);
                acc2 = mfma_bf16(af[kb][0], bh, acc2);
            }
            acc += acc2;
            // bias, ReLU, max over bins (f, f+1, f+2) via two lane shifts; lanes with f in {0,3,6} own a pooled value
            const int p = t * CT_FP + f / 3;
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
                    const float r0 = m[0] - top16(m[0]), r1 = m[1] - top16(m[1]);
                    uint32_t* dst = reinterpret_cast<uint32_t*>(planes + p * PL_STRIDE + co * 2);
                    dst[0] = pack_top16(m[0], m[1]);
                    dst[PLANE_BYTES / 4] = pack_top16(r0, r1);
                    dst[2 * PLANE_BYTES / 4] = pack_top16(r0 - top16(r0), r1 - top16(r1));
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
        const int ct = wv & 1;
        int p[NTILE], t[NTILE], fp[NTILE];
        bool pvalid[NTILE];
        floatx16 acc[NTILE];
#pragma unroll
        for (int i = 0; i < NTILE; ++i) {
            p[i] = (tile0 + i) * 32 + col;
            pvalid[i] = p[i] < CT_P2;
            t[i] = p[i] / CT_FP;  // columns past the map keep their natural geometry: they read zeros from their own bank slot
            fp[i] = p[i] % CT_FP;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        }
        const uintx4* asrc = reinterpret_cast<const uintx4*>(w.c2_split) + ct * (3 * 64) + lane;
        // Pipeline at k-block granularity: the A operands of (kk+1, cb) are requested right after those of (kk, cb)
        // have been used (four k-blocks ahead of their use, an L2 round trip); the B operands of the next k-block
        // are read from LDS one k-block ahead.
        uintx4 af[4][3];         // [cb][piece] of this wavefront's channel tile
        uintx4 bf[2][NTILE][3];  // [parity of the k-block][tile][piece]
        auto a_load = [&](int kk, int cb) {
#pragma unroll
            for (int pc2 = 0; pc2 < 3; ++pc2) af[cb][pc2] = asrc[(((size_t)kk * 4 + cb) * 2 * 3 + pc2) * 64];
        };
        auto b_addr = [&](int kk, int i) -> const unsigned char* {
            const int kh = kk >> 2, kw = kk & 3;
            const int tin = t[i] + kh - 4, fin = fp[i] + kw - 1;
            const bool ok = pvalid[i] && (unsigned)tin < (unsigned)CT_T && (unsigned)fin < (unsigned)CT_FP;
            const int natural = (tin * CT_FP + fin) * PL_STRIDE + half * 16;  // may lie outside the plane
            return planes + (ok ? natural : PL_ZERO_OFF + (natural & 255));
        };
        auto b_load = [&](const unsigned char* const (&ba)[NTILE], int cb, uintx4 (&dst)[NTILE][3]) {
#pragma unroll
            for (int i = 0; i < NTILE; ++i)
#pragma unroll
                for (int pc2 = 0; pc2 < 3; ++pc2) dst[i][pc2] = *reinterpret_cast<const uintx4*>(ba[i] + cb * 32 + pc2 * PLANE_BYTES);
        };
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) a_load(0, cb);
        const unsigned char* ba[NTILE];
#pragma unroll
        for (int i = 0; i < NTILE; ++i) ba[i] = b_addr(0, i);
        b_load(ba, 0, bf[0]);
        for (int kk = 0; kk < CT_K2H * CT_K2W; ++kk) {
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
                // (kk+1, cb) right behind its last use: lo after product 0, mid after product 3, hi after product 5.
#pragma unroll
                for (int q = 0; q < 6; ++q) {  // six piece products, smallest first, tiles interleaved
                    const int pa = q == 0 ? 2 : (q == 2 || q == 3) ? 1 : 0;
                    const int pb = (q == 0 || q == 3 || q == 5) ? 0 : (q == 1 ? 2 : 1);
#pragma unroll
                    for (int i = 0; i < NTILE; ++i) {
                        acc[i] = mfma_bf16(af[cb][pa], bf[cur][i][pb], acc[i]);
                        __builtin_amdgcn_sched_barrier(0);
                        const int n = q * NTILE + i;  // one B load (tile n / 3, piece n % 3) of the next k-block per MFMA
                        if (n < 3 * NTILE) {
                            const unsigned char* src = (cb < 3 ? ba[n / 3] + (cb + 1) * 32 : ba_next[n / 3]) + (n % 3) * PLANE_BYTES;
                            bf[cur ^ 1][n / 3][n % 3] = *reinterpret_cast<const uintx4*>(src);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    if (more && (q == 0 || q == 3 || q == 5)) {
                        const int pc2 = q == 0 ? 2 : q == 3 ? 1 : 0;
                        af[cb][pc2] = asrc[(((size_t)(kk + 1) * 4 + cb) * 2 * 3 + pc2) * 64];
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
                    o[co * CT_P2] = relu(acc[i][r] + w.c2_b[co]);
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
__global__ __launch_bounds__(DN_WAVES * 64) void kws_cnntrad_dense_kernel(CnnTradWeights w, const float* __restrict__ conv_out, int B,
                                                                          float* __restrict__ logits, int32_t* __restrict__ label) {
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
        const uintx4* wb = reinterpret_cast<const uintx4*>(w.lin_split) + (size_t)wv * DN_KB * 3 * 64 + lane;
        floatx16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        constexpr int DEPTH = 3;
        float4 xr[DEPTH][2];
        uintx4 wr[DEPTH][3];
        auto load = [&](int kb, int slot) {
            xr[slot][0] = xa[kb * 4];
            xr[slot][1] = xa[kb * 4 + 1];
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) wr[slot][pc] = wb[(kb * 3 + pc) * 64];
        };
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) load(d, d);
        for (int kb0 = 0; kb0 < DN_KB; kb0 += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int kb = kb0 + d;
                const float y[8] = {xr[d][0].x, xr[d][0].y, xr[d][0].z, xr[d][0].w, xr[d][1].x, xr[d][1].y, xr[d][1].z, xr[d][1].w};
                uintx4 ah, am, al;
                split3(y, ah, am, al);
                const uintx4 bh = wr[d][0], bm = wr[d][1], bl = wr[d][2];
                if (kb + DEPTH < DN_KB) load(kb + DEPTH, d);
                acc = mfma_bf16(al, bh, acc);
                acc = mfma_bf16(ah, bl, acc);
                acc = mfma_bf16(am, bm, acc);
                acc = mfma_bf16(am, bh, acc);
                acc = mfma_bf16(ah, bm, acc);
                acc = mfma_bf16(ah, bh, acc);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) part[wv][row_of(r, half) * 32 + col] = acc[r];  // D row = clip slot, column = output
    }
    __syncthreads();
    for (int i = tid; i < DN_CLIPS * CT_LIN; i += DN_WAVES * 64) {
        float a = w.lin_b[i & 31];
#pragma unroll
        for (int k = 0; k < DN_WAVES; ++k) a += part[k][i];
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

hipError_t cnntrad_init_device() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kws_cnntrad_conv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               CT_LDS_BYTES);
}

hipError_t launch_cnntrad_conv(hipStream_t s, const CnnTradWeights& w, const float* d_feat, int B, float* d_conv_ws) {
    hipLaunchKernelGGL(kws_cnntrad_conv_kernel, dim3(B), dim3(CT_NT), CT_LDS_BYTES, s, w, d_feat, B, d_conv_ws);
    return hipGetLastError();
}

hipError_t launch_cnntrad_dense(hipStream_t s, const CnnTradWeights& w, const float* d_conv_ws, int B, float* d_logits,
                                int32_t* d_label) {
    hipLaunchKernelGGL(kws_cnntrad_dense_kernel, dim3((B + DN_CLIPS - 1) / DN_CLIPS), dim3(DN_WAVES * 64), 0, s, w, d_conv_ws, B, d_logits, d_label);
    return hipGetLastError();
}

}  // namespace kws
