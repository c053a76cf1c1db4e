// DS-CNN forward for gfx950 (MI355X): one 512-thread workgroup per clip, every activation resident in
// the CU's 160 KiB LDS, pointwise 1x1 convolutions and conv1 on the exact-f32 matrix cores
// (v_mfma_f32_32x32x2_f32), depthwise 3x3 on the VALU straight into the MFMA B-operand registers.
//
// Replaces DepthwiseSeparableConv.forward (reference kws/libs/models.py:160-183; rows a9-a15 of
// SURVEY.md section 8) for the [1,99,10] MFCC map:
//   conv1  1->64, 10x10, stride 2, pad 2, ReLU                      -> 64 x 47 x 3
//   4 x { depthwise 3x3 pad 1 ; pointwise 1x1 *padding=1* ; ReLU }  -> 64 x (49x5, 51x7, 53x9, 55x11)
//   global average pool, Linear(64 -> C), argmax (first maximum wins)
//
// The reference's 1x1 convolution with padding=1 surrounds each block's output with a ring equal to
// relu(bias) (models.py:104-106).  The ring is never stored: each channel plane in LDS holds only the
// "interior" H x W values followed by two extra slots, [P] = relu(bias[c]) and [P+1] = 0.  A depthwise
// tap that falls on the ring reads slot P, one that falls outside the padded map reads slot P+1, so
// the 3x3 stencil is nine unconditional LDS reads at per-lane precomputed addresses.
//
// MFMA mapping (32x32x2, D[i][j] += A[i][k] * B[k][j]): i = output channel, j = position, k = input
// channel.  Lane l supplies A[i = l&31][k = l>>5] (weights, held in registers for the whole block) and
// B[k = l>>5][j = l&31]: lane l therefore computes the depthwise output of position l&31 for the
// input channels 2s + (l>>5), s = 0..31, and feeds it to the matrix core without touching LDS.
// D: column = lane&31 (position), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (output channel).
//
// LDS map (floats): planes are channel-major [64][P+2]
//   Z3 (block3 out, 51x7)  @ 0      .. 22976     Z2 (block2 out, 49x5) @ 22976 .. 38784
//   Z1 (block1 out, 47x3)  @ 0      .. 9152      Z0 (conv1 out, 47x3)  @ 9152  .. 18304
//   padded MFCC 103x14     @ 18304  .. 19746     (conv1 phase only)
//   misc                   @ 38784  .. 40960     depthwise table, pool scratch
#include "kws_internal.h"

namespace kws {
namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int NW = 8;          // wavefronts per workgroup
constexpr int NT = NW * 64;    // 512 threads

constexpr int P0 = C1_H * C1_W;                  // 141
constexpr int FEAT_H = 103, FEAT_W = 14;         // MFCC zero-padded by 2 (top/left) and up to the conv1 reach
constexpr int OFF_Z3 = 0, OFF_Z2 = 22976, OFF_Z1 = 0, OFF_Z0 = 9152, OFF_FEAT = 18304;
constexpr int OFF_DWTAB = 38784;                 // [64][12]
constexpr int OFF_POOLBUF = OFF_DWTAB + 768;     // [NW][64]
constexpr int OFF_POOLED = OFF_POOLBUF + NW * 64;// [64]
constexpr int OFF_PWB = OFF_POOLED + 64;         // [64] pointwise bias of the running block
constexpr int LDS_FLOATS = 40960;                // 160 KiB
static_assert(OFF_PWB + 64 <= LDS_FLOATS, "LDS overflow");
static_assert(OFF_FEAT + FEAT_H * FEAT_W <= OFF_Z2, "feature pad overlaps Z2");

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
    static constexpr int TILES = (POUT + 31) / 32;
};

__device__ __forceinline__ float relu(float x) { return x > 0.f ? x : 0.f; }
__device__ __forceinline__ int row_of(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// ------------------------------------------------------------------------------------------------
// conv1: D[cout][pos] = sum_k W[cout][k] * im2col[k][pos], k = kh*10 + kw, as 50 MFMA k-steps.
template <bool MFMA>
__device__ __forceinline__ void conv1_phase(const DscnnWeights& w, float* lds, int tid) {
    const float* featp = lds + OFF_FEAT;
    float* z0 = lds + OFF_Z0;
    if constexpr (MFMA) {
        const int lane = tid & 63, wv = tid >> 6, half = lane >> 5, col = lane & 31;
        const int ct = wv & 1;  // units u = wv, wv + 8 share the output-channel tile
        float a[50];
#pragma unroll
        for (int s = 0; s < 50; ++s) a[s] = w.c1_w[(2 * s + half) * CH + ct * 32 + col];
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
                    z0[co * (P0 + 2) + pos] = relu(acc[r] + w.c1_b[co]);
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
            z0[co * (P0 + 2) + pos] = relu(acc);
        }
    }
    if (tid < CH) {  // extra slots of the conv1 planes: no ring in block 1, slot P+1 is the zero pad
        z0[tid * (P0 + 2) + P0] = 0.f;
        z0[tid * (P0 + 2) + P0 + 1] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------------
// One depthwise-separable block.  psum (block 4 only): per-lane partial sums of relu outputs for the
// global average pool, indexed like the MFMA accumulators ([ct][r]) or by output channel (VALU path).
template <int N, bool MFMA>
__device__ __forceinline__ void block_phase(const DscnnWeights& w, float* lds, int tid) {
    using G = Blk<N>;
    const int lane = tid & 63, wv = tid >> 6, half = lane >> 5, col = lane & 31;
    float* zout = lds + G::OFF_OUT;
    float* dwtab = lds + OFF_DWTAB;
    float* poolbuf = lds + OFF_POOLBUF;
    const float* pw_w = w.pw_w + (N - 1) * CH * CH;
    const float* pw_b = w.pw_b + (N - 1) * CH;

    // stage this block's depthwise table [64][12]; ring / zero slots of the output planes
    for (int i = tid; i < CH * 12; i += NT) dwtab[i] = w.dw_w[(N - 1) * CH * 12 + i];
    if (tid < CH) {
        const float b = pw_b[tid];
        lds[OFF_PWB + tid] = b;
        if (N < 4) {
            zout[tid * G::SOUT + G::POUT] = relu(b);
            zout[tid * G::SOUT + G::POUT + 1] = 0.f;
        }
    }
    __syncthreads();

    // pointwise weights / bias for this lane
    float wa[2][32];
    if constexpr (MFMA) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
            for (int s = 0; s < 32; ++s) wa[ct][s] = pw_w[(2 * s + half) * CH + ct * 32 + col];
        }
    }
    // accumulator rows 4q..4q+3 of tile ct are output channels ct*32 + 8q + 4*half + (0..3): one float4
    const float4* bias4 = reinterpret_cast<const float4*>(lds + OFF_PWB) + half;
    float psum[2][16];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) psum[ct][r] = 0.f;
    for (int t = wv; t < G::TILES; t += NW) {
        const int pos = t * 32 + col;
        const bool valid = pos < G::POUT;
        const int posc = valid ? pos : G::POUT - 1;
        const int h = posc / G::W, x = posc % G::W;
        // nine tap addresses (float index inside this lane's first channel plane)
        int ta[9];
#pragma unroll
        for (int dh = -1; dh <= 1; ++dh) {
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int o = G::RING ? 1 : 0;
                const int hh = h + dh - o, xx = x + dx - o;
                const bool inside = (unsigned)hh < (unsigned)G::HI && (unsigned)xx < (unsigned)G::WI;
                const bool in_map = (unsigned)(h + dh) < (unsigned)G::H && (unsigned)(x + dx) < (unsigned)G::W;
                int a = inside ? hh * G::WI + xx : ((G::RING && in_map) ? G::PIN : G::PIN + 1);
                ta[(dh + 1) * 3 + (dx + 1)] = a + half * G::SIN;
            }
        }
        // Two pointer sets (channel pairs 0..15 and 16..31) keep every ds_read inside the 64 KiB
        // immediate-offset window; the empty asm stops the compiler from re-deriving one base per step.
        int tlo[9], thi[9];  // float indices into lds
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            tlo[i] = G::OFF_IN + ta[i];
            thi[i] = tlo[i] + 32 * G::SIN;
            asm volatile("" : "+v"(tlo[i]));
            asm volatile("" : "+v"(thi[i]));
        }
        // depthwise 3x3 (+bias) of channel 2s + half at this lane's position -> one MFMA B operand.
        // Split into "issue the 12 LDS reads" and "9 FMAs" so the reads of step s+1 can be in flight
        // while step s is consumed.
        const float4* dwt4 = reinterpret_cast<const float4*>(dwtab) + half * 3;
        struct Taps {
            float4 q0, q1, q2;
            float x[9];
        };
        auto dw_load = [&](int s, Taps& t) {
            t.q0 = dwt4[s * 6 + 0];
            t.q1 = dwt4[s * 6 + 1];
            t.q2 = dwt4[s * 6 + 2];
            const int* tp = s < 16 ? tlo : thi;
            const int o = 2 * (s & 15) * G::SIN;
#pragma unroll
            for (int i = 0; i < 9; ++i) t.x[i] = lds[tp[i] + o];
        };
        auto dw_eval = [&](const Taps& t) -> float {
            float acc = t.q2.y;  // bias
            acc = fmaf(t.q0.x, t.x[0], acc);
            acc = fmaf(t.q0.y, t.x[1], acc);
            acc = fmaf(t.q0.z, t.x[2], acc);
            acc = fmaf(t.q0.w, t.x[3], acc);
            acc = fmaf(t.q1.x, t.x[4], acc);
            acc = fmaf(t.q1.y, t.x[5], acc);
            acc = fmaf(t.q1.z, t.x[6], acc);
            acc = fmaf(t.q1.w, t.x[7], acc);
            acc = fmaf(t.q2.x, t.x[8], acc);
            return acc;
        };

        if constexpr (MFMA) {
            floatx16 acc0, acc1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 b0 = bias4[2 * q], b1 = bias4[8 + 2 * q];
                acc0[4 * q + 0] = b0.x; acc0[4 * q + 1] = b0.y; acc0[4 * q + 2] = b0.z; acc0[4 * q + 3] = b0.w;
                acc1[4 * q + 0] = b1.x; acc1[4 * q + 1] = b1.y; acc1[4 * q + 2] = b1.z; acc1[4 * q + 3] = b1.w;
            }
            // software pipeline: the LDS reads of step s+1 are issued before step s is evaluated, and
            // the matrix core works on step s while the VALU/LDS side runs ahead
            Taps tp0, tp1;
            dw_load(0, tp0);
#pragma unroll
            for (int s = 0; s < 32; s += 2) {
                dw_load(s + 1, tp1);
                const float y0 = dw_eval(tp0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[0][s], y0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[1][s], y0, acc1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (s + 2 < 32) dw_load(s + 2, tp0);
                const float y1 = dw_eval(tp1);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[0][s + 1], y1, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[1][s + 1], y1, acc1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (N < 4) {
                if (valid) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        zout[row_of(r, half) * G::SOUT + pos] = relu(acc0[r]);
                        zout[(32 + row_of(r, half)) * G::SOUT + pos] = relu(acc1[r]);
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    psum[0][r] += valid ? relu(acc0[r]) : 0.f;
                    psum[1][r] += valid ? relu(acc1[r]) : 0.f;
                }
            }
        } else {
            // VALU cross-check of the pointwise GEMM: each half sums its 32 input channels, halves are
            // combined with a lane exchange.
            float y[32];
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                Taps tp;
                dw_load(s, tp);
                y[s] = dw_eval(tp);
            }
#pragma unroll 1
            for (int co = 0; co < CH; ++co) {
                float part = 0.f;
#pragma unroll
                for (int s = 0; s < 32; ++s) part = fmaf(pw_w[(2 * s + half) * CH + co], y[s], part);
                const float tot = relu(part + __shfl_xor(part, 32, 64) + pw_b[co]);
                if constexpr (N < 4) {
                    if (valid && half == 0) zout[co * G::SOUT + pos] = tot;
                } else {
                    // pool: sum this tile's 32 positions and accumulate into the wave's own scratch row
                    float sum = (valid && half == 0) ? tot : 0.f;
#pragma unroll
                    for (int o = 16; o >= 1; o >>= 1) sum += __shfl_xor(sum, o, 64);
                    if (lane == 0) poolbuf[wv * CH + co] += sum;
                }
            }
        }
    }

    if constexpr (N == 4 && MFMA) {
        // reduce the pool partials over the 32 positions held by each half-wave
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float s = psum[ct][r];
#pragma unroll
                for (int o = 16; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
                if (col == 0) poolbuf[wv * CH + ct * 32 + row_of(r, half)] = s;
            }
        }
    }
    (void)bias4;
    (void)wa;
}

template <bool MFMA>
__global__ __launch_bounds__(NT) void kws_dscnn_fwd_kernel(DscnnWeights w, const float* __restrict__ feat, int B,
                                                           float* __restrict__ logits, int32_t* __restrict__ label,
                                                           float* __restrict__ act) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;

    // One clip per workgroup (no persistent loop: hoisting the ~400 weight addresses out of a clip
    // loop costs more registers than the relaunch saves).
    const int clip = blockIdx.x;
    if (clip >= B) return;
    {
        // ---- phase 0: MFCC map -> zero-padded [103][14] in LDS; clear the pool scratch ----------
        float* featp = lds + OFF_FEAT;
        for (int i = tid; i < FEAT_H * FEAT_W; i += NT) featp[i] = 0.f;
        for (int i = tid; i < NW * CH; i += NT) lds[OFF_POOLBUF + i] = 0.f;
        __syncthreads();
        const float* f = feat + (size_t)clip * (IN_T * IN_F);
        for (int i = tid; i < IN_T * IN_F; i += NT) featp[(i / IN_F + 2) * FEAT_W + (i % IN_F) + 2] = f[i];
        __syncthreads();

        conv1_phase<MFMA>(w, lds, tid);
        __syncthreads();
        float* a = act ? act + (size_t)clip * KWS_ACT_FLOATS_PER_CLIP : nullptr;
        if (a) {
            for (int i = tid; i < CH * P0; i += NT) a[i] = lds[OFF_Z0 + (i / P0) * (P0 + 2) + i % P0];
            a += CH * P0;
        }

        block_phase<1, MFMA>(w, lds, tid);
        __syncthreads();
        if (a) {
            for (int i = tid; i < CH * Blk<1>::POUT; i += NT)
                a[i] = lds[OFF_Z1 + (i / Blk<1>::POUT) * Blk<1>::SOUT + i % Blk<1>::POUT];
            a += CH * Blk<1>::POUT;
        }
        block_phase<2, MFMA>(w, lds, tid);
        __syncthreads();
        if (a) {
            for (int i = tid; i < CH * Blk<2>::POUT; i += NT)
                a[i] = lds[OFF_Z2 + (i / Blk<2>::POUT) * Blk<2>::SOUT + i % Blk<2>::POUT];
            a += CH * Blk<2>::POUT;
        }
        block_phase<3, MFMA>(w, lds, tid);
        __syncthreads();
        if (a) {
            for (int i = tid; i < CH * Blk<3>::POUT; i += NT)
                a[i] = lds[OFF_Z3 + (i / Blk<3>::POUT) * Blk<3>::SOUT + i % Blk<3>::POUT];
            a += CH * Blk<3>::POUT;
        }
        block_phase<4, MFMA>(w, lds, tid);
        __syncthreads();

        // ---- global average pool over 55 x 11 = 477 interior + 128 ring positions ---------------
        float* pooled = lds + OFF_POOLED;
        if (tid < CH) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < NW; ++k) s += lds[OFF_POOLBUF + k * CH + tid];
            constexpr float RING_N = 55.f * 11.f - 53.f * 9.f;  // 128
            s = fmaf(RING_N, relu(w.pw_b[3 * CH + tid]), s) * (1.0f / (55.f * 11.f));
            pooled[tid] = s;
            if (a) a[tid] = s;
        }
        __syncthreads();

        // ---- Linear(64 -> C) + argmax (first maximum wins) on wavefront 0 ------------------------
        if (wv == 0) {
            const int C = w.num_classes;
            float v = -INFINITY;
            if (lane < C) {
                float acc = w.fc_b[lane];
                const float* wr = w.fc_w + lane * CH;
#pragma unroll 8
                for (int c = 0; c < CH; ++c) acc = fmaf(wr[c], pooled[c], acc);
                logits[(size_t)clip * C + lane] = acc;
                v = acc;
            }
            int idx = lane < C ? lane : 0x7fffffff;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                const float ov = __shfl_xor(v, o, 64);
                const int oi = __shfl_xor(idx, o, 64);
                if (ov > v || (ov == v && oi < idx)) {
                    v = ov;
                    idx = oi;
                }
            }
            if (label && lane == 0) label[clip] = idx;
        }
    }
}

}  // namespace

// The kernel needs the CU's whole 160 KiB of LDS as dynamic shared memory: opt in once per device.
hipError_t dscnn_init_device() {
    const int lds = LDS_FLOATS * (int)sizeof(float);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kws_dscnn_fwd_kernel<true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kws_dscnn_fwd_kernel<false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, lds);
}

hipError_t launch_dscnn(hipStream_t s, const DscnnWeights& w, const float* d_feat, int B, float* d_logits,
                        int32_t* d_label, float* d_act, bool use_mfma) {
    const size_t lds = LDS_FLOATS * sizeof(float);
    const int grid = B;  // one clip per workgroup; one workgroup per CU (160 KiB LDS)
    if (use_mfma)
        hipLaunchKernelGGL(kws_dscnn_fwd_kernel<true>, dim3(grid), dim3(NT), lds, s, w, d_feat, B, d_logits, d_label, d_act);
    else
        hipLaunchKernelGGL(kws_dscnn_fwd_kernel<false>, dim3(grid), dim3(NT), lds, s, w, d_feat, B, d_logits, d_label, d_act);
    return hipGetLastError();
}

}  // namespace kws
