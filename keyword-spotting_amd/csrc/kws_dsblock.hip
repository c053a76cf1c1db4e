// Standalone depthwise-separable block for an arbitrary [B, C_in, H, W] map -- the drop-in for
// DepthwiseSeparableConvBlock.forward (reference kws/libs/models.py:108-119) outside the fused DS-CNN:
//     depthwise Conv2d(C_in, C_in, k, stride, padding, groups=C_in) + bias          (:96-103, :117)
//     pointwise Conv2d(C_in, C_out, 1, stride 1, padding=padding) + bias, ReLU      (:104-106, :118-119)
// The pointwise padding surrounds its output with a ring of `padding` positions equal to relu(bias) -- reference
// behaviour (defect 8), reproduced on purpose.
//
// Not the hot path (the four blocks of the DS-CNN run fused and LDS-resident in kws_dscnn.hip); this is the general-shape
// operator, so it is two plain kernels through a context workspace:
//   kws_dsblock_depthwise_kernel   one thread per output element, taps in registers per channel
//   kws_dsblock_pointwise_kernel   64 positions x 64 output channels per workgroup, the depthwise tile staged through
//                                  LDS in slabs of 16 input channels, 4 x 4 outputs per thread, f32 FMA in channel
//                                  order; ring positions are written by the same kernel (relu(bias), no arithmetic)
#include <algorithm>

#include "kws_internal.h"
#include "kws_split_mfma.h"

namespace kws {
namespace {

__global__ __launch_bounds__(256) void kws_dsblock_depthwise_kernel(const float* __restrict__ x, int C, int H, int W,
                                                                    const float* __restrict__ w, const float* __restrict__ b,
                                                                    int k, int stride, int pad, int Ho, int Wo,
                                                                    float* __restrict__ y, long total) {
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int wo = (int)(idx % Wo);
        const int ho = (int)((idx / Wo) % Ho);
        const long bc = idx / ((long)Wo * Ho);  // b * C + c
        const int c = (int)(bc % C);
        const float* xp = x + bc * (long)H * W;
        const float* wp = w + (long)c * k * k;
        float acc = b[c];
        for (int kh = 0; kh < k; ++kh) {
            const int hi = ho * stride - pad + kh;
            if ((unsigned)hi >= (unsigned)H) continue;
            for (int kw = 0; kw < k; ++kw) {
                const int wi = wo * stride - pad + kw;
                if ((unsigned)wi < (unsigned)W) acc = fmaf(wp[kh * k + kw], xp[(long)hi * W + wi], acc);
            }
        }
        y[idx] = acc;
    }
}

constexpr int TP = 64, TC = 64, TK = 16;  // positions x output channels per workgroup, input channels per slab

__global__ __launch_bounds__(256) void kws_dsblock_pointwise_kernel(const float* __restrict__ dw, int C_in, int P /*Ho*Wo*/,
                                                                    const float* __restrict__ w, const float* __restrict__ bias,
                                                                    int C_out, int Ho, int Wo, int pad, float* __restrict__ out) {
    __shared__ float s_x[TK][TP + 1];
    __shared__ float s_w[TK][TC + 1];
    const int tid = threadIdx.x;
    const int b = blockIdx.z, p0 = blockIdx.x * TP, c0 = blockIdx.y * TC;
    const int tp = tid & 15, tc = tid >> 4;  // this thread: positions tp + 16 i, output channels tc + 16 j
    const int Hp = Ho + 2 * pad, Wp = Wo + 2 * pad;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    const float* dwb = dw + (long)b * C_in * P;
    for (int k0 = 0; k0 < C_in; k0 += TK) {
        for (int e = tid; e < TK * TP; e += 256) {
            const int kk = e / TP, pp = e % TP;
            s_x[kk][pp] = (k0 + kk < C_in && p0 + pp < P) ? dwb[(long)(k0 + kk) * P + p0 + pp] : 0.f;
        }
        for (int e = tid; e < TK * TC; e += 256) {
            const int cc = e / TK, kk = e % TK;
            s_w[kk][cc] = (k0 + kk < C_in && c0 + cc < C_out) ? w[(long)(c0 + cc) * C_in + k0 + kk] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < TK; ++kk) {
            float xv[4], wv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) xv[i] = s_x[kk][tp + 16 * i];
#pragma unroll
            for (int j = 0; j < 4; ++j) wv[j] = s_w[kk][tc + 16 * j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(wv[j], xv[i], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int co = c0 + tc + 16 * j;
        if (co >= C_out) continue;
        const float bv = bias[co];
        float* op = out + ((long)b * C_out + co) * Hp * Wp;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = p0 + tp + 16 * i;
            if (p < P) op[(long)(p / Wo + pad) * Wp + (p % Wo) + pad] = relu(acc[i][j] + bv);
        }
    }
    // the ring: the 1x1 convolution sees only zero padding there, so its value is relu(bias); written once per
    // (batch, channel tile) by the workgroup of the first position tile
    if (pad > 0 && blockIdx.x == 0) {
        for (int cc = 0; cc < TC && c0 + cc < C_out; ++cc) {
            const float rv = relu(bias[c0 + cc]);
            float* op = out + ((long)b * C_out + c0 + cc) * Hp * Wp;
            for (int e = tid; e < Hp * Wp; e += 256) {
                const int h = e / Wp, x = e % Wp;
                if (h < pad || h >= Ho + pad || x < pad || x >= Wo + pad) op[e] = rv;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The rest of DepthwiseSeparableConv.forward for feature maps other than 99 x 10 (reference kws/libs/models.py:160-183 takes
// any [B, C, T, F]: AudioConfig.clip_duration_ms / num_cepstral_coeffs change T and F, audio_processor.py:37-46).  The
// fused LDS-resident kernel is built for the reference geometry; any other map runs composed and HBM-resident:
// conv1 (below) -> 4 x launch_dsblock -> pool + fc (below).
//
// conv1: Conv2d(C_in, 64, 10, stride 2, padding 2) + bias + ReLU (models.py:135,170).  One workgroup per (clip, tile of 64
// output positions); thread (co = tid & 63, pg = tid >> 6) owns output channel co at positions pg, pg + 4, ... of the tile.
// Per input channel the zero-padded plane goes through LDS (every lane of a wavefront reads the same address: a
// broadcast) and the 100 taps come from the [ci][tap][co] weight image (coalesced).
constexpr int C1A_POS = 64, C1A_PER = C1A_POS / 4;
__global__ __launch_bounds__(256) void kws_conv1_any_kernel(const float* __restrict__ x, int C_in, int T, int F, const float* __restrict__ wt,
                                                            const float* __restrict__ bias, int Ho, int Wo, float* __restrict__ out) {
    extern __shared__ float plane[];  // (T + 4) x (F + 4)
    const int tid = threadIdx.x, co = tid & 63, pg = tid >> 6;
    const int Hp = T + 4, Wp = F + 4, P = Ho * Wo, p0 = blockIdx.x * C1A_POS;
    float acc[C1A_PER];
    int base[C1A_PER];  // plane offset of the position's top-left tap (position p: rows 2*(p / Wo) .., columns 2*(p % Wo) ..)
#pragma unroll
    for (int k = 0; k < C1A_PER; ++k) {
        acc[k] = 0.f;
        const int p = min(p0 + pg + 4 * k, P - 1);
        base[k] = 2 * (p / Wo) * Wp + 2 * (p % Wo);
    }
    const float* xc = x + (size_t)blockIdx.y * C_in * T * F;
    for (int ci = 0; ci < C_in; ++ci) {
        __syncthreads();
        for (int i = tid; i < Hp * Wp; i += 256) {
            const int r = i / Wp - 2, c = i % Wp - 2;
            plane[i] = ((unsigned)r < (unsigned)T && (unsigned)c < (unsigned)F) ? xc[(size_t)ci * T * F + r * F + c] : 0.f;
        }
        __syncthreads();
        const float* wc = wt + (size_t)ci * 100 * 64 + co;
        for (int tap = 0; tap < 100; ++tap) {
            const float wv = wc[(size_t)tap * 64];
            const int off = (tap / 10) * Wp + tap % 10;
#pragma unroll
            for (int k = 0; k < C1A_PER; ++k) acc[k] = fmaf(wv, plane[base[k] + off], acc[k]);
        }
    }
    float* o = out + ((size_t)blockIdx.y * 64 + co) * P;
    const float bv = bias[co];
#pragma unroll
    for (int k = 0; k < C1A_PER; ++k) {
        const int p = p0 + pg + 4 * k;
        if (p < P) o[p] = relu(acc[k] + bv);
    }
}

// F.adaptive_avg_pool2d(x, (1, 1)) + view + fc (models.py:179-181) + argmax, first maximum wins (training.py:371).
// x: [B][64][HW] (the last block's output, ring included).  One workgroup per clip: four partial sums per channel, fc and
// argmax on wavefront 0.
__global__ __launch_bounds__(256) void kws_pool_fc_kernel(const float* __restrict__ x, int HW, const float* __restrict__ fc_w,
                                                          const float* __restrict__ fc_b, int C, float* __restrict__ logits,
                                                          int32_t* __restrict__ label) {
    __shared__ float part[4][64];
    __shared__ float pooled[64];
    __shared__ float lg[MAX_CLASSES];
    const int tid = threadIdx.x, c = tid & 63, q = tid >> 6;
    const float* xp = x + ((size_t)blockIdx.x * 64 + c) * HW;
    float s = 0.f;
    for (int i = q; i < HW; i += 4) s += xp[i];
    part[q][c] = s;
    __syncthreads();
    if (tid < 64) pooled[tid] = ((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid])) * (1.0f / (float)HW);
    __syncthreads();
    if (tid < C) {
        float acc = fc_b[tid];
        for (int k = 0; k < 64; ++k) acc = fmaf(fc_w[tid * 64 + k], pooled[k], acc);
        logits[(size_t)blockIdx.x * C + tid] = acc;
        lg[tid] = acc;
    }
    __syncthreads();
    if (tid == 0 && label) {
        int best = 0;
        for (int k = 1; k < C; ++k)
            if (lg[k] > lg[best]) best = k;  // strict: the first maximum wins
        label[blockIdx.x] = best;
    }
}

}  // namespace

hipError_t launch_conv1_any(hipStream_t s, const float* d_x, int B, int C_in, int T, int F, const float* d_wt, const float* d_bias,
                            float* d_out) {
    const int Ho = (T + 4 - 10) / 2 + 1, Wo = (F + 4 - 10) / 2 + 1;
    const size_t lds = sizeof(float) * (size_t)(T + 4) * (F + 4);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kws_conv1_any_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kws_conv1_any_kernel, dim3((Ho * Wo + C1A_POS - 1) / C1A_POS, B), dim3(256), lds, s, d_x, C_in, T, F, d_wt, d_bias,
                       Ho, Wo, d_out);
    return hipGetLastError();
}

hipError_t launch_pool_fc(hipStream_t s, const float* d_x, int B, int HW, const float* d_fc_w, const float* d_fc_b, int C,
                          float* d_logits, int32_t* d_label) {
    hipLaunchKernelGGL(kws_pool_fc_kernel, dim3(B), dim3(256), 0, s, d_x, HW, d_fc_w, d_fc_b, C, d_logits, d_label);
    return hipGetLastError();
}

hipError_t launch_dsblock(hipStream_t s, const float* d_x, int B, int C_in, int H, int W, const float* d_dw_w, const float* d_dw_b,
                          const float* d_pw_w, const float* d_pw_b, int C_out, int k, int stride, int pad, float* d_ws,
                          float* d_out) {
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    const long total = (long)B * C_in * Ho * Wo;
    const int blocks = (int)std::min<long>((total + 255) / 256, 1 << 20);
    hipLaunchKernelGGL(kws_dsblock_depthwise_kernel, dim3(blocks), dim3(256), 0, s, d_x, C_in, H, W, d_dw_w, d_dw_b, k, stride, pad,
                       Ho, Wo, d_ws, total);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int P = Ho * Wo;
    hipLaunchKernelGGL(kws_dsblock_pointwise_kernel, dim3((P + TP - 1) / TP, (C_out + TC - 1) / TC, B), dim3(256), 0, s, d_ws, C_in,
                       P, d_pw_w, d_pw_b, C_out, Ho, Wo, pad, d_out);
    return hipGetLastError();
}

}  // namespace kws
