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

}  // namespace

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
