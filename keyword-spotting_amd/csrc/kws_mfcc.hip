// Batched MFCC front end for gfx950 (MI355X), replacing per-clip psf.mfcc calls
// (reference call site kws/libs/audio_processor.py:270-278; stages a1-a8 of SURVEY.md section 8).
//
// Work decomposition
//   grid  = (ceil(num_frames / (2*MFCC_PAIRS)), B); one 64-lane wavefront per workgroup.
//   A workgroup stages the PCM span of its 10 frames (1840 samples for 400/160) from HBM with 16-byte
//   loads, converts to float32, applies pre-emphasis once, and keeps the span in LDS.
//   Two real frames are packed into one 512-point complex FFT (z = a + i*b): lane l holds
//   z[64*n1 + l], n1 = 0..7, and the transform is three radix-8 passes in registers with two LDS
//   exchanges (8 x 8 x 8).  The spectra of the two frames are separated with the conjugate-symmetry
//   identity, |X|^2/512 goes to LDS once, and the 26 triangular mel filters are evaluated sparsely:
//   the bins [edge_s, edge_s+1) between two mel edges are cut into chunks of <= 8 bins, one chunk per
//   lane, each lane accumulating the rising weights (filter s) and falling weights (filter s-1) of its
//   bins; a filter is the sum of <= a few chunk partials.  log, then a [numcep x nfilt] DCT-II(ortho)
//   x lifter table from LDS; coefficient 0 is log(frame energy) (psf appendEnergy).
//
// Numerics: PCM/32768 and pre-emphasis are bit-exact float32 as in the reference pipeline (separate
// multiply and subtract roundings); everything after is float32 here vs float64 in psf
// (tolerance 1e-4 on MFCC, see tests/test_mfcc_gpu.py).
#include "kws_internal.h"

namespace kws {
namespace {

struct cf {
    float x, y;
};

__device__ __forceinline__ cf cadd(cf a, cf b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cf csub(cf a, cf b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cf cmul(cf a, cf b) {
    return {fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x)};
}
// multiply by -i
__device__ __forceinline__ cf mul_mi(cf a) { return {a.y, -a.x}; }

// In-place 8-point forward DFT, natural order in and out: v[k] = sum_n v[n] * exp(-2*pi*i*n*k/8).
__device__ __forceinline__ void dft8(cf (&v)[8]) {
    constexpr float R = 0.70710678118654752440f;
    cf b0 = cadd(v[0], v[4]), b4 = csub(v[0], v[4]);
    cf b1 = cadd(v[1], v[5]), b5 = csub(v[1], v[5]);
    cf b2 = cadd(v[2], v[6]), b6 = csub(v[2], v[6]);
    cf b3 = cadd(v[3], v[7]), b7 = csub(v[3], v[7]);
    // odd branch pre-twiddles W8^n
    b5 = {(b5.x + b5.y) * R, (b5.y - b5.x) * R};   // * (1 - i)/sqrt2
    b6 = mul_mi(b6);                               // * -i
    b7 = {(b7.y - b7.x) * R, -(b7.x + b7.y) * R};  // * (-1 - i)/sqrt2
    // even outputs: 4-point DFT of b0..b3
    cf d0 = cadd(b0, b2), d1 = csub(b0, b2), d2 = cadd(b1, b3), d3 = mul_mi(csub(b1, b3));
    v[0] = cadd(d0, d2);
    v[4] = csub(d0, d2);
    v[2] = cadd(d1, d3);
    v[6] = csub(d1, d3);
    // odd outputs: 4-point DFT of b4..b7
    cf e0 = cadd(b4, b6), e1 = csub(b4, b6), e2 = cadd(b5, b7), e3 = mul_mi(csub(b5, b7));
    v[1] = cadd(e0, e2);
    v[5] = csub(e0, e2);
    v[3] = cadd(e1, e3);
    v[7] = csub(e1, e3);
}

// Complex row strides of the two register<->LDS exchanges.  72 (64 + 8) makes the strided column gather of
// the first exchange conflict-free for ds_read_b64; 66 puts the sixteen 64-byte row segments that one
// ds_read_b128 lane group fetches in the second exchange on sixteen different 16-byte slots.
constexpr int XROW1 = 72, XROW2 = 66;

// Every workgroup is ONE wavefront: LDS instructions of a wavefront execute in order, so data written by one
// lane is visible to a later read of another lane without s_barrier or s_waitcnt; only the compiler has to
// keep the order.
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
// xbuf: 8*XROW1 complex of LDS private to the wavefront.
__device__ __forceinline__ void fft512(cf (&v)[8], cf* xbuf, const cf (&t1)[8], const cf (&t2)[8], int lane) {
    const int k1 = lane >> 3, q = lane & 7;
    dft8(v);  // over n1 -> k1 (register index)
#pragma unroll
    for (int i = 1; i < 8; ++i) v[i] = cmul(v[i], t1[i]);  // W512^(lane*k1)
#pragma unroll
    for (int i = 0; i < 8; ++i) xbuf[i * XROW1 + lane] = v[i];
    wave_lds_order();
    // lane (k1, b=q): gather y[k1][8a + b], a = 0..7
#pragma unroll
    for (int a = 0; a < 8; ++a) v[a] = xbuf[k1 * XROW1 + 8 * a + q];
    wave_lds_order();
    dft8(v);  // over a -> c
#pragma unroll
    for (int i = 1; i < 8; ++i) v[i] = cmul(v[i], t2[i]);  // W64^(b*c)
#pragma unroll
    for (int c = 0; c < 8; ++c) xbuf[k1 * XROW2 + 8 * c + q] = v[c];
    wave_lds_order();
    // lane (k1, c=q): gather u[k1][c][b], b = 0..7 (contiguous)
#pragma unroll
    for (int b = 0; b < 8; ++b) v[b] = xbuf[k1 * XROW2 + 8 * q + b];
    wave_lds_order();
    dft8(v);  // over b -> d
}

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

// Load the per-lane FFT twiddles.
__device__ __forceinline__ void load_twiddles(const float2* __restrict__ tw, int lane, cf (&t1)[8], cf (&t2)[8]) {
    const int q = lane & 7;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float2 a = tw[(lane * i) & 511];
        float2 b = tw[(8 * q * i) & 511];
        t1[i] = {a.x, a.y};
        t2[i] = {b.x, b.y};
    }
}

// Write Z to LDS in natural order and return, for the bins this lane owns (k = lane + 64*j, j<4, and
// k = 256 on lane 0), the power spectra 1/512*|A|^2, 1/512*|B|^2 of the two packed real frames.
// nza / nzb: whether frame a / b has any non-zero sample.  An all-zero frame must give exactly 0 (the
// reference then floors to eps); computed through the packed transform it would instead pick up the
// partner frame's float32 rounding noise (-140 dB), so it is forced.
__device__ __forceinline__ void split_power(const cf (&v)[8], cf* zbuf, float2* pbuf, int lane, int power, bool nza,
                                            bool nzb, float& ea, float& eb) {
    const int k1 = lane >> 3, q = lane & 7;
#pragma unroll
    for (int d = 0; d < 8; ++d) zbuf[zswz(k1 + 8 * q + 64 * d)] = v[d];
    wave_lds_order();
    ea = 0.f;
    eb = 0.f;
    const float scale = power ? (1.0f / (4.0f * NFFT)) : 0.25f;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int k = (j < 4) ? lane + 64 * j : 256;
        if (j == 4 && lane != 0) break;
        cf z = zbuf[zswz(k)], w = zbuf[zswz((NFFT - k) & (NFFT - 1))];
        float ar = z.x + w.x, ai = z.y - w.y;  // 2*A
        float br = z.y + w.y, bi = z.x - w.x;  // 2*B (up to a unit factor)
        float pa = fmaf(ar, ar, ai * ai) * scale;
        float pb = fmaf(br, br, bi * bi) * scale;
        if (!power) {
            pa = sqrtf(pa);
            pb = sqrtf(pb);
        }
        pa = nza ? pa : 0.f;
        pb = nzb ? pb : 0.f;
        pbuf[k] = make_float2(pa, pb);
        ea += pa;
        eb += pb;
    }
    wave_lds_order();
}

constexpr float PSF_EPS = 2.220446049250313e-16f;  // numpy.finfo(float).eps, exactly 2^-52

// ------------------------------------------------------------------------------------------------
// Sample -> float32 in [-1, 1): int16 PCM is scaled like librosa/soundfile do (x / 32768, exact);
// float32 input is taken as is (a signal the caller already decoded / augmented).
__device__ __forceinline__ float to_unit(int16_t s) { return (float)s * (1.0f / 32768.0f); }
__device__ __forceinline__ float to_unit(float s) { return s; }

template <typename T>
__device__ __forceinline__ void mfcc_body(const FrontendParams& p, const FrontendTables& t, const T* __restrict__ wav,
                                          float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // carve (all offsets multiples of 16 bytes)
    cf* xbuf = reinterpret_cast<cf*>(smem);                                  // 8*72 complex = 4608 B
    float2* pbuf = reinterpret_cast<float2*>(smem + 4608);                   // 264 float2 = 2112 B
    float* lbuf = reinterpret_cast<float*>(smem + 4608 + 2112);              // 2*64 floats = 512 B
    float* dctb = reinterpret_cast<float*>(smem + 4608 + 2112 + 512);        // numcep*nfilt (<= 2048 floats)
    const int dct_n = p.numcep * p.nfilt;
    float* ybuf = dctb + ((dct_n + 3) & ~3);                                 // chunk_samples floats

    const int lane = threadIdx.x;
    const int clip = blockIdx.y;
    const int f0 = blockIdx.x * (2 * MFCC_PAIRS);
    const T* __restrict__ x = wav + (size_t)clip * p.n_samples;

    // ---- stage PCM span -> float32, pre-emphasised, in LDS ---------------------------------------
    const int s0 = f0 * p.frame_step;
    const float c = p.preemph;
    if (sizeof(T) == 2 && p.vec_ok && s0 + p.chunk_samples <= p.n_samples) {
        // chunk_samples is a multiple of 8 here and every 8-sample group is 16-byte aligned
        for (int g = lane; g * 8 < p.chunk_samples; g += 64) {
            const int n = s0 + g * 8;
            const uint4 raw = *reinterpret_cast<const uint4*>(x + n);
            float prev = (n > 0) ? to_unit(x[n - 1]) : 0.f;
            const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
            float y[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int16_t s = (int16_t)((w[i >> 1] >> (16 * (i & 1))) & 0xffffu);
                const float cur = to_unit(s);
                y[i] = (n + i > 0) ? __fsub_rn(cur, __fmul_rn(c, prev)) : cur;
                prev = cur;
            }
            float4* dst = reinterpret_cast<float4*>(ybuf + g * 8);
            dst[0] = make_float4(y[0], y[1], y[2], y[3]);
            dst[1] = make_float4(y[4], y[5], y[6], y[7]);
        }
    } else {
        for (int i = lane; i < p.chunk_samples; i += 64) {
            const int n = s0 + i;
            float y = 0.f;
            if (n < p.n_samples) {
                const float cur = to_unit(x[n]);
                y = (n > 0) ? __fsub_rn(cur, __fmul_rn(c, to_unit(x[n - 1]))) : cur;
            }
            ybuf[i] = y;
        }
    }
    for (int i = lane; i < dct_n; i += 64) dctb[i] = t.dct[i];

    // ---- per-lane constants -----------------------------------------------------------------------
    cf t1[8], t2[8];
    load_twiddles(t.twiddle, lane, t1, t2);
    const int mk0 = t.mel_k0[lane];
    float rw[MEL_CHUNK], fw[MEL_CHUNK];
#pragma unroll
    for (int i = 0; i < MEL_CHUNK; ++i) {
        rw[i] = t.mel_rw[i * 64 + lane];
        fw[i] = t.mel_fw[i * 64 + lane];
    }
    const uint32_t gth = t.mel_gather[lane];
    wave_lds_order();

    float4* cbuf = reinterpret_cast<float4*>(xbuf);  // 64 float4 chunk partials (aliases the exchange buffer)

    for (int pr = 0; pr < MFCC_PAIRS; ++pr) {
        const int fa = f0 + 2 * pr;
        if (fa >= p.num_frames) break;  // uniform
        const bool has_b = (fa + 1) < p.num_frames;
        const float* ya = ybuf + (2 * pr) * p.frame_step;
        const float* yb = ya + p.frame_step;

        cf v[8];
        bool nza = false, nzb = false;
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) {
            const int i = 64 * n1 + lane;
            const bool in = i < p.frame_len;
            v[n1].x = in ? ya[i] : 0.f;
            v[n1].y = (in && has_b) ? yb[i] : 0.f;
            nza |= v[n1].x != 0.f;
            nzb |= v[n1].y != 0.f;
        }
        nza = __any(nza);
        nzb = __any(nzb);
        fft512(v, xbuf, t1, t2, lane);

        float ea, eb;
        split_power(v, xbuf, pbuf, lane, 1, nza, nzb, ea, eb);
        ea = wave_sum(ea);
        eb = wave_sum(eb);

        // sparse mel: this lane's chunk of <= 8 bins
        float ra = 0.f, fa_ = 0.f, rb = 0.f, fb_ = 0.f;
#pragma unroll
        for (int i = 0; i < MEL_CHUNK; ++i) {
            const int k = min(mk0 + i, NBINS - 1);
            const float2 pw = pbuf[k];
            ra = fmaf(rw[i], pw.x, ra);
            fa_ = fmaf(fw[i], pw.x, fa_);
            rb = fmaf(rw[i], pw.y, rb);
            fb_ = fmaf(fw[i], pw.y, fb_);
        }
        cbuf[lane] = make_float4(ra, fa_, rb, fb_);
        wave_lds_order();
        if (lane < p.nfilt) {
            const int r0 = gth & 255, nr = (gth >> 8) & 255, q0 = (gth >> 16) & 255, nq = gth >> 24;
            float sa = 0.f, sb = 0.f;
            for (int i = 0; i < nr; ++i) {
                const float4 qv = cbuf[r0 + i];
                sa += qv.x;
                sb += qv.z;
            }
            for (int i = 0; i < nq; ++i) {
                const float4 qv = cbuf[q0 + i];
                sa += qv.y;
                sb += qv.w;
            }
            lbuf[lane] = logf(sa == 0.f ? PSF_EPS : sa);
            lbuf[64 + lane] = logf(sb == 0.f ? PSF_EPS : sb);
        }
        wave_lds_order();

        // DCT-II(ortho) x lifter: lane -> (frame f = lane>>5, coefficient i = lane&31)
        {
            const int f = lane >> 5, i = lane & 31;
            if (i < p.numcep && (f == 0 || has_b)) {
                const float* L = lbuf + 64 * f;
                const float* D = dctb + i * p.nfilt;
                const float m = L[0];
                float acc = 0.f, dsum = 0.f;
                for (int j = 0; j < p.nfilt; ++j) {
                    acc = fmaf(D[j], L[j] - m, acc);
                    dsum += D[j];
                }
                if (i == 0) {
                    if (p.append_energy) {
                        const float e = f ? eb : ea;
                        acc = logf(e == 0.f ? PSF_EPS : e);
                    } else {
                        acc = fmaf(m, dsum, acc);
                    }
                }
                out[((size_t)clip * p.num_frames + (fa + f)) * p.numcep + i] = acc;
            }
        }
        wave_lds_order();
    }
}

__global__ __launch_bounds__(64) void kws_mfcc_i16_kernel(FrontendParams p, FrontendTables t,
                                                          const int16_t* __restrict__ wav, float* __restrict__ out) {
    mfcc_body<int16_t>(p, t, wav, out);
}
__global__ __launch_bounds__(64) void kws_mfcc_f32_kernel(FrontendParams p, FrontendTables t,
                                                          const float* __restrict__ wav, float* __restrict__ out) {
    mfcc_body<float>(p, t, wav, out);
}

// ------------------------------------------------------------------------------------------------
// sigproc operators (kws/libs/speech_features/sigproc.py), float32 device versions.
__global__ void kws_preemphasis_f32_kernel(const float* __restrict__ in, int n, float coeff, float* __restrict__ out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        out[i] = (i > 0) ? __fsub_rn(in[i], __fmul_rn(coeff, in[i - 1])) : in[i];
}

__global__ void kws_framesig_f32_kernel(const float* __restrict__ in, int n, int frame_len, int frame_step,
                                        int num_frames, const float* __restrict__ window,
                                        float* __restrict__ frames) {
    const long total = (long)num_frames * frame_len;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int f = (int)(idx / frame_len), i = (int)(idx % frame_len);
        const long s = (long)f * frame_step + i;
        float v = (s < n) ? in[s] : 0.f;
        if (window) v *= window[i];
        frames[idx] = v;
    }
}

// magspec / powspec with NFFT = 512: one wavefront per pair of frames.
__global__ __launch_bounds__(64) void kws_spec512_f32_kernel(FrontendTables t, const float* __restrict__ frames,
                                                             int num_frames, int frame_len, int power,
                                                             float* __restrict__ spec) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[4608 + 2112];
    cf* xbuf = reinterpret_cast<cf*>(smem);
    float2* pbuf = reinterpret_cast<float2*>(smem + 4608);
    const int lane = threadIdx.x;
    const int fa = 2 * blockIdx.x;
    const bool has_b = fa + 1 < num_frames;
    cf t1[8], t2[8];
    load_twiddles(t.twiddle, lane, t1, t2);
    const float* ya = frames + (size_t)fa * frame_len;
    const float* yb = ya + frame_len;
    cf v[8];
    bool nza = false, nzb = false;
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) {
        const int i = 64 * n1 + lane;
        const bool in = i < frame_len;  // frames longer than NFFT are truncated (sigproc.py:65-66)
        v[n1].x = in ? ya[i] : 0.f;
        v[n1].y = (in && has_b) ? yb[i] : 0.f;
        nza |= v[n1].x != 0.f;
        nzb |= v[n1].y != 0.f;
    }
    nza = __any(nza);
    nzb = __any(nzb);
    fft512(v, xbuf, t1, t2, lane);
    float ea, eb;
    split_power(v, xbuf, pbuf, lane, power, nza, nzb, ea, eb);
    for (int k = lane; k < NBINS; k += 64) {
        const float2 pw = pbuf[k];
        spec[(size_t)fa * NBINS + k] = pw.x;
        if (has_b) spec[(size_t)(fa + 1) * NBINS + k] = pw.y;
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
template <typename T, typename K>
static hipError_t launch_mfcc_t(K kernel, hipStream_t s, const FrontendParams& p, const FrontendTables& t, const T* d_wav,
                                int B, float* d_out) {
    const int per_wg = 2 * MFCC_PAIRS;
    dim3 grid((p.num_frames + per_wg - 1) / per_wg, B);
    const int dct_n = p.numcep * p.nfilt;
    const size_t lds = 4608 + 2112 + 512 + sizeof(float) * (size_t)(((dct_n + 3) & ~3) + ((p.chunk_samples + 3) & ~3));
    // grid.y is limited to 65535: split very large batches
    for (int b0 = 0; b0 < B; b0 += 65535) {
        const int nb = (B - b0 < 65535) ? (B - b0) : 65535;
        grid.y = nb;
        hipLaunchKernelGGL(kernel, grid, dim3(64), lds, s, p, t, d_wav + (size_t)b0 * p.n_samples,
                           d_out + (size_t)b0 * p.num_frames * p.numcep);
    }
    return hipGetLastError();
}

hipError_t launch_mfcc(hipStream_t s, const FrontendParams& p, const FrontendTables& t, const int16_t* d_wav, int B,
                       float* d_out) {
    return launch_mfcc_t(kws_mfcc_i16_kernel, s, p, t, d_wav, B, d_out);
}
hipError_t launch_mfcc_f32(hipStream_t s, const FrontendParams& p, const FrontendTables& t, const float* d_wav, int B,
                           float* d_out) {
    return launch_mfcc_t(kws_mfcc_f32_kernel, s, p, t, d_wav, B, d_out);
}

hipError_t launch_preemphasis(hipStream_t s, const float* d_in, int n, float coeff, float* d_out) {
    const int blocks = (n + 255) / 256;
    hipLaunchKernelGGL(kws_preemphasis_f32_kernel, dim3(blocks < 2048 ? blocks : 2048), dim3(256), 0, s, d_in, n, coeff, d_out);
    return hipGetLastError();
}

hipError_t launch_framesig(hipStream_t s, const float* d_in, int n, int frame_len, int frame_step, int num_frames,
                           const float* d_window, float* d_frames) {
    const long total = (long)num_frames * frame_len;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(kws_framesig_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, s, d_in, n, frame_len, frame_step,
                       num_frames, d_window, d_frames);
    return hipGetLastError();
}

hipError_t launch_spec512(hipStream_t s, const FrontendTables& t, const float* d_frames, int num_frames, int frame_len,
                          int power, float* d_spec) {
    hipLaunchKernelGGL(kws_spec512_f32_kernel, dim3((num_frames + 1) / 2), dim3(64), 0, s, t, d_frames, num_frames,
                       frame_len, power, d_spec);
    return hipGetLastError();
}

}  // namespace kws
