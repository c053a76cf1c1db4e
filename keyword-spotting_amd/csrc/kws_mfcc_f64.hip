// Reference-grade MFCC front end in float64 for gfx950 -- the general-geometry and accurate-arithmetic companion of the
// float32 kernel in kws_mfcc.hip.  Same stages as psf.mfcc as the reference calls it (kws/libs/audio_processor.py:270-278;
// SURVEY.md section 8 a1-a8) and the same dtypes NumPy uses there: PCM scaling and pre-emphasis in float32 (bit-exact),
// everything after framing in float64, the result cast to float32 (kws/libs/data_loader.py:103).
//
// When it runs
//   * any geometry the fast kernel is not built for: nfft != 512 (the reference derives nfft = max(fft_size,
//     int(winlen * samplerate)), audio_processor.py:268, so a 40 ms window at 16 kHz means nfft = 640), frames longer
//     than 512 samples, filterbanks its sparse lane layout cannot hold;
//   * on request (kws_set_frontend_math(ctx, KWS_FE_F64)) for the default geometry: a float32 transform leaves rounding
//     noise ~138 dB below a frame's strongest component, which a clean tone over a quiet floor turns into cepstral errors
//     of up to 6e-4; float64 matches the reference to the float32 rounding of the output ("Precision").
//
// Work decomposition: one wavefront per pair of frames (the two real frames are the real and imaginary part of one complex
// transform; in float64 the cross-talk between them is 1e-16 of the larger one, so no level equalisation is needed).
//   nfft a power of two (64..4096): radix-2 decimation-in-frequency FFT in LDS, in place, bit-reversed read-out
//   any other nfft (<= 2048):       direct DFT, each lane a bin, twiddles W^(n k mod nfft) from an LDS copy of the table
// then |X|^2/nfft, frame energy, the triangular mel filters evaluated from psf's bin edges (weights formed in float64
// exactly as psf's get_filterbanks does), eps floors, log, DCT-II(ortho) x lifter from a float64 table, c0 = log(energy).
#include "kws_internal.h"

namespace kws {
namespace {

typedef double d2 __attribute__((ext_vector_type(2)));

constexpr double PSF_EPS64 = 2.220446049250313e-16;

__device__ __forceinline__ float to_unit64(int16_t s) { return (float)s * (1.0f / 32768.0f); }
__device__ __forceinline__ float to_unit64(float s) { return s; }

// sample m of the clip after pre-emphasis, float32 arithmetic as NumPy does it (two roundings), 0 outside the clip
template <typename T>
__device__ __forceinline__ float preemph_sample(const T* __restrict__ x, long m, int n_samples, float c) {
    if (m < 0 || m >= n_samples) return 0.f;
    const float cur = to_unit64(x[m]);
    return m > 0 ? __fsub_rn(cur, __fmul_rn(c, to_unit64(x[m - 1]))) : cur;
}

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Spectra of the frame pair whose samples sit in X[n] = (a[n], b[n]), n < nfft (zero beyond the frame): writes
// P[k] = (|A[k]|^2, |B[k]|^2) * scale for k = 0..nfft/2 (scale = 1/nfft for the power spectrum, 1 for |X|^2).
// One wavefront; LDS instructions of a wavefront execute in order, so __syncthreads() (a single wave) only keeps the
// compiler from reordering across the stages.
template <bool POW2>
__device__ __forceinline__ void spectrum_pair(d2* X, d2* P, const d2* __restrict__ tw_global, d2* tw_lds, int nfft, int log2n,
                                              int n_used, double scale, int lane) {
    const int nb = nfft / 2 + 1;
    if constexpr (POW2) {
        // radix-2 decimation in frequency, in place: stage with half-size s pairs (j, j + s) and multiplies the
        // difference by W_nfft^((i mod s) * nfft / (2 s))
        for (int s = nfft >> 1, step = 1; s >= 1; s >>= 1, step <<= 1) {
            for (int i = lane; i < (nfft >> 1); i += 64) {
                const int r = i & (s - 1);
                const int j = ((i - r) << 1) + r;
                const d2 a = X[j], b = X[j + s];
                const d2 w = tw_global[r * step];
                const d2 d = a - b;
                X[j] = a + b;
                X[j + s] = d2{__builtin_fma(d.x, w.x, -(d.y * w.y)), __builtin_fma(d.x, w.y, d.y * w.x)};
            }
            __syncthreads();
        }
        for (int k = lane; k < nb; k += 64) {
            const int km = (nfft - k) & (nfft - 1);
            const d2 z = X[__brev((unsigned)k) >> (32 - log2n)], w = X[__brev((unsigned)km) >> (32 - log2n)];
            const double ar = 0.5 * (z.x + w.x), ai = 0.5 * (z.y - w.y);  // A = (Z[k] + conj Z[N-k]) / 2
            const double br = 0.5 * (z.y + w.y), bi = 0.5 * (w.x - z.x);  // B = (Z[k] - conj Z[N-k]) / (2i)
            P[k] = d2{__builtin_fma(ar, ar, ai * ai) * scale, __builtin_fma(br, br, bi * bi) * scale};
        }
    } else {
        for (int i = lane; i < nfft; i += 64) tw_lds[i] = tw_global[i];
        __syncthreads();
        for (int k = lane; k < nb; k += 64) {
            double ar = 0.0, ai = 0.0, br = 0.0, bi = 0.0;
            int idx = 0;  // n * k mod nfft
            for (int n = 0; n < n_used; ++n) {
                const d2 x = X[n], w = tw_lds[idx];
                ar = __builtin_fma(x.x, w.x, ar);
                ai = __builtin_fma(x.x, w.y, ai);
                br = __builtin_fma(x.y, w.x, br);
                bi = __builtin_fma(x.y, w.y, bi);
                idx += k;
                if (idx >= nfft) idx -= nfft;
            }
            P[k] = d2{__builtin_fma(ar, ar, ai * ai) * scale, __builtin_fma(br, br, bi * bi) * scale};
        }
    }
    __syncthreads();
}

template <typename T, bool POW2>
__global__ __launch_bounds__(64) void kws_mfcc_f64_kernel(FrontendParams p, FrontendTables t, const T* __restrict__ wav,
                                                         float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem64[];
    const int nfft = p.nfft, nb = nfft / 2 + 1;
    d2* X = reinterpret_cast<d2*>(smem64);
    d2* P = X + nfft;
    double* L = reinterpret_cast<double*>(P + nb);       // [2][64] log-mel
    d2* tw_lds = reinterpret_cast<d2*>(L + 128);          // [nfft], direct path only
    const int lane = threadIdx.x;
    const int clip = blockIdx.y;
    const int fa = 2 * blockIdx.x;
    const bool has_b = fa + 1 < p.num_frames;
    const T* __restrict__ x = wav + (size_t)clip * p.n_samples;
    const int n_used = p.frame_len < nfft ? p.frame_len : nfft;  // frames longer than nfft are truncated (np.fft.rfft(frames, NFFT))

    const long sa = (long)fa * p.frame_step, sb = sa + p.frame_step;
    for (int n = lane; n < nfft; n += 64) {
        double a = 0.0, b = 0.0;
        if (n < n_used) {
            a = (double)preemph_sample(x, sa + n, p.n_samples, p.preemph);
            if (has_b) b = (double)preemph_sample(x, sb + n, p.n_samples, p.preemph);
        }
        X[n] = d2{a, b};
    }
    __syncthreads();
    spectrum_pair<POW2>(X, P, reinterpret_cast<const d2*>(t.tw64), tw_lds, nfft, p.log2_nfft, n_used, 1.0 / (double)nfft, lane);

    // frame energy = sum over all bins (psf fbank), zero -> eps
    double ea = 0.0, eb = 0.0;
    for (int k = lane; k < nb; k += 64) {
        const d2 pw = P[k];
        ea += pw.x;
        eb += pw.y;
    }
    ea = wave_sum_f64(ea);
    eb = wave_sum_f64(eb);
    if (ea == 0.0) ea = PSF_EPS64;
    if (eb == 0.0) eb = PSF_EPS64;

    // mel filter `lane`: rising edge over [e0, e1), falling edge over [e1, e2), weights as psf get_filterbanks forms them
    double la = 0.0, lb = 0.0;
    if (lane < p.nfilt) {
        const int e0 = t.mel_edges[lane], e1 = t.mel_edges[lane + 1], e2 = t.mel_edges[lane + 2];
        double fa_ = 0.0, fb_ = 0.0;
        for (int i = e0; i < e1; ++i) {
            const double w = (double)(i - e0) / (double)(e1 - e0);
            const d2 pw = P[i];
            fa_ = __builtin_fma(w, pw.x, fa_);
            fb_ = __builtin_fma(w, pw.y, fb_);
        }
        for (int i = e1; i < e2; ++i) {
            const double w = (double)(e2 - i) / (double)(e2 - e1);
            const d2 pw = P[i];
            fa_ = __builtin_fma(w, pw.x, fa_);
            fb_ = __builtin_fma(w, pw.y, fb_);
        }
        la = log(fa_ == 0.0 ? PSF_EPS64 : fa_);
        lb = log(fb_ == 0.0 ? PSF_EPS64 : fb_);
    }
    L[lane] = la;
    L[64 + lane] = lb;
    __syncthreads();

    // DCT-II(ortho) x lifter: lane -> (frame f = lane >> 5, coefficient i = lane & 31)
    const int f = lane >> 5, i = lane & 31;
    if (i < p.numcep && (f == 0 || has_b)) {
        const double* __restrict__ D = t.dct64 + (size_t)i * p.nfilt;
        const double* Lf = L + 64 * f;
        double acc = 0.0;
        for (int j = 0; j < p.nfilt; ++j) acc = __builtin_fma(D[j], Lf[j], acc);
        if (i == 0 && p.append_energy) acc = log(f ? eb : ea);
        out[((size_t)clip * p.num_frames + fa + f) * p.numcep + i] = (float)acc;
    }
}

// magspec / powspec for any NFFT (kws/libs/speech_features/sigproc.py:55-90): frames float32 [num_frames][frame_len]
// (zero-padded to nfft, or truncated), one wavefront per pair of frames, float64 inside, float32 out.
template <bool POW2>
__global__ __launch_bounds__(64) void kws_spec_f64_kernel(const float* __restrict__ frames, int num_frames, int frame_len, int nfft,
                                                         int log2n, int power, const d2* __restrict__ tw, float* __restrict__ spec) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem64[];
    const int nb = nfft / 2 + 1;
    d2* X = reinterpret_cast<d2*>(smem64);
    d2* P = X + nfft;
    d2* tw_lds = P + nb;
    const int lane = threadIdx.x;
    const int fa = 2 * blockIdx.x;
    const bool has_b = fa + 1 < num_frames;
    const int n_used = frame_len < nfft ? frame_len : nfft;
    for (int n = lane; n < nfft; n += 64) {
        double a = 0.0, b = 0.0;
        if (n < n_used) {
            a = (double)frames[(size_t)fa * frame_len + n];
            if (has_b) b = (double)frames[(size_t)(fa + 1) * frame_len + n];
        }
        X[n] = d2{a, b};
    }
    __syncthreads();
    spectrum_pair<POW2>(X, P, tw, tw_lds, nfft, log2n, n_used, power ? 1.0 / (double)nfft : 1.0, lane);
    for (int k = lane; k < nb; k += 64) {
        const d2 pw = P[k];
        spec[(size_t)fa * nb + k] = (float)(power ? pw.x : sqrt(pw.x));
        if (has_b) spec[(size_t)(fa + 1) * nb + k] = (float)(power ? pw.y : sqrt(pw.y));
    }
}

size_t f64_lds_bytes(int nfft, bool pow2) {
    const size_t nb = nfft / 2 + 1;
    return sizeof(d2) * ((size_t)nfft + nb + (pow2 ? 0 : (size_t)nfft)) + sizeof(double) * 128;
}

template <typename K>
hipError_t raise_lds_limit(K kernel, size_t lds) {
    if (lds <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

template <typename T>
hipError_t launch_mfcc_f64_t(hipStream_t s, const FrontendParams& p, const FrontendTables& t, const T* d_wav, int B, float* d_out) {
    const bool pow2 = p.log2_nfft > 0;
    const size_t lds = f64_lds_bytes(p.nfft, pow2);
    auto kernel = pow2 ? kws_mfcc_f64_kernel<T, true> : kws_mfcc_f64_kernel<T, false>;
    hipError_t e = raise_lds_limit(kernel, lds);
    if (e != hipSuccess) return e;
    dim3 grid((p.num_frames + 1) / 2, 1);
    for (int b0 = 0; b0 < B; b0 += 65535) {
        grid.y = (B - b0 < 65535) ? (B - b0) : 65535;
        hipLaunchKernelGGL(kernel, grid, dim3(64), lds, s, p, t, d_wav + (size_t)b0 * p.n_samples,
                           d_out + (size_t)b0 * p.num_frames * p.numcep);
    }
    return hipGetLastError();
}

}  // namespace

hipError_t launch_mfcc_f64(hipStream_t s, const FrontendParams& p, const FrontendTables& t, const int16_t* d_wav, int B, float* d_out) {
    return launch_mfcc_f64_t(s, p, t, d_wav, B, d_out);
}
hipError_t launch_mfcc_f64_f32in(hipStream_t s, const FrontendParams& p, const FrontendTables& t, const float* d_wav, int B,
                                 float* d_out) {
    return launch_mfcc_f64_t(s, p, t, d_wav, B, d_out);
}

hipError_t launch_spec_f64(hipStream_t s, const double* d_tw64, const float* d_frames, int num_frames, int frame_len, int nfft,
                           int log2n, int power, float* d_spec) {
    const bool pow2 = log2n > 0;
    const size_t lds = f64_lds_bytes(nfft, pow2);
    auto kernel = pow2 ? kws_spec_f64_kernel<true> : kws_spec_f64_kernel<false>;
    hipError_t e = raise_lds_limit(kernel, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3((num_frames + 1) / 2), dim3(64), lds, s, d_frames, num_frames, frame_len, nfft, log2n, power,
                       reinterpret_cast<const d2*>(d_tw64), d_spec);
    return hipGetLastError();
}

}  // namespace kws
