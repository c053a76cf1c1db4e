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

// for (i = lane; i < n; i += 64) body(i) -- with the bound known at compile time (BOUND > 0, n <= BOUND) the loop has a
// fixed trip count and unrolls, so the loads of all its iterations are issued together.
template <int BOUND, typename F>
__device__ __forceinline__ void lane_loop(int lane, int n, F&& body) {
    if constexpr (BOUND > 0) {
#pragma unroll
        for (int it = 0; it < (BOUND + 63) / 64; ++it) {
            const int i = lane + 64 * it;
            if (i < n) body(i);
        }
    } else {
        for (int i = lane; i < n; i += 64) body(i);
    }
}

// Spectra of the frame pair whose samples sit in X[n] = (a[n], b[n]), n < nfft (zero beyond the frame): writes
// P[k] = (|A[k]|^2, |B[k]|^2) * scale for k = 0..nfft/2 (scale = 1/nfft for the power spectrum, 1 for |X|^2).
// One wavefront; LDS instructions of a wavefront execute in order, so __syncthreads() (a single wave) only keeps the
// compiler from reordering across the stages.
// NCT: the transform length when it is known at compile time (512, the reference geometry: every loop unrolls and the
// loads of a stage are issued together), or 0 for a run-time length.  tw_lds: the twiddle table staged in LDS by the caller
// (nfft/2 entries for the FFT, nfft for the direct DFT).
template <bool POW2, int NCT>
__device__ __forceinline__ void spectrum_pair(d2* X, d2* P, const d2* tw_lds, int nfft_rt, int log2n_rt, int n_used, double scale,
                                              int lane) {
    const int nfft = NCT ? NCT : nfft_rt;
    const int log2n = NCT ? (31 - __builtin_clz(NCT ? NCT : 1)) : log2n_rt;
    const int nb = nfft / 2 + 1;
    if constexpr (POW2) {
        // radix-2 decimation in frequency, in place: stage with half-size s pairs (j, j + s) and multiplies the
        // difference by W_nfft^((i mod s) * nfft / (2 s))
        for (int s = nfft >> 1, step = 1; s >= 1; s >>= 1, step <<= 1) {
            lane_loop<NCT / 2>(lane, nfft >> 1, [&](int i) {
                const int r = i & (s - 1);
                const int j = ((i - r) << 1) + r;
                const d2 a = X[j], b = X[j + s];
                const d2 w = tw_lds[r * step];
                const d2 d = a - b;
                X[j] = a + b;
                X[j + s] = d2{__builtin_fma(d.x, w.x, -(d.y * w.y)), __builtin_fma(d.x, w.y, d.y * w.x)};
            });
            __syncthreads();
        }
        lane_loop<NCT ? NCT / 2 + 1 : 0>(lane, nb, [&](int k) {
            const int km = (nfft - k) & (nfft - 1);
            const d2 z = X[__brev((unsigned)k) >> (32 - log2n)], w = X[__brev((unsigned)km) >> (32 - log2n)];
            const double ar = 0.5 * (z.x + w.x), ai = 0.5 * (z.y - w.y);  // A = (Z[k] + conj Z[N-k]) / 2
            const double br = 0.5 * (z.y + w.y), bi = 0.5 * (w.x - z.x);  // B = (Z[k] - conj Z[N-k]) / (2i)
            P[k] = d2{__builtin_fma(ar, ar, ai * ai) * scale, __builtin_fma(br, br, bi * bi) * scale};
        });
    } else {
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

template <typename T, bool POW2, int NCT>
__global__ __launch_bounds__(64) void kws_mfcc_f64_kernel(FrontendParams p, FrontendTables t, const T* __restrict__ wav,
                                                         float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem64[];
    const int nfft = NCT ? NCT : p.nfft, nb = nfft / 2 + 1;
    // LDS: X [nfft] | L [2][64] | twiddles | P [nfft/2+1].  The FFT is done with its twiddles when the powers are formed,
    // so there P takes the twiddles' place (13 KB per wavefront instead of 17 at nfft = 512: 12 wavefronts per CU, not 9);
    // the direct DFT reads its table while it writes P.
    d2* X = reinterpret_cast<d2*>(smem64);
    double* L = reinterpret_cast<double*>(X + nfft);      // [2][64] log-mel
    d2* tw_lds = reinterpret_cast<d2*>(L + 128);          // [nfft/2] (FFT) or [nfft] (direct DFT)
    d2* P = POW2 ? tw_lds : tw_lds + nfft;
    const int lane = threadIdx.x;
    const int clip = blockIdx.y;
    const int fa = 2 * blockIdx.x;
    const bool has_b = fa + 1 < p.num_frames;
    const T* __restrict__ x = wav + (size_t)clip * p.n_samples;
    const int n_used = p.frame_len < nfft ? p.frame_len : nfft;  // frames longer than nfft are truncated (np.fft.rfft(frames, NFFT))

    const d2* __restrict__ twg = reinterpret_cast<const d2*>(t.tw64);
    const int n_tw = POW2 ? nfft / 2 : nfft;
    lane_loop<POW2 ? NCT / 2 : NCT>(lane, n_tw, [&](int i) { tw_lds[i] = twg[i]; });
    const long sa = (long)fa * p.frame_step, sb = sa + p.frame_step;
    bool nza = false, nzb = false;
    lane_loop<NCT>(lane, nfft, [&](int n) {
        double a = 0.0, b = 0.0;
        if (n < n_used) {
            a = (double)preemph_sample(x, sa + n, p.n_samples, p.preemph);
            if (has_b) b = (double)preemph_sample(x, sb + n, p.n_samples, p.preemph);
        }
        nza |= a != 0.0;
        nzb |= b != 0.0;
        X[n] = d2{a, b};
    });
    nza = __any(nza);
    nzb = __any(nzb);
    __syncthreads();
    spectrum_pair<POW2, NCT>(X, P, tw_lds, nfft, p.log2_nfft, n_used, 1.0 / (double)nfft, lane);
    // An all-zero frame must give an exactly zero spectrum (the reference then floors to eps); separated from its partner
    // in the packed transform it would keep the partner's rounding residue (1e-17 of it) instead.
    if (!(nza && nzb)) {
        for (int k = lane; k < nb; k += 64) {
            d2 pw = P[k];
            if (!nza) pw.x = 0.0;
            if (!nzb) pw.y = 0.0;
            P[k] = pw;
        }
        __syncthreads();
    }

    // frame energy = sum over all bins (psf fbank), zero -> eps
    double ea = 0.0, eb = 0.0;
    lane_loop<NCT ? NCT / 2 + 1 : 0>(lane, nb, [&](int k) {
        const d2 pw = P[k];
        ea += pw.x;
        eb += pw.y;
    });
    ea = wave_sum_f64(ea);
    eb = wave_sum_f64(eb);
    if (ea == 0.0) ea = PSF_EPS64;
    if (eb == 0.0) eb = PSF_EPS64;

    // Mel filter j = rising edge over [e_j, e_j+1) + falling edge over [e_j+1, e_j+2); the per-bin weights come from the
    // host (mel_w64: formed in float64 with psf's own divisions).  Lane j sums the rising part, lane 32 + j the falling
    // part (nfilt <= 32; more filters: lane j does both), then the halves meet through one shuffle.
    const bool split = p.nfilt <= 32;
    const int j = split ? (lane & 31) : lane;
    double fa_ = 0.0, fb_ = 0.0;
    if (j < p.nfilt) {
        const int e0 = t.mel_edges[j], e1 = t.mel_edges[j + 1], e2 = t.mel_edges[j + 2];
        const double* __restrict__ rise = t.mel_w64;
        const double* __restrict__ fall = t.mel_w64 + nb;
        if (!split || lane < 32)
            for (int i = e0; i < e1; ++i) {
                const double w = rise[i];
                const d2 pw = P[i];
                fa_ = __builtin_fma(w, pw.x, fa_);
                fb_ = __builtin_fma(w, pw.y, fb_);
            }
        if (!split || lane >= 32)
            for (int i = e1; i < e2; ++i) {
                const double w = fall[i];
                const d2 pw = P[i];
                fa_ = __builtin_fma(w, pw.x, fa_);
                fb_ = __builtin_fma(w, pw.y, fb_);
            }
    }
    if (split) {
        fa_ += __shfl_xor(fa_, 32, 64);
        fb_ += __shfl_xor(fb_, 32, 64);
    }
    double la = 0.0, lb = 0.0;
    if (lane < p.nfilt) {
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
        for (int q = 0; q < p.nfilt; ++q) acc = __builtin_fma(D[q], Lf[q], acc);
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
    d2* tw_lds = X + nfft;
    d2* P = POW2 ? tw_lds : tw_lds + nfft;  // as in the MFCC kernel: the powers replace the FFT's twiddles
    const int lane = threadIdx.x;
    const int fa = 2 * blockIdx.x;
    const bool has_b = fa + 1 < num_frames;
    const int n_used = frame_len < nfft ? frame_len : nfft;
    bool nza = false, nzb = false;
    for (int n = lane; n < nfft; n += 64) {
        double a = 0.0, b = 0.0;
        if (n < n_used) {
            a = (double)frames[(size_t)fa * frame_len + n];
            if (has_b) b = (double)frames[(size_t)(fa + 1) * frame_len + n];
        }
        nza |= a != 0.0;
        nzb |= b != 0.0;
        X[n] = d2{a, b};
    }
    nza = __any(nza);
    nzb = __any(nzb);
    for (int i = lane; i < (POW2 ? nfft / 2 : nfft); i += 64) tw_lds[i] = tw[i];
    __syncthreads();
    spectrum_pair<POW2, 0>(X, P, tw_lds, nfft, log2n, n_used, power ? 1.0 / (double)nfft : 1.0, lane);
    for (int k = lane; k < nb; k += 64) {
        d2 pw = P[k];
        if (!nza) pw.x = 0.0;  // an all-zero frame has an exactly zero spectrum, whatever shares its transform
        if (!nzb) pw.y = 0.0;
        spec[(size_t)fa * nb + k] = (float)(power ? pw.x : sqrt(pw.x));
        if (has_b) spec[(size_t)(fa + 1) * nb + k] = (float)(power ? pw.y : sqrt(pw.y));
    }
}

size_t f64_lds_bytes(int nfft, bool pow2) {
    const size_t nb = nfft / 2 + 1;
    return sizeof(d2) * ((size_t)nfft + (pow2 ? nb : nb + (size_t)nfft)) + sizeof(double) * 128;
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
    auto kernel = p.nfft == 512 ? kws_mfcc_f64_kernel<T, true, 512> : pow2 ? kws_mfcc_f64_kernel<T, true, 0> : kws_mfcc_f64_kernel<T, false, 0>;
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
