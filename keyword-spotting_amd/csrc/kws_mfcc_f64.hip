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

// LDS instructions of one wavefront execute in order: data written by one lane is visible to a later read of another lane
// of the SAME wavefront without a barrier; only the compiler has to keep the order.
__device__ __forceinline__ void wave_order() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Full-wavefront sum of a double without touching LDS (a __shfl_xor is a ds_bpermute per dword: 24 of them per frame
// pair were a quarter of this kernel's LDS time): DPP moves of both halves -- scan inside the 16-lane rows, fold the rows,
// broadcast lane 63.  Zeros are shifted in at the row ends.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_shift_add_f64(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, ROW_MASK, 0xf, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, ROW_MASK, 0xf, true);
    return v + __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double wave_sum_f64(double v) {
    v = dpp_shift_add_f64<0x111, 0xf>(v);  // row_shr:1
    v = dpp_shift_add_f64<0x112, 0xf>(v);  // row_shr:2
    v = dpp_shift_add_f64<0x114, 0xf>(v);  // row_shr:4
    v = dpp_shift_add_f64<0x118, 0xf>(v);  // row_shr:8
    v = dpp_shift_add_f64<0x142, 0xa>(v);  // row_bcast:15 -> rows 1, 3
    v = dpp_shift_add_f64<0x143, 0xc>(v);  // row_bcast:31 -> rows 2, 3
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), 63);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
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

__device__ __forceinline__ d2 cmul64(d2 a, d2 b) {
    return d2{__builtin_fma(a.x, b.x, -(a.y * b.y)), __builtin_fma(a.x, b.y, a.y * b.x)};
}
// a + (-i) b and a - (-i) b
__device__ __forceinline__ d2 add_mi64(d2 a, d2 b) { return d2{a.x + b.y, a.y - b.x}; }
__device__ __forceinline__ d2 sub_mi64(d2 a, d2 b) { return d2{a.x - b.y, a.y + b.x}; }

// In-place 8-point forward DFT, natural order in and out: v[k] = sum_n v[n] exp(-2 pi i n k / 8).
__device__ __forceinline__ void dft8_f64(d2 (&v)[8]) {
    constexpr double R = 0.70710678118654752440;
    const d2 b0 = v[0] + v[4], b4 = v[0] - v[4];
    const d2 b1 = v[1] + v[5], c5 = v[1] - v[5];
    const d2 b2 = v[2] + v[6], b6 = v[2] - v[6];
    const d2 b3 = v[3] + v[7], c7 = v[3] - v[7];
    const d2 b5 = d2{(c5.x + c5.y) * R, (c5.y - c5.x) * R};    // * (1 - i)/sqrt2
    const d2 b7 = d2{(c7.y - c7.x) * R, -(c7.x + c7.y) * R};   // * (-1 - i)/sqrt2
    const d2 d0 = b0 + b2, d1 = b0 - b2, d2_ = b1 + b3, d3 = b1 - b3;
    v[0] = d0 + d2_;
    v[4] = d0 - d2_;
    v[2] = add_mi64(d1, d3);
    v[6] = sub_mi64(d1, d3);
    const d2 e0 = add_mi64(b4, b6), e1 = sub_mi64(b4, b6), e2 = b5 + b7, e3 = b5 - b7;
    v[1] = e0 + e2;
    v[5] = e0 - e2;
    v[3] = add_mi64(e1, e3);
    v[7] = sub_mi64(e1, e3);
}

// LDS index swizzle of the 512-point path: element p lives at p ^ ((p >> 3) & 7).  The third radix-8 stage and the sample
// staging touch X with a 128-byte stride between lanes (lane l: elements 8l .. 8l+7), which puts the eight lanes of a
// 16-byte access group on the same four banks; XORing the low three index bits with the next three spreads them over all
// 32 banks and leaves the other stages' unit-stride patterns a permutation inside aligned groups of eight.
__device__ __forceinline__ int sw512(int p) { return p ^ ((p >> 3) & 7); }

// 512-point decimation-in-frequency FFT as three radix-8 stages in LDS (one butterfly per lane and stage: 8 reads, 7
// twiddles, 8 writes -- a third of the LDS traffic of nine radix-2 stages, which is what bounds this kernel).  In place;
// position p = (d2 d1 d0)_8 ends up holding bin (d0 d1 d2)_8.  tw: W_512^m for m < 256 (W^(m + 256) = -W^m).
__device__ __forceinline__ void fft512_radix8(d2* X, const d2* tw, int lane) {
#pragma unroll
    for (int stage = 0; stage < 3; ++stage) {
        const int s = 64 >> (3 * stage);        // distance between the inputs of a butterfly: 64, 8, 1
        const int stride = 1 << (3 * stage);    // twiddle exponent step: N / (8 s)
        const int r = lane & (s - 1);
        const int j = ((lane - r) << 3) + r;
        d2 v[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) v[m] = X[sw512(j + m * s)];
        dft8_f64(v);
        if (stage < 2) {
#pragma unroll
            for (int q = 1; q < 8; ++q) {
                const int e = r * q * stride;   // < 7/8 * 512
                d2 w = tw[e & 255];
                if (e & 256) w = -w;
                v[q] = cmul64(v[q], w);
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) X[sw512(j + q * s)] = v[q];
        wave_order();
    }
}
__device__ __forceinline__ int digit_rev8x3(int k) { return sw512(((k & 7) << 6) | (k & 56) | (k >> 6)); }

// Spectra of the frame pair whose samples sit in X[n] = (a[n], b[n]), n < nfft (zero beyond the frame): writes
// P[k] = (|A[k]|^2, |B[k]|^2) * scale for k = 0..nfft/2 (scale = 1/nfft for the power spectrum, 1 for |X|^2).
// One wavefront on LDS private to it (X, P) and a twiddle table shared by the workgroup.
// NCT: the transform length when it is known at compile time (512, the reference geometry: every loop unrolls and the
// loads of a stage are issued together), or 0 for a run-time length.  With NCT > 0 P may ALIAS X: every lane reads all the
// spectrum values it needs into registers before the first power is written.
template <bool POW2, int NCT>
__device__ __forceinline__ void spectrum_pair(d2* X, d2* P, const d2* tw_lds, int nfft_rt, int log2n_rt, int n_used, double scale,
                                              int lane) {
    const int nfft = NCT ? NCT : nfft_rt;
    const int log2n = NCT ? (31 - __builtin_clz(NCT ? NCT : 1)) : log2n_rt;
    const int nb = nfft / 2 + 1;
    if constexpr (POW2) {
        auto where = [&](int k) { return NCT == 512 ? digit_rev8x3(k) : (int)(__brev((unsigned)k) >> (32 - log2n)); };
        if constexpr (NCT == 512) {
#ifndef KWS_X_F64_NOFFT
            fft512_radix8(X, tw_lds, lane);
#endif
        } else {
        // radix-2 decimation in frequency, in place: stage with half-size s pairs (j, j + s) and multiplies the
        // difference by W_nfft^((i mod s) * nfft / (2 s))
#ifdef KWS_X_F64_NOFFT
        for (int s = nfft >> 1, step = 1; s >= nfft; s >>= 1, step <<= 1) {  // timing ablation: no transform stages (wrong results)
#else
#ifdef KWS_X_F64_ROLLED
#pragma unroll 1
#endif
        for (int s = nfft >> 1, step = 1; s >= 1; s >>= 1, step <<= 1) {
#endif
            lane_loop<NCT / 2>(lane, nfft >> 1, [&](int i) {
                const int r = i & (s - 1);
                const int j = ((i - r) << 1) + r;
                const d2 a = X[j], b = X[j + s];
                const d2 w = tw_lds[r * step];
                const d2 d = a - b;
                X[j] = a + b;
                X[j + s] = d2{__builtin_fma(d.x, w.x, -(d.y * w.y)), __builtin_fma(d.x, w.y, d.y * w.x)};
            });
            wave_order();
        }
        }
        auto power_of = [&](const d2 z, const d2 w) {
            const double ar = 0.5 * (z.x + w.x), ai = 0.5 * (z.y - w.y);  // A = (Z[k] + conj Z[N-k]) / 2
            const double br = 0.5 * (z.y + w.y), bi = 0.5 * (w.x - z.x);  // B = (Z[k] - conj Z[N-k]) / (2i)
            return d2{__builtin_fma(ar, ar, ai * ai) * scale, __builtin_fma(br, br, bi * bi) * scale};
        };
        if constexpr (NCT > 0) {
            constexpr int IT = (NCT / 2 + 1 + 63) / 64;
            d2 z[IT], w[IT];
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int k = lane + 64 * it;
                if (k < nb) {
                    z[it] = X[where(k)];
                    w[it] = X[where((nfft - k) & (nfft - 1))];
                }
            }
            wave_order();  // all reads of the spectrum are issued before the first write of a power (P may alias X)
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int k = lane + 64 * it;
                if (k < nb) P[k] = power_of(z[it], w[it]);
            }
        } else {
            for (int k = lane; k < nb; k += 64)
                P[k] = power_of(X[where(k)], X[where((nfft - k) & (nfft - 1))]);
        }
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
    wave_order();
}

constexpr int F64_MAX_WAVES = 4;         // wavefronts per workgroup of the MFCC kernel (fewer when a long transform fills the LDS)
constexpr int F64_PAIRS_PER_WAVE = 4;    // consecutive frame pairs per wavefront
constexpr int F64_STAGE_MELW_MAX_NFFT = 1024;  // beyond it the per-bin mel weights are read from global memory, not staged

// Shared LDS tables of a workgroup: twiddles, per-bin mel weights, DCT x lifter, mel edges -- staged once, used by every
// frame pair of the workgroup.  Per wavefront: X [nfft] (+ P [nfft/2+1] unless it aliases X) and the log-mel vectors.
struct F64Layout {
    size_t tw, melw, dct, edges, wave0, per_wave, total;
};
__host__ __device__ inline F64Layout f64_layout(int nfft, bool pow2, bool alias_p, int nfilt, int numcep, int n_waves) {
    const size_t nb = nfft / 2 + 1, n_tw = pow2 ? nfft / 2 : nfft;
    F64Layout l;
    l.tw = 0;
    l.melw = l.tw + 16 * n_tw;
    l.dct = l.melw + (nfft <= F64_STAGE_MELW_MAX_NFFT ? 8 * 2 * nb : 0);
    l.edges = l.dct + 8 * (size_t)numcep * nfilt;
    l.wave0 = (l.edges + 4 * (size_t)(nfilt + 2) + 15) & ~(size_t)15;
    l.per_wave = 16 * ((size_t)nfft + (alias_p ? 0 : nb)) + 8 * 128;
    l.total = l.wave0 + (size_t)n_waves * l.per_wave;
    return l;
}

template <typename T, bool POW2, int NCT>
__global__ __launch_bounds__(F64_MAX_WAVES * 64) void kws_mfcc_f64_kernel(FrontendParams p, FrontendTables t, const T* __restrict__ wav,
                                                                      float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem64[];
    const int nfft = NCT ? NCT : p.nfft, nb = nfft / 2 + 1;
    constexpr bool ALIAS_P = POW2 && NCT > 0;
    const int n_waves = blockDim.x >> 6, n_threads = blockDim.x;
    const F64Layout lay = f64_layout(nfft, POW2, ALIAS_P, p.nfilt, p.numcep, n_waves);
    const bool stage_melw = nfft <= F64_STAGE_MELW_MAX_NFFT;
    d2* tw_lds = reinterpret_cast<d2*>(smem64 + lay.tw);
    const double* melw = stage_melw ? reinterpret_cast<const double*>(smem64 + lay.melw) : t.mel_w64;  // [2][nb]: rising / falling weight of every bin
    double* dct = reinterpret_cast<double*>(smem64 + lay.dct);     // [numcep][nfilt]
    int* edges = reinterpret_cast<int*>(smem64 + lay.edges);       // [nfilt + 2]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    unsigned char* mine = smem64 + lay.wave0 + wv * lay.per_wave;
    d2* X = reinterpret_cast<d2*>(mine);
    d2* P = ALIAS_P ? X : X + nfft;
    double* L = reinterpret_cast<double*>(mine + lay.per_wave - 8 * 128);  // [2][64] log-mel

    {   // stage the shared tables
        const d2* __restrict__ twg = reinterpret_cast<const d2*>(t.tw64);
        const int n_tw = POW2 ? nfft / 2 : nfft;
        for (int i = tid; i < n_tw; i += n_threads) tw_lds[i] = twg[i];
        if (stage_melw)
            for (int i = tid; i < 2 * nb; i += n_threads) reinterpret_cast<double*>(smem64 + lay.melw)[i] = t.mel_w64[i];
        for (int i = tid; i < p.numcep * p.nfilt; i += n_threads) dct[i] = t.dct64[i];
        for (int i = tid; i < p.nfilt + 2; i += n_threads) edges[i] = t.mel_edges[i];
    }
    __syncthreads();  // the only workgroup barrier: after it the wavefronts work on private LDS

    const int clip = blockIdx.y;
    const T* __restrict__ x = wav + (size_t)clip * p.n_samples;
    const int n_used = p.frame_len < nfft ? p.frame_len : nfft;  // frames longer than nfft are truncated (np.fft.rfft(frames, NFFT))
    const int n_pairs = (p.num_frames + 1) / 2;
    const int pairs_per_wg = (n_pairs + gridDim.x - 1) / gridDim.x;  // the clip's pairs spread evenly over its workgroups
    for (int pr = blockIdx.x * pairs_per_wg + wv; pr < n_pairs && pr < (blockIdx.x + 1) * pairs_per_wg; pr += n_waves) {
        const int fa = 2 * pr;
        const bool has_b = fa + 1 < p.num_frames;
        const long sa = (long)fa * p.frame_step, sb = sa + p.frame_step;
        bool nza = false, nzb = false;
        if (sizeof(T) == 2 && p.vec_ok && (p.frame_step % 8) == 0) {
            // 8 consecutive samples per lane and frame: one 16-byte load (every group is 16-byte aligned here), the sample
            // before the group from the neighbouring lane's registers
            for (int g = lane; 8 * g < nfft; g += 64) {
                double va[8], vb[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) va[i] = vb[i] = 0.0;
                auto load8 = [&](long s0, double (&v)[8]) {
                    const long m0 = s0 + 8 * g;
                    if (8 * g < n_used && m0 < p.n_samples) {  // n_samples % 8 == 0: a group is inside the clip or outside it
                        const uint4 raw = *reinterpret_cast<const uint4*>(x + m0);
                        const uint32_t wd[4] = {raw.x, raw.y, raw.z, raw.w};
                        float prev = m0 > 0 ? to_unit64(x[m0 - 1]) : 0.f;
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const float cur = to_unit64((int16_t)((wd[i >> 1] >> (16 * (i & 1))) & 0xffffu));
                            const float y = (m0 + i > 0) ? __fsub_rn(cur, __fmul_rn(p.preemph, prev)) : cur;
                            v[i] = (8 * g + i < n_used) ? (double)y : 0.0;
                            prev = cur;
                        }
                    }
                };
                load8(sa, va);
                if (has_b) load8(sb, vb);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    nza |= va[i] != 0.0;
                    nzb |= vb[i] != 0.0;
                    X[NCT == 512 ? sw512(8 * g + i) : 8 * g + i] = d2{va[i], vb[i]};
                }
            }
        } else {
            lane_loop<NCT>(lane, nfft, [&](int n) {
                double a = 0.0, b = 0.0;
                if (n < n_used) {
                    a = (double)preemph_sample(x, sa + n, p.n_samples, p.preemph);
                    if (has_b) b = (double)preemph_sample(x, sb + n, p.n_samples, p.preemph);
                }
                nza |= a != 0.0;
                nzb |= b != 0.0;
                X[NCT == 512 ? sw512(n) : n] = d2{a, b};
            });
        }
        nza = __any(nza);
        nzb = __any(nzb);
        wave_order();
        spectrum_pair<POW2, NCT>(X, P, tw_lds, nfft, p.log2_nfft, n_used, 1.0 / (double)nfft, lane);

        // frame energy = sum over all bins (psf fbank), zero -> eps.  An all-zero frame must give an exactly zero spectrum
        // (the reference then floors to eps); separated from its partner in the packed transform it would keep the partner's
        // rounding residue (1e-17 of it) instead, so its powers are taken as zero.
        double ea = 0.0, eb = 0.0;
        lane_loop<NCT ? NCT / 2 + 1 : 0>(lane, nb, [&](int k) {
            const d2 pw = P[k];
            ea += pw.x;
            eb += pw.y;
        });
        ea = nza ? wave_sum_f64(ea) : 0.0;
        eb = nzb ? wave_sum_f64(eb) : 0.0;
        if (ea == 0.0) ea = PSF_EPS64;
        if (eb == 0.0) eb = PSF_EPS64;
#ifdef KWS_X_F64_NOTAIL
        if (lane < p.numcep) out[((size_t)clip * p.num_frames + fa) * p.numcep + lane] = (float)(ea + eb);  // timing ablation
        continue;
#endif
        // Mel filter j = rising edge over [e_j, e_j+1) + falling edge over [e_j+1, e_j+2); the per-bin weights come from
        // the host (formed in float64 with psf's own divisions).  Lane j sums the rising part, lane 32 + j the falling part
        // (nfilt <= 32; more filters: lane j does both), then the halves meet through one shuffle.
        const bool split = p.nfilt <= 32;
        const int j = split ? (lane & 31) : lane;
        double fa_ = 0.0, fb_ = 0.0;
        if (j < p.nfilt) {
            const int e0 = edges[j], e1 = edges[j + 1], e2 = edges[j + 2];
            const double* rise = melw;
            const double* fall = melw + nb;
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
        if (split) {  // lane j += lane j + 32 (and vice versa): the two halves of the wavefront swap through v_permlane32_swap
            auto other_half = [](double v) {
                const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
                unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32), lo2 = lo, hi2 = hi;
                asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(lo2));  // lo: lanes >= 32 get lo2 of lanes < 32; lo2: lanes < 32 get lo of lanes >= 32
                asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(hi), "+v"(hi2));
                const bool upper = (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) & 32) != 0;
                return __builtin_bit_cast(double, ((unsigned long long)(upper ? hi : hi2) << 32) | (upper ? lo : lo2));
            };
            fa_ += other_half(fa_);
            fb_ += other_half(fb_);
        }
        if (!nza) fa_ = 0.0;
        if (!nzb) fb_ = 0.0;
        double la = 0.0, lb = 0.0;
        if (lane < p.nfilt) {
            la = log(fa_ == 0.0 ? PSF_EPS64 : fa_);
            lb = log(fb_ == 0.0 ? PSF_EPS64 : fb_);
        }
        L[lane] = la;
        L[64 + lane] = lb;
        wave_order();

        // DCT-II(ortho) x lifter: lane -> (frame f = lane >> 5, coefficient i = lane & 31)
        const int f = lane >> 5, i = lane & 31;
        if (i < p.numcep && (f == 0 || has_b)) {
            const double* D = dct + (size_t)i * p.nfilt;
            const double* Lf = L + 64 * f;
            double acc = 0.0;
            for (int q = 0; q < p.nfilt; ++q) acc = __builtin_fma(D[q], Lf[q], acc);
            if (i == 0 && p.append_energy) acc = log(f ? eb : ea);
            out[((size_t)clip * p.num_frames + fa + f) * p.numcep + i] = (float)acc;
        }
        wave_order();
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
    wave_order();
    spectrum_pair<POW2, 0>(X, P, tw_lds, nfft, log2n, n_used, power ? 1.0 / (double)nfft : 1.0, lane);
    for (int k = lane; k < nb; k += 64) {
        d2 pw = P[k];
        if (!nza) pw.x = 0.0;  // an all-zero frame has an exactly zero spectrum, whatever shares its transform
        if (!nzb) pw.y = 0.0;
        spec[(size_t)fa * nb + k] = (float)(power ? pw.x : sqrt(pw.x));
        if (has_b) spec[(size_t)(fa + 1) * nb + k] = (float)(power ? pw.y : sqrt(pw.y));
    }
}

size_t spec_lds_bytes(int nfft, bool pow2) {
    const size_t nb = nfft / 2 + 1;
    return sizeof(d2) * ((size_t)nfft + nb + (pow2 ? (size_t)nfft / 2 : (size_t)nfft));
}

template <typename K>
hipError_t raise_lds_limit(K kernel, size_t lds) {
    if (lds <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

template <typename T>
hipError_t launch_mfcc_f64_t(hipStream_t s, const FrontendParams& p, const FrontendTables& t, const T* d_wav, int B, float* d_out) {
    const bool pow2 = p.log2_nfft > 0;
    const bool fixed = p.nfft == 512;
    int n_waves = F64_MAX_WAVES;  // as many wavefronts as the CU's 160 KiB of LDS hold at this transform length
    while (n_waves > 1 && f64_layout(p.nfft, pow2, pow2 && fixed, p.nfilt, p.numcep, n_waves).total > 160 * 1024) --n_waves;
    const size_t lds = f64_layout(p.nfft, pow2, pow2 && fixed, p.nfilt, p.numcep, n_waves).total;
    if (lds > 160 * 1024) return hipErrorInvalidValue;  // refused by kws_set_frontend before it gets here
    auto kernel = fixed ? kws_mfcc_f64_kernel<T, true, 512> : pow2 ? kws_mfcc_f64_kernel<T, true, 0> : kws_mfcc_f64_kernel<T, false, 0>;
    hipError_t e = raise_lds_limit(kernel, lds);
    if (e != hipSuccess) return e;
    const int n_pairs = (p.num_frames + 1) / 2;
    const int pairs_per_wg = F64_PAIRS_PER_WAVE * n_waves;
    dim3 grid((n_pairs + pairs_per_wg - 1) / pairs_per_wg, 1);
    for (int b0 = 0; b0 < B; b0 += 65535) {
        grid.y = (B - b0 < 65535) ? (B - b0) : 65535;
        hipLaunchKernelGGL(kernel, grid, dim3(n_waves * 64), lds, s, p, t, d_wav + (size_t)b0 * p.n_samples,
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
    const size_t lds = spec_lds_bytes(nfft, pow2);
    auto kernel = pow2 ? kws_spec_f64_kernel<true> : kws_spec_f64_kernel<false>;
    hipError_t e = raise_lds_limit(kernel, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3((num_frames + 1) / 2), dim3(64), lds, s, d_frames, num_frames, frame_len, nfft, log2n, power,
                       reinterpret_cast<const d2*>(d_tw64), d_spec);
    return hipGetLastError();
}

}  // namespace kws
