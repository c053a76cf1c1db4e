// Device-side building blocks of the float64 front end for gfx950: the 512-point radix-8 / any-length transform of a packed
// frame pair in LDS (spectrum_pair) and the energy / mel / log / DCT tail (f64_tail), shared by the batched float64 kernel,
// the selective-refinement kernel (kws_mfcc_f64.hip) and the streaming push, whose flagged frames are redone in float64 by
// the wavefront that computed them (kws_mfcc_dev.h: stream_frame_wave).  Anonymous namespace: each translation unit gets
// its own inlined copy.  See kws_mfcc_f64.hip for the work decomposition and the numerics (psf's float64 arithmetic at
// the reference call site kws/libs/audio_processor.py:270-278).
#pragma once
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


// Samples of the frame pair (frame a = samples [sa, sa + frame_len) of clip xa, frame b likewise; pre-emphasised in float32
// as NumPy does, zero beyond the frame and the clip) -> X[n] = (a[n], b[n]), n < nfft.  Returns through nza / nzb whether
// the frames hold any non-zero sample.
template <typename T, int NCT>
__device__ __forceinline__ void f64_load_pair(const FrontendParams& p, const T* __restrict__ xa, long sa, const T* __restrict__ xb,
                                              long sb, bool has_b, d2* X, int nfft, int n_used, int lane, bool& nza, bool& nzb) {
    nza = false;
    nzb = false;
    if (sizeof(T) == 2 && p.vec_ok && (p.frame_step % 8) == 0) {
        // 8 consecutive samples per lane and frame: one 16-byte load (every group is 16-byte aligned here), the sample
        // before the group from the neighbouring lane's registers
        for (int g = lane; 8 * g < nfft; g += 64) {
            double va[8], vb[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) va[i] = vb[i] = 0.0;
            auto load8 = [&](const T* __restrict__ x, long s0, double (&v)[8]) {
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
            load8(xa, sa, va);
            if (has_b) load8(xb, sb, vb);
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
                a = (double)preemph_sample(xa, sa + n, p.n_samples, p.preemph);
                if (has_b) b = (double)preemph_sample(xb, sb + n, p.n_samples, p.preemph);
            }
            nza |= a != 0.0;
            nzb |= b != 0.0;
            X[NCT == 512 ? sw512(n) : n] = d2{a, b};
        });
    }
    nza = __any(nza);
    nzb = __any(nzb);
    wave_order();
}

// Tables of the float64 tail: LDS copies (batched kernels) or the global tables themselves (the rare streaming redo).
struct F64Tabs {
    const d2* tw;          // W_nfft^m: m < nfft/2 for a power-of-two length, m < nfft for the direct DFT
    const double* melw;    // [2][nb]: rising / falling weight of every bin
    const double* dct;     // [numcep][nfilt]
    const int* edges;      // [nfilt + 2]
};

// From the powers P[k] = (|A[k]|^2, |B[k]|^2)/nfft of a frame pair to its cepstra: frame energy (zero -> eps), triangular
// mel filters from psf's bin edges, eps floors, log, DCT-II(ortho) x lifter, c0 = log(energy); cast to float32.
// NB: compile-time number of bins (nfft/2 + 1) or 0.  L: 2 x 64 doubles of LDS private to the wavefront.
// lds_out_a: when not null, frame a's row is also written there (the streaming push: the LDS feature map).
template <int NB>
__device__ __forceinline__ void f64_tail(const FrontendParams& p, const F64Tabs& tb, int nb_rt, const d2* P, double* L, bool nza,
                                         bool nzb, bool has_b, float* __restrict__ out_a, float* __restrict__ out_b,
                                         float* lds_out_a, int lane) {
    const int nb = NB ? NB : nb_rt;
    // frame energy = sum over all bins (psf fbank), zero -> eps.  An all-zero frame must give an exactly zero spectrum
    // (the reference then floors to eps); separated from its partner in the packed transform it would keep the partner's
    // rounding residue (1e-17 of it) instead, so its powers are taken as zero.
    double ea = 0.0, eb = 0.0;
    lane_loop<NB>(lane, nb, [&](int k) {
        const d2 pw = P[k];
        ea += pw.x;
        eb += pw.y;
    });
    ea = nza ? wave_sum_f64(ea) : 0.0;
    eb = nzb ? wave_sum_f64(eb) : 0.0;
    if (ea == 0.0) ea = PSF_EPS64;
    if (eb == 0.0) eb = PSF_EPS64;
#ifdef KWS_X_F64_NOTAIL
    if (out_a && lane < p.numcep) out_a[lane] = (float)(ea + eb);  // timing ablation (out_a is null for a row the refinement does not rewrite)
    return;
#endif
    // Mel filter j = rising edge over [e_j, e_j+1) + falling edge over [e_j+1, e_j+2); the per-bin weights come from
    // the host (formed in float64 with psf's own divisions).  Lane j sums the rising part, lane 32 + j the falling part
    // (nfilt <= 32; more filters: lane j does both), then the halves meet through one shuffle.  Four bins per trip with
    // their own accumulators: the widest filter has ~50 bins per edge, and one bin per trip left every multiply-add
    // waiting for its own pair of LDS reads (a third of a lone frame's latency in the refinement kernel).
    const bool split = p.nfilt <= 32;
    const int j = split ? (lane & 31) : lane;
    double fa_ = 0.0, fb_ = 0.0;
    if (j < p.nfilt) {
        const int e0 = tb.edges[j], e1 = tb.edges[j + 1], e2 = tb.edges[j + 2];
        auto edge_sum = [&](const double* __restrict__ wgt, int lo, int hi) {
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
            int i = lo;
            for (; i + 4 <= hi; i += 4) {
                const double w0 = wgt[i], w1 = wgt[i + 1], w2 = wgt[i + 2], w3 = wgt[i + 3];
                const d2 p0 = P[i], p1 = P[i + 1], p2 = P[i + 2], p3 = P[i + 3];
                a0 = __builtin_fma(w0, p0.x, a0);
                b0 = __builtin_fma(w0, p0.y, b0);
                a1 = __builtin_fma(w1, p1.x, a1);
                b1 = __builtin_fma(w1, p1.y, b1);
                a2 = __builtin_fma(w2, p2.x, a2);
                b2 = __builtin_fma(w2, p2.y, b2);
                a3 = __builtin_fma(w3, p3.x, a3);
                b3 = __builtin_fma(w3, p3.y, b3);
            }
            for (; i < hi; ++i) {
                const double w = wgt[i];
                const d2 pw = P[i];
                a0 = __builtin_fma(w, pw.x, a0);
                b0 = __builtin_fma(w, pw.y, b0);
            }
            fa_ += (a0 + a1) + (a2 + a3);
            fb_ += (b0 + b1) + (b2 + b3);
        };
        if (!split || lane < 32) edge_sum(tb.melw, e0, e1);
        if (!split || lane >= 32) edge_sum(tb.melw + nb, e1, e2);
    }
    if (split) {  // lane j += lane j + 32 (and vice versa): the two halves of the wavefront swap through v_permlane32_swap
        auto other_half = [](double v) {
            const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
            unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32), lo2 = lo, hi2 = hi;
            // lo: lanes >= 32 get lo2 of lanes < 32; lo2: lanes < 32 get lo of lanes >= 32.  One block, led by two wait
            // states: a VALU write followed by a permlane swap of the same register needs them (the compiler inserts
            // `s_nop 1` for its own swaps but does not look into asm -- found by tools/isa_hazard_lint.py, rule R2).
            asm volatile("s_nop 1\n\t"
                         "v_permlane32_swap_b32 %0, %1\n\t"
                         "v_permlane32_swap_b32 %2, %3"
                         : "+v"(lo), "+v"(lo2), "+v"(hi), "+v"(hi2));
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
    float* __restrict__ const row = f ? out_b : out_a;  // a null row is computed but not stored (the refinement stores only listed frames)
    if (i < p.numcep && (f == 0 || has_b) && (row || (lds_out_a && f == 0))) {
        const double* D = tb.dct + (size_t)i * p.nfilt;
        const double* Lf = L + 64 * f;
        double acc0 = 0.0, acc1 = 0.0;  // two chains, eight terms in flight per trip
        int q = 0;
        for (; q + 8 <= p.nfilt; q += 8) {
            double dv[8], lv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) dv[u] = D[q + u], lv[u] = Lf[q + u];
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                acc0 = __builtin_fma(dv[u], lv[u], acc0);
                acc1 = __builtin_fma(dv[u + 1], lv[u + 1], acc1);
            }
        }
        for (; q < p.nfilt; ++q) acc0 = __builtin_fma(D[q], Lf[q], acc0);
        double acc = acc0 + acc1;
        if (i == 0 && p.append_energy) acc = log(f ? eb : ea);
        if (row) row[i] = (float)acc;
        if (lds_out_a && f == 0) lds_out_a[i] = (float)acc;
    }
    wave_order();
}

}  // namespace
}  // namespace kws
