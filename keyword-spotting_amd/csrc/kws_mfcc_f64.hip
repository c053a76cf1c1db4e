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
#include "kws_mfcc_f64_dev.h"

namespace kws {
namespace {

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

// Workgroup-shared staging of the float64 tables into LDS and the wavefront's private views.
struct F64Wave {
    F64Tabs tb;
    d2* X;
    d2* P;
    double* L;
};
template <bool POW2, bool ALIAS_P>
__device__ __forceinline__ F64Wave f64_stage_tables(const FrontendParams& p, const FrontendTables& t, unsigned char* smem, int nfft) {
    const int nb = nfft / 2 + 1;
    const int n_waves = blockDim.x >> 6, n_threads = blockDim.x;
    const F64Layout lay = f64_layout(nfft, POW2, ALIAS_P, p.nfilt, p.numcep, n_waves);
    const bool stage_melw = nfft <= F64_STAGE_MELW_MAX_NFFT;
    d2* tw_lds = reinterpret_cast<d2*>(smem + lay.tw);
    double* dct = reinterpret_cast<double*>(smem + lay.dct);     // [numcep][nfilt]
    int* edges = reinterpret_cast<int*>(smem + lay.edges);       // [nfilt + 2]
    const int tid = threadIdx.x, wv = tid >> 6;
    const d2* __restrict__ twg = reinterpret_cast<const d2*>(t.tw64);
    const int n_tw = POW2 ? nfft / 2 : nfft;
    for (int i = tid; i < n_tw; i += n_threads) tw_lds[i] = twg[i];
    if (stage_melw)
        for (int i = tid; i < 2 * nb; i += n_threads) reinterpret_cast<double*>(smem + lay.melw)[i] = t.mel_w64[i];
    for (int i = tid; i < p.numcep * p.nfilt; i += n_threads) dct[i] = t.dct64[i];
    for (int i = tid; i < p.nfilt + 2; i += n_threads) edges[i] = t.mel_edges[i];
    unsigned char* mine = smem + lay.wave0 + wv * lay.per_wave;
    F64Wave w;
    w.tb.tw = tw_lds;
    w.tb.melw = stage_melw ? reinterpret_cast<const double*>(smem + lay.melw) : t.mel_w64;  // [2][nb]: rising / falling weight of every bin
    w.tb.dct = dct;
    w.tb.edges = edges;
    w.X = reinterpret_cast<d2*>(mine);
    w.P = ALIAS_P ? w.X : w.X + nfft;
    w.L = reinterpret_cast<double*>(mine + lay.per_wave - 8 * 128);  // [2][64] log-mel
    return w;
}

template <typename T, bool POW2, int NCT>
__global__ __launch_bounds__(F64_MAX_WAVES * 64) void kws_mfcc_f64_kernel(FrontendParams p, FrontendTables t, const T* __restrict__ wav,
                                                                      float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem64[];
    const int nfft = NCT ? NCT : p.nfft, nb = nfft / 2 + 1;
    constexpr bool ALIAS_P = POW2 && NCT > 0;
    const int n_waves = blockDim.x >> 6;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const F64Wave w = f64_stage_tables<POW2, ALIAS_P>(p, t, smem64, nfft);
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
        bool nza, nzb;
        f64_load_pair<T, NCT>(p, x, sa, x, sb, has_b, w.X, nfft, n_used, lane, nza, nzb);
        spectrum_pair<POW2, NCT>(w.X, w.P, w.tb.tw, nfft, p.log2_nfft, n_used, 1.0 / (double)nfft, lane);
        float* oa = out + ((size_t)clip * p.num_frames + fa) * p.numcep;
        f64_tail<NCT ? NCT / 2 + 1 : 0>(p, w.tb, nb, w.P, w.L, nza, nzb, has_b, oa, oa + p.numcep, nullptr, lane);
    }
}

// Selective refinement of the float32 front end.  The float32 kernel appends, for every frame pair with a
// frame whose log-mel vector spans more than the precision threshold, the entry (pair index << 2 | mask of the flagged
// frames) to rl.list and counts entries in rl.ctr[0]; this kernel recomputes those pairs in float64 and overwrites the
// rows of the flagged frames only.  No host read-back: the grid is fixed (three workgroups per CU), every wavefront
// strides over the list, and a workgroup with nothing to do leaves before it stages a table.  A listed frame is
// transformed with its natural partner (frames 2k, 2k+1 of the clip), as the float32 kernel and the always-float64 kernel
// pair them -- so its bits do not depend on the order the atomics filled the list in, and clustered flags (a tone, a
// vowel) cost one transform per two frames.  The last workgroup to finish folds the number of rows rewritten (ctr[1],
// counted by the float32 kernel) into the running total (ctr[2..3], 64 bits), keeps it as "last call" (ctr[4]) and clears
// the counters for the next launch.
#ifndef KWS_X_REFINE_WAVES
// 12: 3 072 wavefronts on 256 CUs take the ~2 200 flagged pairs of a 4096-clip noise batch (span threshold 11.5) in one round
// (8: a second round for some, 25.0 vs 20.5 us; 16 fits the LDS too but spills under the 128-register cap: 28.0 us)
#define KWS_X_REFINE_WAVES 12
#endif
constexpr int REFINE_WAVES = KWS_X_REFINE_WAVES;  // wavefronts per workgroup of the refinement kernel
template <typename T>
__global__ __launch_bounds__(REFINE_WAVES * 64) void kws_mfcc_refine_kernel(FrontendParams p, FrontendTables t, const T* __restrict__ wav,
                                                                         float* __restrict__ out, RefineList rl) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem64[];
    constexpr int NCT = 512;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n_waves = blockDim.x >> 6;
    const int raw = *const_cast<volatile int*>(rl.ctr);
    const int n = raw < rl.cap ? raw : rl.cap;
    // Entry e goes to workgroup e mod grid, wavefront e / grid: a short list (white noise: ~1 000 pairs for 3 072 wavefronts) is spread
    // one wavefront per SIMD over all CUs instead of filling a third of them three deep.
    if ((int)blockIdx.x < n) {  // workgroup-uniform
        const int n_used = p.frame_len < NCT ? p.frame_len : NCT;
        const int pairs_per_clip = (p.num_frames + 1) / 2;
        const F64Layout lay = f64_layout(NCT, true, true, p.nfilt, p.numcep, n_waves);
        d2* const X = reinterpret_cast<d2*>(smem64 + lay.wave0 + wv * lay.per_wave);
        // this wavefront's first pair goes to its private buffer while the workgroup's tables are still on their way
        int e = wv * gridDim.x + blockIdx.x;
        int entry = 0, clip = 0, fa = 0;
        bool has_b = false, nza = false, nzb = false;
        auto fetch = [&]() {
            entry = rl.list[e];
            const int pair = entry >> 2;
            clip = pair / pairs_per_clip;
            fa = 2 * (pair - clip * pairs_per_clip);
            has_b = fa + 1 < p.num_frames;
#ifdef KWS_X_REFINE_SINGLE
            has_b = false;  // timing experiment: wrong rows for frame b
#endif
            const T* __restrict__ x = wav + (size_t)clip * p.n_samples;
            const long sa = (long)fa * p.frame_step;
            f64_load_pair<T, NCT>(p, x, sa, x, sa + p.frame_step, has_b, X, NCT, n_used, lane, nza, nzb);
        };
#ifndef KWS_X_REFINE_LATE_FETCH
        if (e < n) fetch();
#endif
        const F64Wave w = f64_stage_tables<true, true>(p, t, smem64, NCT);
        __syncthreads();
#ifdef KWS_X_REFINE_LATE_FETCH
        if (e < n) fetch();
#endif
        while (e < n) {
            spectrum_pair<true, NCT>(w.X, w.P, w.tb.tw, NCT, 9, n_used, 1.0 / (double)NCT, lane);
            float* oa = out + ((size_t)clip * p.num_frames + fa) * p.numcep;
            f64_tail<NCT / 2 + 1>(p, w.tb, NCT / 2 + 1, w.P, w.L, nza, nzb, has_b, (entry & 1) ? oa : nullptr,
                                  (entry & 2) ? oa + p.numcep : nullptr, nullptr, lane);
            e += gridDim.x * n_waves;
            if (e < n) fetch();
        }
        __syncthreads();  // every wavefront's rows are issued before this workgroup reports
    }
    // No fence anywhere here: on this part an agent-scope release writes the XCD's L2 back (the float32 kernel's 16 MB of
    // rows are still dirty in it) -- measured +12 us per launch.  Nothing inside this launch depends on another
    // workgroup's data; the counters are published to the next kernel by the kernel boundary.
    // Who resets the counters: the last of the workgroups that HAD work (all of them know n, hence how many they are), or
    // workgroup 0 alone when the list is empty.  A workgroup without work does not touch the counter -- with every workgroup
    // adding to one address across eight XCDs the empty launch took 15.6 us, most of it that queue of atomics.  A would-be
    // participant that has not started yet has not added either, so the reset cannot overtake its read of n.
    int participants = n < 1 ? 1 : (n > (int)gridDim.x ? (int)gridDim.x : n);
    if ((int)blockIdx.x >= participants) return;
    if (tid == 0 && atomicAdd(&rl.ctr[6], 1) == participants - 1) {
        const int rows = rl.ctr[1];  // flagged frames, counted by the float32 kernel next to its list entries
        const unsigned long long total = ((unsigned long long)(unsigned)rl.ctr[3] << 32 | (unsigned)rl.ctr[2]) + (unsigned long long)rows;
        rl.ctr[2] = (int)(unsigned)total;
        rl.ctr[3] = (int)(unsigned)(total >> 32);
        rl.ctr[4] = rows;
        rl.ctr[6] = 0;
        rl.ctr[1] = 0;
        rl.ctr[0] = 0;
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

// Refinement launch: a fixed grid of one eight-wavefront workgroup per CU (the kernel's ~200 registers allow two
// wavefronts per SIMD).  Workgroups cost time even when they leave at once: with four-wavefront workgroups, 256 / 512 /
// 768 of them took 14.9 / 18.3 / 22.8 us on a batch with 583 listed frames (same box, same call).
#ifndef KWS_X_REFINE_GRID
#define KWS_X_REFINE_GRID 256
#endif
constexpr int REFINE_GRID = KWS_X_REFINE_GRID;
template <typename T>
hipError_t launch_mfcc_refine_t(hipStream_t s, const FrontendParams& p, const FrontendTables& t, const T* d_wav, float* d_out,
                                const RefineList& rl, int B) {
    const size_t lds = f64_layout(512, true, true, p.nfilt, p.numcep, REFINE_WAVES).total;
    auto kernel = kws_mfcc_refine_kernel<T>;
    hipError_t e = raise_lds_limit(kernel, lds);
    if (e != hipSuccess) return e;
    // workgroups cost time even when they leave at once (~3.5 us per 256 of them): a small batch, whose list is short in
    // proportion, gets a smaller grid -- one workgroup per 16 clips, between 32 and REFINE_GRID
    int grid = (B + 15) / 16;
    grid = grid < 32 ? 32 : (grid > REFINE_GRID ? REFINE_GRID : grid);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(REFINE_WAVES * 64), lds, s, p, t, d_wav, d_out, rl);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_mfcc_refine(hipStream_t s, const FrontendParams& p, const FrontendTables& t, const int16_t* d_wav, float* d_out,
                              const RefineList& rl, int B) {
    return launch_mfcc_refine_t(s, p, t, d_wav, d_out, rl, B);
}
hipError_t launch_mfcc_refine_f32in(hipStream_t s, const FrontendParams& p, const FrontendTables& t, const float* d_wav, float* d_out,
                                    const RefineList& rl, int B) {
    return launch_mfcc_refine_t(s, p, t, d_wav, d_out, rl, B);
}

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
