// Batched MFCC front end for gfx950 (MI355X), replacing per-clip psf.mfcc calls
// (reference call site kws/libs/audio_processor.py:270-278; stages a1-a8 of SURVEY.md section 8).
//
// Work decomposition
//   grid = (ceil(num_frames / 24), B); a workgroup is MFCC_WAVES (4) wavefronts and owns 24 frames
//   (12 frame pairs, three per wavefront; shape chosen by A/B timing on MI355X).  Together
//   the waves stage the PCM span of those frames (4080 samples for 400/160) from HBM with 16-byte loads,
//   convert to float32, apply pre-emphasis once
//   and keep the span in LDS, next to the tables every wave needs (DCT x lifter, sparse-mel weights,
//   second-pass twiddles).  After one barrier the waves never synchronise again: each has a private
//   5 KiB scratch and LDS instructions of one wavefront execute in order.
//
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
// (tolerance 1e-4 on MFCC, see tests/test_gpu_parity.py).
#include <cstdlib>

#include "kws_internal.h"
#include "kws_mfcc_dev.h"

namespace kws {
namespace {

// ------------------------------------------------------------------------------------------------
// TAIL6: frame_len in (384, 448] (the reference's 400): sample blocks n1 < 6 lie wholly inside the frame, block 6 is cut
// by a lane bound and block 7 is zero -- no loads or selects for it, and the zeros fold through the first butterflies.
template <typename T, bool TAIL6>
__device__ __forceinline__ void mfcc_body(const FrontendParams& p, const FrontendTables& t, const T* __restrict__ wav,
                                          float* __restrict__ out, const RefineList& rl) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // workgroup-shared part (every offset a multiple of 16 bytes)
    const int nfp = (p.nfilt + 3) & ~3;                        // DCT rows padded to float4
    float* dctb = reinterpret_cast<float*>(smem);              // [numcep][nfp]
    cf* tw2 = reinterpret_cast<cf*>(dctb + ((p.numcep * nfp + 3) & ~3));  // [8][8]
    float* ybuf = reinterpret_cast<float*>(tw2 + 64);          // chunk_samples floats (padded to 8)
    unsigned char* scr0 = reinterpret_cast<unsigned char*>(ybuf + ((p.chunk_samples + 7) & ~7));

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int clip = blockIdx.y;
    const int f0 = blockIdx.x * MFCC_FRAMES_PER_WG;
    const T* __restrict__ x = wav + (size_t)clip * p.n_samples;

    // ---- stage PCM span -> float32, pre-emphasised, in LDS; shared tables -------------------------
    const int s0 = f0 * p.frame_step;
    const float c = p.preemph;
    // the clip's last workgroup owns fewer frames (3 of 24 for 99 frames): it stages only the span they cover
    const int my_frames = min(MFCC_FRAMES_PER_WG, p.num_frames - f0);
    const int my_samples = (my_frames - 1) * p.frame_step + p.frame_len;
    for (int g = tid; g * 8 < my_samples; g += MFCC_THREADS) {
        const int n = s0 + g * 8;
        float y[8];
        if (sizeof(T) == 2 && p.vec_ok && n + 8 <= p.n_samples) {
            // every 8-sample group is 16-byte aligned here
            const uint4 raw = *reinterpret_cast<const uint4*>(x + n);
            float prev = (n > 0) ? to_unit(x[n - 1]) : 0.f;
            const uint32_t wd[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int16_t s = (int16_t)((wd[i >> 1] >> (16 * (i & 1))) & 0xffffu);
                const float cur = to_unit(s);
                y[i] = (n + i > 0) ? __fsub_rn(cur, __fmul_rn(c, prev)) : cur;
                prev = cur;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = n + i;
                float v = 0.f;
                if (m < p.n_samples) {
                    const float cur = to_unit(x[m]);
                    v = (m > 0) ? __fsub_rn(cur, __fmul_rn(c, to_unit(x[m - 1]))) : cur;
                }
                y[i] = v;
            }
        }
        float4* dst = reinterpret_cast<float4*>(ybuf + g * 8);
        dst[0] = make_float4(y[0], y[1], y[2], y[3]);
        dst[1] = make_float4(y[4], y[5], y[6], y[7]);
    }
    for (int i = tid; i < (p.numcep * nfp + 3) / 4; i += MFCC_THREADS)  // the host keeps the padded LDS image
        reinterpret_cast<float4*>(dctb)[i] = reinterpret_cast<const float4*>(t.dct_pad)[i];
    fill_tw2(t.twiddle, tw2, tid);

    // ---- per-lane constants -----------------------------------------------------------------------
    cf t1[8];
    load_twiddles(t.twiddle, lane, t1);
    MelLane ml;
    load_mel_lane(t, lane, ml);
    unsigned char* scr = scr0 + wv * SCR_BYTES;
    zero_scratch(scr, lane);
    __syncthreads();  // the only workgroup barrier: staged samples and tables are visible to all waves

    const PairScratch sc = {reinterpret_cast<cf*>(scr + SCR_XBUF), reinterpret_cast<float2*>(scr + SCR_PBUF),
                            reinterpret_cast<float*>(scr + SCR_LBUF),
                            dctb, tw2, nfp};

    for (int pr = wv; pr < MFCC_FRAMES_PER_WG / 2; pr += MFCC_WAVES) {
        const int fa = f0 + 2 * pr;
        if (fa >= p.num_frames) break;  // wave-uniform
        const bool has_b = (fa + 1) < p.num_frames;
        const float* ya = ybuf + (2 * pr) * p.frame_step;
        const float* yb = ya + p.frame_step;

        // The loads are unconditional and the frame bound is applied by a select: a read past the frame (at most
        // 511 floats past the staged span) lands in the wavefronts' scratch behind ybuf -- allocated LDS, any bits --
        // and is discarded; conditional loads compile to one branch per load.  The all-zero test ORs the bit
        // patterns (sign bit dropped: -0.0 counts as zero, like the float comparison).
        cf v[8];
        uint32_t ora = 0u, orb = 0u;
        if constexpr (TAIL6) {
#pragma unroll
            for (int n1 = 0; n1 < 7; ++n1) {
                const int i = 64 * n1 + lane;
                float a = ya[i], b = yb[i];
                if (n1 == 6) {
                    const bool in = i < p.frame_len;
                    a = in ? a : 0.f, b = in ? b : 0.f;
                }
                v[n1] = cf{a, b};
                ora |= __builtin_bit_cast(uint32_t, a);
                orb |= __builtin_bit_cast(uint32_t, b);
            }
            v[7] = cf{0.f, 0.f};
            if (!has_b) {  // wave-uniform, the clip's last pair only: frame b is past the clip
#pragma unroll
                for (int n1 = 0; n1 < 7; ++n1) v[n1].y = 0.f;
                orb = 0u;
            }
        } else {
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) {
                const int i = 64 * n1 + lane;
                const bool in = i < p.frame_len;
                const float la = ya[i], lb = yb[i];  // loaded whether or not they are used (see above)
                const float a = in ? la : 0.f, b = (in && has_b) ? lb : 0.f;
                v[n1] = cf{a, b};
                ora |= __builtin_bit_cast(uint32_t, a);  // (bit_cast of a vector ELEMENT reads element 0 with this clang:
                orb |= __builtin_bit_cast(uint32_t, b);  //  cast the scalars)
            }
        }
        const bool nza = __any((ora << 1) != 0u);
        const bool nzb = __any((orb << 1) != 0u);
        const uint32_t flags = mfcc_pair(v, nza, nzb, has_b, p, sc, t1, ml, lane,
                                         out + ((size_t)clip * p.num_frames + fa) * p.numcep,
                                         out + ((size_t)clip * p.num_frames + fa + 1) * p.numcep);
        // a frame the float32 arithmetic cannot hold to 1e-4 (rare; wave-uniform): the pair goes onto the refinement
        // kernel's worklist with the mask of its flagged frames
        // (one 64-bit atomic: entries in the low word -- its old value is the slot --, flagged frames in the high word)
        if (flags && rl.ctr && lane == 0) {
            const unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long*>(rl.ctr), 1ull | ((unsigned long long)__builtin_popcount(flags) << 32));
            const int at = (int)(unsigned)old;
            if (at < rl.cap) rl.list[at] = (int)((((unsigned)(rl.clip0 + clip) * (unsigned)((p.num_frames + 1) / 2) + (unsigned)(fa >> 1)) << 2) | flags);
        }
        wave_lds_order();
    }
}

#ifndef KWS_MFCC_EU
#define KWS_MFCC_EU 4
#endif
// amdgpu_waves_per_eu(4, 4): 128 registers, so that the four 4-wave workgroups the LDS admits per CU (16
// wavefronts, 4 per SIMD) all become resident; the kernel is latency-bound on LDS round trips.
// kws_mfcc_i16_tile_kernel / kws_mfcc_f32_kernel: frame lengths in (384, 448] (the reference's 400 samples), see TAIL6; the
// *_any_kernel pair takes every other frame length up to 512 (one kernel with both bodies spills registers).  int16 input
// at a geometry the wavefront-resident kernel below covers goes there instead (kws_mfcc_i16_kernel, the product path).
__global__ __launch_bounds__(MFCC_THREADS) __attribute__((amdgpu_waves_per_eu(KWS_MFCC_EU, KWS_MFCC_EU))) void kws_mfcc_i16_tile_kernel(FrontendParams p, FrontendTables t,
                                                                    const int16_t* __restrict__ wav,
                                                                    float* __restrict__ out, RefineList rl) {
    mfcc_body<int16_t, true>(p, t, wav, out, rl);
}
__global__ __launch_bounds__(MFCC_THREADS) __attribute__((amdgpu_waves_per_eu(KWS_MFCC_EU, KWS_MFCC_EU))) void kws_mfcc_f32_kernel(FrontendParams p, FrontendTables t,
                                                                    const float* __restrict__ wav,
                                                                    float* __restrict__ out, RefineList rl) {
    mfcc_body<float, true>(p, t, wav, out, rl);
}
__global__ __launch_bounds__(MFCC_THREADS) __attribute__((amdgpu_waves_per_eu(KWS_MFCC_EU, KWS_MFCC_EU))) void kws_mfcc_i16_any_kernel(FrontendParams p, FrontendTables t,
                                                                    const int16_t* __restrict__ wav,
                                                                    float* __restrict__ out, RefineList rl) {
    mfcc_body<int16_t, false>(p, t, wav, out, rl);
}
__global__ __launch_bounds__(MFCC_THREADS) __attribute__((amdgpu_waves_per_eu(KWS_MFCC_EU, KWS_MFCC_EU))) void kws_mfcc_f32_any_kernel(FrontendParams p, FrontendTables t,
                                                                    const float* __restrict__ wav,
                                                                    float* __restrict__ out, RefineList rl) {
    mfcc_body<float, false>(p, t, wav, out, rl);
}

// ------------------------------------------------------------------------------------------------
// The product kernel for int16 PCM at the reference geometry (round 3): WAVEFRONT-RESIDENT runs of a clip.
//
// The tile kernel above pays, for every 12 frame pairs, a workgroup launch, ~40 table loads per lane, a scratch clear, the
// global-memory latency of its PCM span and a workgroup barrier, and its four wavefronts march in step.  Here a wavefront
// owns a run of a clip's chunks (five by default: 20 frames) for its whole lifetime: tables and per-lane constants are
// loaded once, then it walks its run in chunks of four frames (two packed pairs).  A chunk's PCM span (3 steps + one frame
// = 880 samples) arrives as two 16-byte loads per lane, is converted and pre-emphasised in registers (the sample before a
// lane's vector comes from its neighbour by DPP) and lands in the wavefront's own 4 KB of LDS.  No wavefront ever waits for
// another: the one __syncthreads publishes the shared DCT and twiddle tables at the start.  Consecutive spans overlap by
// 240 samples; the re-read is an L2 hit issued by the same wavefront a few microseconds earlier.  Same arithmetic, same
// bits as the tile kernel (PCM scaling and pre-emphasis are the same two float32 roundings; mfcc_pair is shared).
// Measured on 4096 clips, same box and call: tile kernel 0.208 ms, this kernel 0.183 ms.  (Issuing the next chunk's loads
// before this chunk's transforms -- a register prefetch -- was measured and dropped: the eight vector registers it keeps
// alive spill under the 128-register cap, 0.201 ms; the other three wavefronts of the SIMD cover the latency.)
constexpr int WR_PAIRS = 2, WR_FRAMES = 2 * WR_PAIRS;   // frames per chunk
constexpr int WR_SPAN = 1024;                            // floats of LDS per wavefront for the chunk's span (two uint4 of PCM per lane)
constexpr int WR_WAVE_BYTES = WR_SPAN * 4 + SCR_BYTES;

__host__ __device__ inline size_t mfcc_wr_lds_bytes(const FrontendParams& p) {
    const int nfp = (p.nfilt + 3) & ~3;
    return sizeof(float) * (size_t)(((p.numcep * nfp + 3) & ~3) + 2 * 64) + (size_t)MFCC_WAVES * WR_WAVE_BYTES;
}

__global__ __launch_bounds__(MFCC_THREADS) __attribute__((amdgpu_waves_per_eu(KWS_MFCC_EU, KWS_MFCC_EU))) void kws_mfcc_i16_kernel(
    FrontendParams p, FrontendTables t, const int16_t* __restrict__ wav, float* __restrict__ out, RefineList rl, int B, int chunks_per_wave) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nfp = (p.nfilt + 3) & ~3;
    float* dctb = reinterpret_cast<float*>(smem);
    cf* tw2 = reinterpret_cast<cf*>(dctb + ((p.numcep * nfp + 3) & ~3));
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    unsigned char* mine = reinterpret_cast<unsigned char*>(tw2 + 64) + wv * WR_WAVE_BYTES;
    float* ybuf = reinterpret_cast<float*>(mine);
    unsigned char* scr = mine + WR_SPAN * 4;

    for (int i = tid; i < (p.numcep * nfp + 3) / 4; i += MFCC_THREADS)
        reinterpret_cast<float4*>(dctb)[i] = reinterpret_cast<const float4*>(t.dct_pad)[i];
    fill_tw2(t.twiddle, tw2, tid);
    cf t1[8];
    load_twiddles(t.twiddle, lane, t1);
    MelLane ml;
    load_mel_lane(t, lane, ml);
    zero_scratch(scr, lane);

    // this wavefront's share: chunks [c, c_end) of clip `clip`
    const int n_chunks = (p.num_frames + WR_FRAMES - 1) / WR_FRAMES;
    const int waves_per_clip = (n_chunks + chunks_per_wave - 1) / chunks_per_wave;
    const int gw = blockIdx.x * MFCC_WAVES + wv;
    const int clip = gw / waves_per_clip, part = gw - clip * waves_per_clip;
    int c = part * chunks_per_wave;
    const int c_end = min(n_chunks, c + chunks_per_wave);
    const bool active = clip < B && c < c_end;  // wave-uniform
    const int16_t* __restrict__ x = wav + (size_t)(active ? clip : 0) * p.n_samples;
    const int chunk_step = WR_FRAMES * p.frame_step;  // samples from one chunk's span to the next (a multiple of 8: checked by the launcher)

    // two 16-byte vectors of PCM per lane cover the span; a vector is inside the clip or past its end (n_samples % 8 == 0)
    uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0;
    int16_t before = 0;
    auto fetch = [&](int cc) {
        const int s0 = cc * chunk_step;
        const int n0 = s0 + 8 * lane, n1 = n0 + 512;
        v0 = n0 + 8 <= p.n_samples ? *reinterpret_cast<const uint4*>(x + n0) : make_uint4(0, 0, 0, 0);
        v1 = n1 + 8 <= p.n_samples ? *reinterpret_cast<const uint4*>(x + n1) : make_uint4(0, 0, 0, 0);
        before = s0 > 0 ? x[s0 - 1] : (int16_t)0;
    };
#ifdef KWS_X_MFCC_PREFETCH
    if (active) fetch(c);
#endif
    __syncthreads();  // the only workgroup barrier: the shared tables are in place
    if (!active) return;

    const PairScratch sc = {reinterpret_cast<cf*>(scr + SCR_XBUF), reinterpret_cast<float2*>(scr + SCR_PBUF),
                            reinterpret_cast<float*>(scr + SCR_LBUF), dctb, tw2, nfp};
    const float pre = p.preemph;
    // The wavefront's flagged pairs go to the worklist together: two flag bits per pair collect in one scalar mask (pair k of the
    // wavefront at bits 2k, 2k+1; a register per entry would spill under the 128-register cap), and when the wavefront is done --
    // or 16 pairs are in -- lane k rebuilds pair k's entry from the mask, ONE 64-bit atomic reserves the slots (entries in the low
    // word: its old value is the first slot; flagged frames in the high word) and the lanes store side by side.  An atomic per
    // flagged pair queued up at the counter's L2 line when flags are dense (speech-like clips, 5 % of the frames: 0.194 -> 0.262 ms).
    uint32_t fmask = 0u;
    int pair0 = 2 * c;  // pair index (within the clip) of mask position 0
#define KWS_FLUSH_FLAGS()                                                                                                          \
    if (fmask && rl.ctr) {                                                                                                          \
        const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); /* (recomputed: the loop's copy would have to be kept in scratch) */ \
        const uint32_t fk = lane < 16 ? (fmask >> (2 * lane)) & 3u : 0u;                                                            \
        const uint32_t has = (uint32_t)__ballot(fk != 0u);                                                                          \
        unsigned long long old = 0ull;                                                                                              \
        if (lane == 0)                                                                                                              \
            old = atomicAdd(reinterpret_cast<unsigned long long*>(rl.ctr),                                                          \
                            (unsigned long long)__builtin_popcount(has) | ((unsigned long long)__builtin_popcount(fmask) << 32));   \
        const int at = __builtin_amdgcn_readfirstlane((int)(unsigned)old) + __builtin_popcount(has & ((1u << (lane & 31)) - 1u));   \
        if (fk != 0u && at < rl.cap)                                                                                                \
            rl.list[at] = (int)((((unsigned)(rl.clip0 + clip) * (unsigned)((p.num_frames + 1) / 2) + (unsigned)(pair0 + lane)) << 2) | fk); \
    }
    for (; c < c_end; ++c) {
#ifndef KWS_X_MFCC_PREFETCH
        fetch(c);
#endif
        // ---- registers -> float32, pre-emphasised, into this wavefront's span buffer -----------------------
        {
            const int s0 = c * chunk_step;
            // the sample before a lane's vector: the last half-word of the neighbouring lane's vector (wave_shr:1), of lane 63's
            // first vector for lane 0's second, of the clip for lane 0's first
            const uint32_t w0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v0.w, 0x138, 0xf, 0xf, false);
            const uint32_t w1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v1.w, 0x138, 0xf, 0xf, false);
            const uint32_t last0 = (uint32_t)__builtin_amdgcn_readlane((int)v0.w, 63);
            float prev0 = lane == 0 ? to_unit(before) : to_unit((int16_t)(w0 >> 16));
            float prev1 = lane == 0 ? to_unit((int16_t)(last0 >> 16)) : to_unit((int16_t)(w1 >> 16));
            auto convert = [&](const uint4& raw, float prev, int n, float* dst) {
                const uint32_t wd[4] = {raw.x, raw.y, raw.z, raw.w};
                float y[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float cur = to_unit((int16_t)((wd[i >> 1] >> (16 * (i & 1))) & 0xffffu));
                    y[i] = __fsub_rn(cur, __fmul_rn(pre, prev));
                    prev = cur;
                }
                if (n == 0) y[0] = to_unit((int16_t)(wd[0] & 0xffffu));  // the clip's first sample is not pre-emphasised
                if (n + 8 > p.n_samples) {                                 // past the clip: psf pads the PRE-EMPHASISED signal with zeros
#pragma unroll
                    for (int i = 0; i < 8; ++i) y[i] = 0.f;
                }
                reinterpret_cast<float4*>(dst)[0] = make_float4(y[0], y[1], y[2], y[3]);
                reinterpret_cast<float4*>(dst)[1] = make_float4(y[4], y[5], y[6], y[7]);
            };
            convert(v0, prev0, s0 + 8 * lane, ybuf + 8 * lane);
            convert(v1, prev1, s0 + 512 + 8 * lane, ybuf + 512 + 8 * lane);
        }
        wave_lds_order();
#ifdef KWS_X_MFCC_PREFETCH  // (experiment, see the header comment: the prefetched vectors spill under the 128-register cap)
        if (c + 1 < c_end) fetch(c + 1);
#endif
#pragma unroll 1
        for (int pr = 0; pr < WR_PAIRS; ++pr) {
            const int fa = WR_FRAMES * c + 2 * pr;
            if (fa >= p.num_frames) break;  // wave-uniform
            const bool has_b = fa + 1 < p.num_frames;
            const float* ya = ybuf + (2 * pr) * p.frame_step;
            const float* yb = ya + p.frame_step;
            cf v[8];
            uint32_t ora = 0u, orb = 0u;
#pragma unroll
            for (int n1 = 0; n1 < 7; ++n1) {  // frame_len in (384, 448]: block 6 is cut by a lane bound, block 7 is zero
                const int i = 64 * n1 + lane;
                float a = ya[i], b = yb[i];
                if (n1 == 6) {
                    const bool in = i < p.frame_len;
                    a = in ? a : 0.f, b = in ? b : 0.f;
                }
                v[n1] = cf{a, b};
                ora |= __builtin_bit_cast(uint32_t, a);
                orb |= __builtin_bit_cast(uint32_t, b);
            }
            v[7] = cf{0.f, 0.f};
            if (!has_b) {  // the clip's last pair only: frame b is past the clip
#pragma unroll
                for (int n1 = 0; n1 < 7; ++n1) v[n1].y = 0.f;
                orb = 0u;
            }
            const bool nza = __any((ora << 1) != 0u);
            const bool nzb = __any((orb << 1) != 0u);
            const uint32_t flags = mfcc_pair(v, nza, nzb, has_b, p, sc, t1, ml, lane,
                                             out + ((size_t)clip * p.num_frames + fa) * p.numcep,
                                             out + ((size_t)clip * p.num_frames + fa + 1) * p.numcep);
            fmask |= flags << (2 * ((fa >> 1) - pair0));  // wave-uniform
            wave_lds_order();
        }
        if (2 * (c + 1) - pair0 >= 16) {  // the mask is full (more than 8 chunks per wavefront: tuning runs only)
            KWS_FLUSH_FLAGS()
            fmask = 0u;
            pair0 = 2 * (c + 1);
        }
    }
    KWS_FLUSH_FLAGS()
#undef KWS_FLUSH_FLAGS
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
    __shared__ __attribute__((aligned(16))) unsigned char smem[SCR_BYTES + 512];  // scratch + tw2
    cf* xbuf = reinterpret_cast<cf*>(smem + SCR_XBUF);
    float2* pbuf = reinterpret_cast<float2*>(smem + SCR_PBUF);
    cf* tw2 = reinterpret_cast<cf*>(smem + SCR_BYTES);
    const int lane = threadIdx.x;
    const int fa = 2 * blockIdx.x;
    const bool has_b = fa + 1 < num_frames;
    cf t1[8];
    load_twiddles(t.twiddle, lane, t1);
    fill_tw2(t.twiddle, tw2, lane);
    wave_lds_order();
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
    const PairLevel lv = equalise_levels(v);
    fft512(v, xbuf, t1, tw2, lane);
    float ea, eb;
    const int ident[4] = {lane, lane + 64, lane + 128, lane + 192};
    split_power(v, xbuf, pbuf, lane, power, nza, nzb, ident, 256, lv, ea, eb);
    for (int k = lane; k < NBINS; k += 64) {
        const float2 pw = pbuf[k];
        spec[(size_t)fa * NBINS + k] = pw.x;
        if (has_b) spec[(size_t)(fa + 1) * NBINS + k] = pw.y;
    }
}

// ------------------------------------------------------------------------------------------------
// Streaming front end as a kernel of its own (pushes that ask for features only, and models other than the fused
// DS-CNN): one wavefront per pair of streams, see stream_frame_wave.
__global__ __launch_bounds__(64) void kws_stream_frame_kernel(FrontendParams p, FrontendTables t,
                                                              const int16_t* __restrict__ hop, int n_streams,
                                                              int16_t* __restrict__ pcm_ring, int ring_len,
                                                              float* __restrict__ feat_ring,
                                                              int* __restrict__ hops_ptr, int* refine_ctr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const int sa = 2 * blockIdx.x, sb = sa + 1;
    const int hops = *hops_ptr;
    stream_frame_wave(p, t, hop, sa, sb, sb < n_streams, pcm_ring, ring_len, feat_ring, hops, smem, lane, nullptr, refine_ctr);
    // hop counter: every workgroup read hops_ptr[0] at its start (its value fed every address above, so that load is
    // long complete); the workgroup that finishes last advances it (hops_ptr[1] counts finished workgroups).  No fence:
    // the counter orders nothing inside this launch, and the kernel boundary publishes it to the next one.
    if (lane == 0) {
        if (atomicAdd(&hops_ptr[1], 1) == (int)gridDim.x - 1) {
            hops_ptr[1] = 0;
            hops_ptr[0] = hops + 1;
        }
    }
}

// Augmentation of the reference's training transform (kws/libs/audio_processor.py:151-159, 172-233) for a whole
// batch: out[b][i] = (silence_b ? 0 : x_b[i - shift_b] / 32768, zero outside the clip) + vol_b * bg[off_b + i],
// float32 with the same two roundings NumPy makes.
__global__ void kws_augment_i16_kernel(const int16_t* __restrict__ wav, int B, int n, const int32_t* __restrict__ shift,
                                       const float* __restrict__ bg, int bg_len, const int32_t* __restrict__ bg_off,
                                       const float* __restrict__ bg_vol, const uint8_t* __restrict__ silence,
                                       float* __restrict__ out) {
    const int b = blockIdx.y;
    const int sh = shift ? shift[b] : 0;
    const bool sil = silence && silence[b];
    const float vol = (bg && bg_vol) ? bg_vol[b] : 0.f;
    const int off = (bg && bg_off) ? bg_off[b] : 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int j = i - sh;
        float a = (!sil && j >= 0 && j < n) ? to_unit(wav[(size_t)b * n + j]) : 0.f;
        if (bg) {
            const int k = off + i;
            const float g = (k >= 0 && k < bg_len) ? bg[k] : 0.f;
            a = __fadd_rn(a, __fmul_rn(g, vol));
        }
        out[(size_t)b * n + i] = a;
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
size_t mfcc_lds_bytes(const FrontendParams& p) {
    const int nfp = (p.nfilt + 3) & ~3;
    return sizeof(float) * (size_t)(((p.numcep * nfp + 3) & ~3) + 2 * 64 + ((p.chunk_samples + 7) & ~7)) +
           (size_t)MFCC_WAVES * SCR_BYTES;
}

template <typename T, typename K>
static hipError_t launch_mfcc_t(K kernel, hipStream_t s, const FrontendParams& p, const FrontendTables& t, const T* d_wav,
                                int B, float* d_out, RefineList rl) {
    dim3 grid((p.num_frames + MFCC_FRAMES_PER_WG - 1) / MFCC_FRAMES_PER_WG, B);
    const size_t lds = mfcc_lds_bytes(p);
    if (lds > 64 * 1024) {  // geometries (or experiment shapes) beyond the default dynamic-LDS limit opt in per kernel
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    // grid.y is limited to 65535: split very large batches
    for (int b0 = 0; b0 < B; b0 += 65535) {
        const int nb = (B - b0 < 65535) ? (B - b0) : 65535;
        grid.y = nb;
        rl.clip0 = b0;
        hipLaunchKernelGGL(kernel, grid, dim3(MFCC_THREADS), lds, s, p, t, d_wav + (size_t)b0 * p.n_samples,
                           d_out + (size_t)b0 * p.num_frames * p.numcep, rl);
    }
    return hipGetLastError();
}

hipError_t launch_mfcc_flag(hipStream_t s, const FrontendParams& p, const FrontendTables& t, const int16_t* d_wav, int B,
                            float* d_out, const RefineList& rl) {
    const bool tail6 = p.frame_len > 384 && p.frame_len <= 448;
#ifndef KWS_X_MFCC_TILE_KERNEL  // (A/B switch: the round-2 tile kernel for every geometry)
    // the wavefront-resident kernel: 16-byte PCM vectors (aligned clips and chunk starts), a chunk's span within two vectors per lane
    if (tail6 && p.vec_ok && (WR_FRAMES * p.frame_step) % 8 == 0 && (WR_FRAMES - 1) * p.frame_step + 448 <= WR_SPAN) {
        const size_t lds = mfcc_wr_lds_bytes(p);
        const int n_chunks = (p.num_frames + WR_FRAMES - 1) / WR_FRAMES;
        for (int b0 = 0; b0 < B; b0 += (1 << 20)) {  // (grid.x limit: 2^31 workgroups; a million clips per launch keeps indices in 32 bits)
            const int nb = B - b0 < (1 << 20) ? B - b0 : (1 << 20);
            // ~5 chunks (20 frames) per wavefront, split evenly (99 frames = 25 chunks = 5 x 5): measured on 4096 clips, 1 / 5 / 25
            // chunks per wavefront take 0.219 / 0.183 / 0.201 ms -- short-lived wavefronts keep the CUs' phases mixed and the tail
            // short, an uneven split (6 + 6 + 6 + 6 + 1: 0.209 ms) wastes a wavefront's set-up on one chunk.  Small batches spread
            // a clip over more wavefronts, down to one chunk each, so that even one clip uses 25 wavefronts.
            int wpc = (n_chunks + 2) / 5;
            wpc = wpc < 1 ? 1 : wpc;
            int cpw = (n_chunks + wpc - 1) / wpc;
            const int fill = (int)(((long)nb * n_chunks + 4095) / 4096);  // chunks per wavefront that still fill 256 CUs x 16 wavefronts
            cpw = cpw > fill ? (fill < 1 ? 1 : fill) : cpw;
            static const int cpw_env = [] {  // experiment hook, read once (profiles/r03_mfcc_ab.txt: the chunks-per-wavefront scan)
                const char* e = getenv("KWS_X_MFCC_CPW");
                return e ? atoi(e) : 0;
            }();
            if (cpw_env > 0) cpw = cpw_env > 16 ? 16 : cpw_env;
            const int waves_per_clip = (n_chunks + cpw - 1) / cpw;
            const long waves = (long)nb * waves_per_clip;
            RefineList r = rl;
            r.clip0 = b0;
            hipLaunchKernelGGL(kws_mfcc_i16_kernel, dim3((unsigned)((waves + MFCC_WAVES - 1) / MFCC_WAVES)), dim3(MFCC_THREADS), lds, s, p, t,
                               d_wav + (size_t)b0 * p.n_samples, d_out + (size_t)b0 * p.num_frames * p.numcep, r, nb, cpw);
        }
        return hipGetLastError();
    }
#endif
    return launch_mfcc_t(tail6 ? kws_mfcc_i16_tile_kernel : kws_mfcc_i16_any_kernel, s, p, t, d_wav, B, d_out, rl);
}
hipError_t launch_mfcc_f32_flag(hipStream_t s, const FrontendParams& p, const FrontendTables& t, const float* d_wav, int B,
                                float* d_out, const RefineList& rl) {
    const bool tail6 = p.frame_len > 384 && p.frame_len <= 448;
    return launch_mfcc_t(tail6 ? kws_mfcc_f32_kernel : kws_mfcc_f32_any_kernel, s, p, t, d_wav, B, d_out, rl);
}
hipError_t launch_mfcc(hipStream_t s, const FrontendParams& p, const FrontendTables& t, const int16_t* d_wav, int B,
                       float* d_out) {
    return launch_mfcc_flag(s, p, t, d_wav, B, d_out, RefineList{});
}
hipError_t launch_mfcc_f32(hipStream_t s, const FrontendParams& p, const FrontendTables& t, const float* d_wav, int B,
                           float* d_out) {
    return launch_mfcc_f32_flag(s, p, t, d_wav, B, d_out, RefineList{});
}

hipError_t launch_stream_frame(hipStream_t s, const FrontendParams& p, const FrontendTables& t, const int16_t* d_hop,
                               int n_streams, int16_t* d_pcm_ring, int ring_len, float* d_feat_ring, int* d_hops, int* d_refine_ctr) {
    const size_t lds = stream_frame_lds_bytes(p);
    hipLaunchKernelGGL(kws_stream_frame_kernel, dim3((n_streams + 1) / 2), dim3(64), lds, s, p, t, d_hop, n_streams,
                       d_pcm_ring, ring_len, d_feat_ring, d_hops, d_refine_ctr);
    return hipGetLastError();
}

hipError_t launch_augment(hipStream_t s, const int16_t* d_wav, int B, int n, const int32_t* d_shift, const float* d_bg,
                          int bg_len, const int32_t* d_bg_off, const float* d_bg_vol, const uint8_t* d_silence,
                          float* d_out) {
    int bx = (n + 255) / 256;
    if (bx > 64) bx = 64;
    for (int b0 = 0; b0 < B; b0 += 65535) {
        const int nb = (B - b0 < 65535) ? (B - b0) : 65535;
        hipLaunchKernelGGL(kws_augment_i16_kernel, dim3(bx, nb), dim3(256), 0, s, d_wav + (size_t)b0 * n, nb, n,
                           d_shift ? d_shift + b0 : nullptr, d_bg, bg_len, d_bg_off ? d_bg_off + b0 : nullptr,
                           d_bg_vol ? d_bg_vol + b0 : nullptr, d_silence ? d_silence + b0 : nullptr, d_out + (size_t)b0 * n);
    }
    return hipGetLastError();
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
