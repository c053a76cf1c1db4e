// Internal declarations shared by the C-ABI translation unit and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "kws_hip.h"

namespace kws {

// ------------------------------------------------------------------------------------------------
// Front end (MFCC) -- a workgroup of MFCC_WAVES wavefronts handles MFCC_FRAMES_PER_WG frames; each wavefront
// transforms frame pairs (2 real frames are packed into one 512-point complex FFT).
constexpr int NFFT = 512;
constexpr int NBINS = NFFT / 2 + 1;       // 257
constexpr int MEL_CHUNK = 8;              // bins per lane in the sparse mel stage
constexpr int MEL_STRIDE = 10;            // power-buffer slots (float2) from one lane's chunk to the next: 80 bytes, so the sixteen
                                          // lanes of a ds_read_b128 group start on sixteen different 16-byte bank groups (with 64
                                          // bytes every fourth lane collided: 48 of the mel stage's LDS cycles per frame pair)
#ifndef KWS_MFCC_WAVES
#define KWS_MFCC_WAVES 4
#endif
#ifndef KWS_MFCC_PAIRS_PER_WAVE
#define KWS_MFCC_PAIRS_PER_WAVE 3
#endif
constexpr int MFCC_WAVES = KWS_MFCC_WAVES;             // wavefronts per workgroup
constexpr int MFCC_THREADS = MFCC_WAVES * 64;
constexpr int MFCC_FRAMES_PER_WG = 2 * KWS_MFCC_PAIRS_PER_WAVE * MFCC_WAVES;
constexpr int MAX_NFILT = 64;
constexpr int MAX_NUMCEP = 32;

struct FrontendParams {
    int n_samples;
    int frame_len;
    int frame_step;
    int num_frames;
    int nfilt;
    int numcep;
    int append_energy;
    float preemph;
    int chunk_samples;     // samples staged per workgroup = (MFCC_FRAMES_PER_WG-1)*frame_step + frame_len
    int vec_ok;            // 1 -> every workgroup's first sample is 16-byte aligned (vector PCM loads legal)
    int nfft;              // transform length (the float32 kernel is built for 512; the float64 kernel takes any)
    int log2_nfft;         // log2(nfft) when nfft is a power of two, else 0 (float64 kernel: FFT vs direct DFT)
    float refine_span;     // float32 kernels: a frame whose weakest mel band lies more than this (log power) below its largest bin is redone in float64 (0: never)
};

// Worklist of the selective float64 refinement: the float32 MFCC kernel appends, per frame pair with a
// flagged frame, (global pair index << 2 | mask of flagged frames), global pair index = clip * ceil(num_frames / 2) +
// frame / 2; the refinement kernel consumes the list and clears the counters.
// ctr: int[8], 8-byte aligned = {entries listed by the running call, frames flagged by it (the two are bumped by ONE 64-bit
// atomic), rows rewritten in total (low, high 32 bits), rows rewritten by the last completed call, frames redone in float64
// by streaming pushes, finished refinement workgroups, 0}.
struct RefineList {
    int* ctr;
    int* list;
    int cap;       // entries list can hold (frame pairs of the largest batch reserved); 0 with ctr == nullptr: flagging off
    int clip0;     // index of the launch's first clip inside the call's batch (batches beyond 65535 clips are split)
};

// Device tables of the front end (all float32 unless noted), built on the host in double.
struct FrontendTables {
    const float2* twiddle;     // [512]      (cos, -sin)(2*pi*k/512)
    const int* mel_k0;         // [64]       first bin of lane's chunk (or NBINS-1 for idle lanes)
    const float* mel_rw;       // [8][64]    rising-edge weights of the chunk's bins (0 beyond its length)
    const float* mel_fw;       // [8][64]    falling-edge weights
    const uint32_t* mel_gather;// [64]       per filter: r0 | nr<<8 | f0<<16 | nf<<24  (chunk ranges)
    const int* mel_slot;       // [256]      power-buffer slot of bin k = MEL_STRIDE*chunk(k) + (k - first bin of the chunk)
    const int* mel_seg;        // [64]       per chunk: bit d (0..2) = chunk + 2^d is in the same segment; bit 6 = no segment straddles a 16-lane row; bit 7 = deep
    const float* dct;          // [numcep][nfilt]  DCT-II ortho x lifter
    const float* dct_pad;      // [numcep][nfp]    the same, rows zero-padded to nfp = (nfilt+3)&~3 floats (the LDS image, copied as float4)
    // float64 kernel (kws_mfcc_f64.hip)
    const double* tw64;        // [nfft][2]  (cos, -sin)(2*pi*k/nfft)
    const int* mel_edges;      // [nfilt+2]  psf's bin edges
    const double* mel_w64;     // [2][nfft/2+1]  per bin: its weight on the rising edge of its filter / on the falling edge of the one below
    const double* dct64;       // [numcep][nfilt]  DCT-II ortho x lifter
};

// Host-side construction of the sparse mel decomposition (also used by the host-only ABI helpers).
struct MelHost {
    std::vector<int> edges;            // nfilt+2
    std::vector<int> k0;               // 64
    std::vector<float> rw, fw;         // 8*64 each, [i][lane]
    std::vector<uint32_t> gather;      // 64
    std::vector<int> slot;             // nfft/2 (bin 256 belongs to no filter)
    std::vector<int> seg;              // 64
    int n_chunks = 0;
};
bool build_mel_host(int nfilt, int nfft, int sample_rate, MelHost& out, std::string& err);
void build_dct_lifter_host(int nfilt, int numcep, int ceplifter, std::vector<float>& out);
void build_twiddle_host(std::vector<float2>& out);

hipError_t launch_mfcc(hipStream_t s, const FrontendParams& p, const FrontendTables& t,
                       const int16_t* d_wav, int B, float* d_out);
hipError_t launch_mfcc_f32(hipStream_t s, const FrontendParams& p, const FrontendTables& t,
                           const float* d_wav, int B, float* d_out);
size_t mfcc_lds_bytes(const FrontendParams& p);
// The float32 kernels with flagging: frames over p.refine_span are appended to rl (rl.ctr == nullptr: no flagging).
hipError_t launch_mfcc_flag(hipStream_t s, const FrontendParams& p, const FrontendTables& t, const int16_t* d_wav, int B, float* d_out,
                            const RefineList& rl);
hipError_t launch_mfcc_f32_flag(hipStream_t s, const FrontendParams& p, const FrontendTables& t, const float* d_wav, int B,
                                float* d_out, const RefineList& rl);
// Float64 recomputation of the listed frames, in place in d_out (same d_wav / d_out as the float32 launch before it).
hipError_t launch_mfcc_refine(hipStream_t s, const FrontendParams& p, const FrontendTables& t, const int16_t* d_wav, float* d_out,
                              const RefineList& rl, int B);
hipError_t launch_mfcc_refine_f32in(hipStream_t s, const FrontendParams& p, const FrontendTables& t, const float* d_wav, float* d_out,
                                    const RefineList& rl, int B);
// The float64 front end: any nfft (power of two up to 4096: FFT; otherwise up to 2048: direct DFT), any frame length.
hipError_t launch_mfcc_f64(hipStream_t s, const FrontendParams& p, const FrontendTables& t, const int16_t* d_wav, int B, float* d_out);
hipError_t launch_mfcc_f64_f32in(hipStream_t s, const FrontendParams& p, const FrontendTables& t, const float* d_wav, int B,
                                 float* d_out);
hipError_t launch_spec_f64(hipStream_t s, const double* d_tw64, const float* d_frames, int num_frames, int frame_len, int nfft,
                           int log2n, int power, float* d_spec);
// Streaming: one hop of frame_step new samples per stream -> one new MFCC frame per stream in the feature ring.
// d_hops: int[2] = {hops pushed so far, finished-workgroup counter}; the kernel advances the hop count itself.
hipError_t launch_stream_frame(hipStream_t s, const FrontendParams& p, const FrontendTables& t, const int16_t* d_hop,
                               int n_streams, int16_t* d_pcm_ring, int ring_len, float* d_feat_ring, int* d_hops, int* d_refine_ctr);
// Training-time augmentation on the device (time shift, silence, background mix).
hipError_t launch_augment(hipStream_t s, const int16_t* d_wav, int B, int n, const int32_t* d_shift, const float* d_bg,
                          int bg_len, const int32_t* d_bg_off, const float* d_bg_vol, const uint8_t* d_silence,
                          float* d_out);
hipError_t launch_preemphasis(hipStream_t s, const float* d_in, int n, float coeff, float* d_out);
hipError_t launch_framesig(hipStream_t s, const float* d_in, int n, int frame_len, int frame_step,
                           int num_frames, const float* d_window, float* d_frames);
hipError_t launch_spec512(hipStream_t s, const FrontendTables& t, const float* d_frames, int num_frames,
                          int frame_len, int power, float* d_spec);

// ------------------------------------------------------------------------------------------------
// DS-CNN (kws/libs/models.py:122-183) for the [1,99,10] feature map.
constexpr int CH = 64;
constexpr int IN_T = 99, IN_F = 10;
constexpr int C1_K = 10, C1_H = 47, C1_W = 3;             // conv1 output 64 x 47 x 3
constexpr int N_BLOCKS = 4;
constexpr int MAX_CLASSES = 64;

// Device weights, repacked by the host for the kernel's access patterns.
struct DscnnWeights {
    const float* c1_w;     // [100][64]   conv1 weight transposed: [kh*10+kw][cout]
    const float* c1_b;     // [64]
    const float* dw_w;     // [4][32][24] depthwise 3x3, channel pairs interleaved: (tap t of channel 2p, of 2p + 1) at 2t, t = 0..8; the biases at 18, 19; 4 pad
    const float* pw_w;     // [4][64][64] pointwise transposed: [cin][cout]
    const float* pw_b;     // [4][64]
    const uint32_t* pw_split;  // [4][ct 2][m 4][piece 3][lane 64][4]  pointwise weights as bf16 hi/mid/lo MFMA A operands
    const uint32_t* c1_split;  // [ct 2][kb 7][piece 3][lane 64][4]    conv1 weights, same format (K order: see kws_dscnn.hip)
    const float* fc_w;     // [C][64]
    const float* fc_b;     // [C]
    int num_classes;
    int in_channels;           // 1: the fused kernel computes conv1 itself; > 1: kws_conv1_general_kernel + the PRECONV entry
    const float* c1_general;   // [ci][100][64]  conv1 weights as [ci][tap][cout] (multi-channel models and the composed any-map path)
    const float* raw;          // the state_dict blob as loaded (torch layouts): the composed path's depthwise / pointwise operands
    // f16-pair arithmetic (KWS_PW_PAIR_F16, kws_dscnn.hip): every weight scaled by its layer's power of two 2^k (max |w| 2^k <
    // 2^15) as hi = f16(w 2^k), lo = f16(w 2^k - hi); same fragment orders as the bf16 images with two pieces
    const uint32_t* pw_pair;   // [4][ct 2][m 4][piece 2][lane 64][4]
    const uint32_t* c1_pair;   // [ct 2][kb 7][piece 2][lane 64][4]
    int k_c1, k_pw[4];         // the layers' weight-scale exponents
    // bounds the per-clip activation scales are derived from: |conv1 out| <= c1_abs max|x| + c1_bmax; per block
    // |depthwise out| <= dw_abs max|in| + dw_bmax, |pointwise out| <= pw_abs max|depthwise out| + pw_bmax (row sums of |w|,
    // maximised over output channels, rounded up)
    float c1_abs, c1_bmax, dw_abs[4], dw_bmax[4], pw_abs[4], pw_bmax[4];
};

hipError_t dscnn_init_device();
// The streaming push as ONE launch: every stream's workgroup of the DS-CNN kernel computes that stream's new MFCC frame in
// its prologue (kws_mfcc_dev.h: stream_frame_wave), appends the hop to the PCM ring, and the last workgroup to finish
// advances the hop counter.  d_feat_ring: [n_streams][num_frames][numcep]; d_hops: {pushes so far, finished workgroups}.
struct StreamPush {
    FrontendParams p;
    FrontendTables t;
    const int16_t* hop;      // [n_streams][frame_step]
    int16_t* pcm_ring;       // [n_streams][ring_len]
    int ring_len;
    int* hops;
    int* refine_ctr;         // counters of the selective refinement (RefineList::ctr; [5] counts frames redone by pushes) or NULL
    int frames_lag;          // two-launch route only: hops a frame spans, ceil(frame_len / frame_step) (the fused push derives it from p)
    int cluster;             // workgroups per stream of the fused push (time tiles; 1 = one workgroup owns the stream's network)
    float* cl_part;          // [n_streams][cluster][64] pooled partial sums of the tiles
    int* cl_count;           // [n_streams] tiles that have delivered theirs (the last one runs fc + argmax and clears it)
    float* h_logits;         // zero-copy delivery (kws_stream_host_results): pinned host [n_streams][C], or NULL
    int32_t* h_label;        // pinned host [n_streams], or NULL
    int* h_flag;             // pinned host: the push count whose results are complete in h_logits / h_label
};
hipError_t launch_dscnn_stream(hipStream_t s, const DscnnWeights& w, const StreamPush& sp, float* d_feat_ring, int n_streams,
                               float* d_logits, int32_t* d_label, bool pair);
hipError_t launch_dscnn(hipStream_t s, const DscnnWeights& w, const float* d_feat, int B, float* d_logits,
                        int32_t* d_label, float* d_act, int mode, unsigned long long* d_stamps = nullptr,
                        const int* d_ring_hops = nullptr, bool preconv = false, int frames_lag = 3);
// conv1 of a model with input_channels > 1: x [B][C_in][99][10] -> relu(conv1) [B][64][141]; d_wt = weights as [ci][tap][co]
hipError_t launch_conv1_general(hipStream_t s, const float* d_x, int B, int C_in, const float* d_wt, const float* d_bias, float* d_out);

// ------------------------------------------------------------------------------------------------
// cnn-trad-fpool3 (build-defined model-zoo member, kws_cnntrad.hip)
struct CnnTradWeights {
    const uint32_t* c1_split;  // [kb 10][ct 2][piece 3][lane 64][4]    conv1 20x8 as bf16 hi/mid/lo MFMA A operands
    const float* c1_b;         // [64]
    const uint32_t* c2_split;  // [kk 40][cb 4][ct 2][piece 3][lane 64][4]  conv2 10x4x64
    const float* c2_b;         // [64]
    const uint32_t* lin_split; // [kb 1188][piece 3][lane 64][4]  first dense layer as bf16 hi/mid/lo MFMA B operands
    const float* lin_b;        // [32]
    const float* dnn_w;        // [128][32]
    const float* dnn_b;        // [128]
    const float* fc_w;         // [C][128]
    const float* fc_b;         // [C]
    int num_classes;
    // f16-pair arithmetic (kws_cnntrad.hip): every weight scaled by the layer's power of two sw (max |w| sw < 2^15) and written
    // as hi = f16(w sw), lo' = f16((w sw - hi) 2^11); same fragment orders as above with two pieces
    const uint32_t* c1_h2;     // [kb 10][ct 2][piece 2][lane 64][4]
    const uint32_t* c2_h2;     // [kk 40][cb 4][ct 2][piece 2][lane 64][4]
    const uint32_t* lin_h2;    // [kb 1188][piece 2][lane 64][4]
    float inv_sw1, inv_sw2, inv_swl;   // 1 / sw per layer
    float w1_abs, b1_max;      // max over output channels of sum |w1[c]|, max |b1|: |conv1 out| <= w1_abs * max|x| + b1_max
    float w2_abs, b2_max;      // the same for conv2
};
hipError_t cnntrad_init_device();
hipError_t launch_cnntrad_conv(hipStream_t s, const CnnTradWeights& w, const float* d_feat, int B, float* d_conv_ws, bool f16_pair);
hipError_t launch_cnntrad_dense(hipStream_t s, const CnnTradWeights& w, const float* d_conv_ws, int B, float* d_logits,
                                int32_t* d_label, bool f16_pair);

// Standalone depthwise-separable block on an arbitrary [B, C_in, H, W] map (kws_dsblock.hip); d_ws: B*C_in*Ho*Wo floats.
hipError_t launch_dsblock(hipStream_t s, const float* d_x, int B, int C_in, int H, int W, const float* d_dw_w, const float* d_dw_b,
                          const float* d_pw_w, const float* d_pw_b, int C_out, int k, int stride, int pad, float* d_ws,
                          float* d_out);

// DS-CNN on a feature map other than 99 x 10 (composed, HBM-resident): conv1 for any [B][C_in][T][F] (d_wt: [ci][tap][co]) and
// global average pool + fc + argmax over [B][64][HW].
hipError_t launch_conv1_any(hipStream_t s, const float* d_x, int B, int C_in, int T, int F, const float* d_wt, const float* d_bias,
                            float* d_out);
hipError_t launch_pool_fc(hipStream_t s, const float* d_x, int B, int HW, const float* d_fc_w, const float* d_fc_b, int C,
                          float* d_logits, int32_t* d_label);

hipError_t launch_softmax(hipStream_t s, const float* d_logits, int B, int C, float* d_prob);
hipError_t launch_smooth_posteriors(hipStream_t s, const float* d_logits, int S, int C, int window, float* d_ring,
                                    float* d_sum, int* d_count, float* d_smoothed, int32_t* d_label);
hipError_t launch_stream_vad(hipStream_t s, const float* d_feat_ring, const int* d_hops, int n_streams, int num_frames, int numcep,
                             int frames_lag, float threshold, int on_window, int off_window, unsigned char* d_flags, int* d_cursor_trig,
                             int32_t* d_state);

extern const char* const kKernelNames[KWS_K_COUNT];

}  // namespace kws
