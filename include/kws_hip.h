/*
 * kws_hip.h -- C ABI of libkws_hip.so, the MI355X (gfx950) keyword-spotting hot path:
 *
 *     int16 PCM [B, n_samples]  ->  MFCC float32 [B,1,frames,numcep]  ->  DS-CNN  ->  logits [B,C], label [B]
 *
 * The reference (z430/keyword-spotting) is pure Python and has no FFI/plugin layer; its boundary for
 * this path is a handful of Python callables.  Each entry point below names the reference interface
 * it replaces (paths relative to the reference repo).  The reference-side binding is a ctypes stub,
 * shown in INTEGRATION.md; the build's own host mirror of the reference classes lives in
 * keyword-spotting_amd/kws/.
 *
 * Conventions
 *   - every function returns KWS_OK (0) or a negative KWS_E* code; no exception crosses the ABI;
 *     kws_last_error() gives the message of the last failure on that context.
 *   - pointers named d_* are DEVICE pointers owned by the caller (e.g. torch-ROCm tensor.data_ptr());
 *     the library never frees or retains them after the call's work has drained (kws_sync).
 *   - all work is enqueued asynchronously on the context's HIP stream.
 *   - a context belongs to one GPU and is not thread-safe; contexts are independent (one per GPU,
 *     no collectives: utterances are independent end to end).
 *   - there is NO CPU fallback: without a usable HIP device kws_create fails with KWS_EHIP.
 */
#ifndef KWS_HIP_H
#define KWS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kws_ctx kws_ctx;

enum {
    KWS_OK = 0,
    KWS_EINVAL = -1,       /* bad argument (null pointer, non-positive size, ...) */
    KWS_ENOMEM = -2,       /* host or device allocation failed */
    KWS_EHIP = -3,         /* a HIP runtime call failed (message in kws_last_error) */
    KWS_ESTATE = -4,       /* front end / model not configured yet */
    KWS_EUNSUPPORTED = -5  /* configuration outside what the gfx950 kernels implement */
};

/* ABI version of this header; kws_abi_version() returns the library's. */
#define KWS_ABI_VERSION 1
int kws_abi_version(void);

/* ---- context ------------------------------------------------------------------------------- */

/* Create a context on GPU `device_id` with its own non-blocking HIP stream and the default front
 * end (kws_set_frontend defaults).  Replaces the implicit device selection of
 * kws/libs/models.py:67 and kws/libs/data_loader.py:63 (`torch.device("cuda" ...)`). */
int kws_create(kws_ctx** out, int device_id);
void kws_destroy(kws_ctx* ctx);

/* external != 0: enqueue on the caller's stream `hip_stream` (a hipStream_t, e.g.
 * torch.cuda.current_stream().cuda_stream; NULL is the device's default stream, which is what torch
 * uses unless told otherwise).  external == 0: return to the context's own stream. */
int kws_set_stream(kws_ctx* ctx, void* hip_stream, int external);

/* Block the host until everything enqueued on the context's stream has finished. */
int kws_sync(kws_ctx* ctx);

/* Message of the last failure on this context ("" if none).  ctx == NULL: message of the last
 * failed kws_create on this thread. */
const char* kws_last_error(kws_ctx* ctx);

/* ---- front end: AudioProcessor.extract_features (kws/libs/audio_processor.py:235-278) ------- */

/* Parameters of psf.mfcc as the reference calls it.  Defaults (also set by kws_create):
 *   sample_rate 16000, n_samples 16000, frame_len 400, frame_step 160, nfft 512, nfilt 26,
 *   numcep 10, preemph 0.97, ceplifter 22                     (AudioConfig, audio_processor.py:37-46;
 *   psf.mfcc defaults for preemph / ceplifter / lowfreq 0 / highfreq sr/2 / appendEnergy True /
 *   rectangular window).  frame_len and frame_step are in samples (winlen*sr, winstep*sr rounded
 *   half up, as psf does).  Two kernels serve it:
 *     the float32 kernel (the hot path) for nfft == 512, frame_len <= 512 and a filterbank that spans bins 0..256;
 *     the float64 kernel for every other geometry: nfft a power of two in [64, 4096] or any value in [2, 2048] (the
 *     reference derives nfft = max(fft_size, int(winlen*samplerate)), audio_processor.py:268 -- e.g. 640 for a 40 ms
 *     window), any frame_len (frames longer than nfft are truncated, as numpy.fft.rfft does).
 *   Limits of both: nfilt <= 64, numcep <= min(nfilt, 32); anything else returns KWS_EUNSUPPORTED. */
int kws_set_frontend(kws_ctx* ctx, int sample_rate, int n_samples, int frame_len, int frame_step,
                     int nfft, int nfilt, int numcep, float preemph, int ceplifter);

/* Arithmetic of the front end -- replaces nothing in the reference (psf computes in float64 after a float32
 * pre-emphasis); PCM scaling and pre-emphasis are bit-exact float32 in both.
 *   KWS_FE_F32 (default): transform, mel sums, log and DCT in float32 -- the fast kernel.  Its rounding noise sits ~138 dB
 *     below a frame's strongest spectral component: cepstra within 1e-4 of the reference for frames whose mel bands span
 *     less than ~50 dB (noise, speech-like spectra), up to ~6e-4 on a clean tone over a quiet floor; logits within 1e-4.
 *   KWS_FE_F64: everything after framing in float64, as psf does -- cepstra within the float32 rounding of the output
 *     (~4e-6) on every input, at 4-5 times the kernel time (0.96 vs 0.21 ms per 4096 clips).  Geometries the float32 kernel is not built for run in
 *     float64 whatever this setting is; kws_frontend_math returns the arithmetic actually in use.  The streaming frame
 *     kernel (kws_stream_push_i16) is float32 only. */
#define KWS_FE_F32 0
#define KWS_FE_F64 1
int kws_set_frontend_math(kws_ctx* ctx, int math);
int kws_frontend_math(kws_ctx* ctx);

/* Selective float64 refinement of KWS_FE_F32 -- what makes the default front end meet psf's float64 arithmetic
 * (kws/libs/audio_processor.py:270-278) to 1e-4 on every frame.  A float32 transform leaves rounding noise a fixed distance
 * below the frame's strongest spectral component, so the float32 kernel measures, per frame, how far its weakest mel band
 * lies below its largest spectral bin: r = log(max bin power) - min log(mel energy) (natural-log units of power).  A frame
 * with r over `log_ratio` (default 10.2 = 44 dB) goes onto a device worklist and a second launch recomputes exactly those
 * rows in float64 (no host read-back; a batch with nothing listed pays one empty launch).  Frames under the threshold keep
 * the float32 kernel's bits.  Measured against the float64 oracle on 10.7 M frames of noise, tones, chirps, gated bursts,
 * mixtures and speech-like clips (tools/fe_precision_audit.py, profiles/r03_precision_audit.txt): no frame over 1e-4, the
 * worst of a seed's 297 000 frames 6.6e-5 .. 8.4e-5 (two seeds of 36: 9.3e-5 and 9.4e-5).  White noise lists ~0.3 % of its frames,
 * speech-like clips ~5 %, a clean tone over a quiet floor all of them.  (Until late in round 3 the flag was the span max - min
 * of the log-mel values with a threshold of 11.5: it cannot tell white noise, whose peak bin lies well below its strongest
 * band, from a tone, and left four frames of 2.4 M at 1.2-1.5e-4.)  The streaming push redoes a flagged frame in float64
 * inside the same launch.  log_ratio <= 0 switches the refinement off (the float32 kernel alone: up to 2e-3 on such
 * frames).  Takes effect from the next call; KWS_FE_F64 (always float64) is unaffected.  The macro keeps its round-3 name. */
#define KWS_FE_REFINE_SPAN_DEFAULT 10.2f
int kws_set_frontend_refine(kws_ctx* ctx, float log_ratio);
/* Frames that went through the float32 front end since kws_create (*frames_total), how many of them the refinement
 * recomputed in float64 (*frames_refined), and the number the last completed batched call listed (*last_call_refined).
 * Synchronises the context's stream.  Any pointer may be NULL. */
int kws_frontend_stats(kws_ctx* ctx, uint64_t* frames_total, uint64_t* frames_refined, int* last_call_refined);

/* Frames per clip (1 + ceil((n_samples - frame_len)/frame_step), sigproc.py:31-35) and numcep. */
int kws_frontend_shape(kws_ctx* ctx, int* num_frames, int* numcep);

/* Batched MFCC.  d_wav: int16 [B, n_samples] (PCM as stored in a 16-bit WAV; the kernel applies the
 * x/32768 scaling librosa.load applies, audio_processor.py:145).  d_out: float32
 * [B, 1, num_frames, numcep] -- the collated batch of SpeechCommandsDataLoader.__getitem__
 * (kws/libs/data_loader.py:96-105: float32 cast + channel axis) stacked by torch default_collate. */
int kws_mfcc_i16(kws_ctx* ctx, const int16_t* d_wav, int B, float* d_out);

/* Same, for a float32 signal in [-1, 1] that the caller has already decoded and possibly augmented
 * (time shift / background noise, audio_processor.py:154-159): d_wav float32 [B, n_samples].  This is
 * the exact input type of AudioProcessor.extract_features(signal) (audio_processor.py:235). */
int kws_mfcc_f32(kws_ctx* ctx, const float* d_wav, int B, float* d_out);

/* ---- model: DepthwiseSeparableConv (kws/libs/models.py:122-183) ----------------------------- */

/* Load weights: `blob` is a HOST pointer to the 20 state_dict tensors concatenated in state_dict
 * order (conv1.weight, conv1.bias, dsconv{1..4}.{depthwise,pointwise}.{weight,bias}, fc.weight,
 * fc.bias), float32; n_floats must be 25664 + 65*num_classes.  Copied to the device; the caller
 * keeps ownership.  Replaces KeywordSpottingModel.load (models.py:55-72). */
int kws_load_dscnn(kws_ctx* ctx, const float* blob, size_t n_floats, int num_classes);
/* The same for DepthwiseSeparableConv(num_classes, input_channels) with input_channels > 1 (models.py:125,135):
 * conv1.weight is then [64, input_channels, 10, 10], n_floats = 6400*input_channels + 19264 + 65*num_classes, and
 * kws_forward_f32 takes d_feat float32 [B, input_channels, 99, 10]: conv1 runs in a general kernel and the fused kernel
 * starts at block 1.  The wav -> label entry points need input_channels == 1 (an MFCC map has one channel). */
int kws_load_dscnn_ex(kws_ctx* ctx, const float* blob, size_t n_floats, int num_classes, int input_channels);

/* Forward on precomputed features.  d_feat: float32 [B,1,99,10]; d_logits: float32
 * [B,num_classes]; d_label: int32 [B] = argmax (first maximum wins, torch.max semantics,
 * kws/libs/training.py:371) or NULL.  Replaces DepthwiseSeparableConv.forward (models.py:160-183). */
int kws_forward_f32(kws_ctx* ctx, const float* d_feat, int B, float* d_logits, int32_t* d_label);

/* The same forward for a feature map of ANY size: d_feat float32 [B, input_channels, T, F] -- DepthwiseSeparableConv.forward
 * takes any [B,C,T,F] (models.py:160-183; the global average pool is adaptive) and AudioConfig.clip_duration_ms /
 * num_cepstral_coeffs change T and F (audio_processor.py:37-46).  T x F == 99 x 10 runs the fused LDS-resident kernel;
 * any other map runs composed through HBM: conv1 (10x10, stride 2, padding 2) -> four depthwise-separable blocks (each
 * adds its relu(bias) ring) -> global average pool + fc + argmax.  Needs T >= 6, F >= 6 and (T+4)*(F+4) <= 40960.
 * kws_infer_i16 / kws_infer_f32 / kws_infer_host_i16 follow kws_frontend_shape through this entry. */
int kws_forward_map_f32(kws_ctx* ctx, const float* d_feat, int B, int T, int F, float* d_logits, int32_t* d_label);
/* Parity aid for the composed path: also stores every stage's output to d_layers, stage after stage, each stage as
 * [B][64][H][W] -- conv1 (H1 x W1, H1 = (T-6)/2+1, W1 = (F-6)/2+1), then blocks 1..4 WITH their rings ((H1+2k) x (W1+2k)).
 * A 99 x 10 map takes the composed path here too (an independent check of the fused kernel). */
int kws_forward_map_debug_f32(kws_ctx* ctx, const float* d_feat, int B, int T, int F, float* d_logits, int32_t* d_label,
                              float* d_layers);

/* One depthwise-separable block on an arbitrary map -- replaces DepthwiseSeparableConvBlock.forward
 * (kws/libs/models.py:108-119) used on its own: depthwise Conv2d(C_in, C_in, kernel_size, stride, padding, groups=C_in)
 * + bias, then pointwise Conv2d(C_in, C_out, 1, padding=padding) + bias, then ReLU.  d_x float32 [B,C_in,H,W]; d_dw_w
 * [C_in,1,k,k], d_dw_b [C_in], d_pw_w [C_out,C_in,1,1], d_pw_b [C_out] (device pointers, the module's parameters as torch
 * stores them); d_out float32 [B, C_out, Ho + 2*padding, Wo + 2*padding] with Ho = (H + 2*padding - k)/stride + 1: the
 * pointwise padding adds a ring equal to relu(bias), as in the reference.  The four blocks of the DS-CNN do NOT go
 * through this entry point: they run fused inside kws_forward_f32. */
int kws_dsblock_forward_f32(kws_ctx* ctx, const float* d_x, int B, int C_in, int H, int W, const float* d_dw_w,
                            const float* d_dw_b, const float* d_pw_w, const float* d_pw_b, int C_out, int kernel_size,
                            int stride, int padding, float* d_out);

/* Fused wav -> label: kws_mfcc_i16 into an internal workspace, then kws_forward_f32.  This is the
 * shape of inference(wav) -> label (kws/inference/inference_local.py:67-81), batched. */
int kws_infer_i16(kws_ctx* ctx, const int16_t* d_wav, int B, float* d_logits, int32_t* d_label);

/* The same for float32 signals in [-1, 1] (d_wav float32 [B, n_samples]): what librosa.load hands the reference for files
 * that are not 16-bit PCM (24-bit, float, stereo mixed down in float: audio_processor.py:145). */
int kws_infer_f32(kws_ctx* ctx, const float* d_wav, int B, float* d_logits, int32_t* d_label);

/* The same from HOST memory to HOST memory -- what the reference does between the decoded audio and the model input:
 * DataLoader workers collate batches into pinned memory and the trainer calls inputs.to(device)
 * (kws/libs/data_loader.py:96-105, train.py:108-121, kws/libs/training.py:286).  h_wav: int16 [B, n_samples] in host
 * memory (pageable or pinned), h_logits: float32 [B, C], h_label: int32 [B] or NULL, both host.  The batch is cut into
 * chunks; a pool of host threads packs chunk k+1 into pinned staging while chunk k travels over PCIe on a copy stream,
 * chunk k-1 runs MFCC + DS-CNN and the results of chunk k-2 return on a second copy stream.  Synchronous: the results
 * are complete on return.  A pinned h_wav (hipHostMalloc / hipHostRegister / torch pin_memory) is read by the DMA
 * directly, without the pack stage. */
int kws_infer_host_i16(kws_ctx* ctx, const int16_t* h_wav, int B, float* h_logits, int32_t* h_label);
/* The same in two halves, so that a caller with MANY batches keeps the pipeline full across them (kws_infer_host_i16 alone
 * fills and drains it inside every call): submit enqueues every chunk of the batch and returns -- the caller's pageable h_wav
 * has been packed into pinned staging by then and may be reused; a PINNED h_wav, h_logits and h_label must stay valid until
 * the wait -- and hands back a ticket; kws_infer_host_wait(ctx, ticket) blocks until every batch up to that ticket has its
 * results in its h_logits / h_label (ticket 0: everything in flight).  Submitting batch k+1 before waiting for batch k lets
 * k+1's pack and H2D run under k's kernels.  Results of older batches are also delivered whenever a later submit needs
 * their staging slot.  kws_infer_host_i16 == submit + wait. */
int kws_infer_host_submit_i16(kws_ctx* ctx, const int16_t* h_wav, int B, float* h_logits, int32_t* h_label, uint64_t* ticket);
int kws_infer_host_wait(kws_ctx* ctx, uint64_t ticket);
/* Pipeline shape of kws_infer_host_i16: clips per chunk (default 1024), staging slots in flight (default 3, 2..16), host
 * threads of the pack stage (default min(16, cores/2); negative = pack on the calling thread).  0 keeps a default.  A batch
 * smaller than chunk x slots is cut into `slots` chunks (not below 128 clips), so that its stages overlap too. */
int kws_ingest_config(kws_ctx* ctx, int chunk_clips, int n_slots, int pack_threads);

/* Pre-size the internal workspaces for batches up to max_batch (otherwise grown on demand, which
 * allocates and must not happen inside stream capture). */
int kws_reserve(kws_ctx* ctx, int max_batch);

/* Arithmetic of conv1 and the 1x1 (pointwise) convolutions on the matrix cores -- replaces nothing in the reference
 * (torch's f32 conv2d, kws/libs/models.py:104-106); every setting keeps the logits within 1e-4 of it.
 *   KWS_PW_PAIR_F16 (default): every f32 operand, after an exact power-of-two scaling into f16's range, as an f16 pair
 *     hi + lo (22 significant bits); three v_mfma_f32_32x32x16_f16 per k-block, f32 accumulate.  Weights are scaled per
 *     layer at load time, activations per CLIP: the kernel keeps them in LDS in power-of-two units whose exponents it derives,
 *     two layers ahead, from the measured maximum of an earlier stage and bounds that hold for any input -- nothing can
 *     overflow, and logits differ from a float64 evaluation by what a plain f32 evaluation differs by.
 *   KWS_PW_SPLIT_BF16: every f32 operand is split exactly into three bf16 pieces (hi + mid + lo == x)
 *     and the six piece products of combined order <= 2 are accumulated in f32 by v_mfma_f32_32x32x16_bf16;
 *     each bf16 x bf16 product is exact, the dropped terms are <= 2^-24 relative -- the size of one f32
 *     rounding.  Twice the matrix instructions of the pair.
 *   KWS_PW_F32: v_mfma_f32_32x32x2_f32 (shares the FP32 datapath with the VALU; slower).
 * Changing it invalidates a captured streaming graph (re-captured on the next push). */
#define KWS_PW_F32 1
#define KWS_PW_SPLIT_BF16 4
#define KWS_PW_PAIR_F16 5
int kws_set_pointwise_math(kws_ctx* ctx, int math);

/* Debug/parity aid: run the DS-CNN and also store every stored activation per clip to d_act
 * (float32 [B, KWS_ACT_FLOATS_PER_CLIP]): conv1 out [64][47*3], block1 out interior [64][47*3],
 * block2 out interior [64][49*5], block3 out interior [64][51*7], pooled mean [64], block4 out interior
 * [64][53*9] (never stored by the product kernel: it is pooled in registers).  "Interior" =
 * the pointwise output without the relu(bias) ring its padding=1 adds (models.py:104-106).
 * use_mfma: KWS_PW_SPLIT_BF16 (4) / KWS_PW_F32 (1) = the two matrix-core kernels, 0 = a variant whose
 * pointwise / conv1 GEMMs run on the VALU (an independent check of the matrix-core operand mappings). */
#define KWS_ACT_FLOATS_PER_CLIP (64 * (141 + 141 + 245 + 357) + 64 + 64 * 477)
int kws_forward_debug_f32(kws_ctx* ctx, const float* d_feat, int B, float* d_logits, int32_t* d_label,
                          float* d_act, int use_mfma);

/* ---- streaming: 10 ms hops over concurrent streams (BASELINE.json config 5) ------------------------ */

/* The reference's streaming use case is VAD-segmented capture followed by one whole-clip inference
 * (kws/inference/inference_local.py:114-192).  Here every push of frame_step (160) new samples per stream
 * completes one MFCC frame per stream (frames of the continuous signal: frame f = samples
 * [160 f, 160 f + 400)), appends it to a ring of num_frames (99) frames, and classifies the last 99 frames
 * (one second) of every stream.  Until a stream has produced 99 frames the missing rows are zeros. */
int kws_stream_open(kws_ctx* ctx, int n_streams);
int kws_stream_close(kws_ctx* ctx);
/* d_hop: int16 [n_streams, frame_step] new samples; d_logits float32 [n_streams, C] (NULL: features only);
 * d_label int32 [n_streams] or NULL.  With logits and the default pointwise math a push is ONE launch: each stream's
 * workgroup of the DS-CNN kernel computes the stream's new frame in its prologue and the last workgroup advances the hop
 * counter.  Features-only pushes, and pushes under KWS_PW_F32 / the VALU check, are the frame kernel (which then advances
 * the counter) followed by the DS-CNN kernel.  use_graph != 0 replays a MULTI-launch push as a hipGraph (built on first use
 * for the given pointer triple); the one-launch push is always launched directly -- a one-node graph replay is 8 us slower
 * than the plain launch on this stack (tools/graph_overhead.hip). */
int kws_stream_push_i16(kws_ctx* ctx, const int16_t* d_hop, float* d_logits, int32_t* d_label, int use_graph);
/* Shape of the one-launch push: workgroups per stream.  1: one workgroup owns a stream's whole DS-CNN (the layout of the
 * batched kernel).  2 / 4: the stream's network is cut into that many TIME TILES, one workgroup each -- every stage's rows
 * a tile's share of block 4's output depends on are recomputed inside the tile (about four rows of halo per side), nothing
 * is exchanged between the workgroups but 64 pooled partial sums per tile at the very end, where the last workgroup to
 * arrive adds them in tile order and runs fc + argmax.  At 64 streams one workgroup per stream leaves three quarters of the
 * CUs idle and the push latency is one clip's serial path; four tiles bring the kernel from 37.6 to ~17 us.  0 (default):
 * 4 up to 64 streams, 2 up to 128, else 1.  Logits of different shapes agree to the float32 rounding of the pooled sums
 * (their order of addition differs); a given shape is deterministic. */
int kws_stream_cluster(kws_ctx* ctx, int workgroups_per_stream);
/* Zero-copy result delivery for the one-launch push (the latency path of BASELINE config 5).  enable != 0: the context
 * allocates pinned, device-mapped host arrays; from then on every push that asks for logits ALSO writes its logits
 * [n_streams, C] and labels [n_streams] there with system-scope stores, and the workgroup that finishes last raises a flag in
 * host memory.  kws_stream_wait_host spins on that flag (falling back to the stream after ~2 ms) and returns the host
 * arrays (owned by the context, overwritten by the next push): the caller has the results in hand without a
 * hipStreamSynchronize round trip and without a device-to-host copy.  d_logits / d_label of kws_stream_push_i16 are still
 * written.  Call after kws_stream_open and kws_load_dscnn; kws_stream_open / kws_stream_close / enable == 0 release it. */
int kws_stream_host_results(kws_ctx* ctx, int enable);
int kws_stream_wait_host(kws_ctx* ctx, const float** h_logits, const int32_t** h_label);
/* The whole hop from host memory to host memory in one call: h_hop int16 [n_streams, frame_step] (pageable is fine) is copied
 * into pinned device-mapped memory, the one-launch push reads it from there (no H2D submission), and the call returns when
 * the kernel has delivered logits [n_streams, C] and labels [n_streams] to the context's host arrays (see above; valid until
 * the next push).  Enables kws_stream_host_results by itself.  The reference's live path hands a captured utterance to the
 * model from host memory (kws/inference/inference_local.py:168-192); this is its per-hop counterpart. */
int kws_stream_push_host_i16(kws_ctx* ctx, const int16_t* h_hop, const float** h_logits, const int32_t** h_label);
/* Synchronises and returns the feature ring (float32 [n_streams, num_frames, numcep], device memory owned
 * by the context) and the number of pushes so far; the newest frame is row (hops - K) mod num_frames, K = ceil(frame_len / frame_step)
 * hops per frame (3 for the reference's 400 / 160). */
int kws_stream_state(kws_ctx* ctx, const float** d_feat_ring, int* hops);
/* Copy the raw feature ring (float32 [n_streams, num_frames, numcep], ring order) into caller memory. */
int kws_stream_copy_features(kws_ctx* ctx, float* d_out);

/* Energy endpointer for the streams opened with kws_stream_open (build-defined: it stands in for
 * the webrtcvad endpointing of the reference's live loop, kws/inference/inference_local.py:131-166, with the same
 * hysteresis at hop granularity).  Call after kws_stream_push_i16.  The hop is voiced when the newest frame's log
 * energy (cepstrum 0) exceeds the threshold; an utterance opens when more than 80 % of the last on_window hops are
 * voiced (:151), closes when more than 90 % of the last off_window hops are unvoiced (:161); 400 ms / 800 ms in the
 * reference = 40 / 80 hops of 10 ms.  d_state int32 [n_streams]: bit 0 = inside an utterance, bits 1-2 = event at
 * this hop (1 opened, 2 closed).  The history lives in the context; kws_stream_open or a change of windows resets it. */
int kws_stream_vad_f32(kws_ctx* ctx, float log_energy_threshold, int on_window, int off_window, int32_t* d_state);

/* ---- cnn-trad-fpool3 (build-defined: the reference only names the model, test.py:80) ----
 * Sainath & Parada's cnn-trad-fpool3 on the [1,99,10] MFCC map with SAME padding: conv 64 x (20x8) + ReLU,
 * max-pool 1x3 over frequency, conv 64 x (10x4) + ReLU, flatten, Linear 32, Linear 128 + ReLU, Linear C.
 * blob = the ten state_dict tensors in order, float32: conv1.weight [64,1,20,8], conv1.bias [64], conv2.weight
 * [64,64,10,4], conv2.bias [64], lin.weight [32,19008], lin.bias [32], dnn.weight [128,32], dnn.bias [128],
 * fc.weight [C,128], fc.bias [C] (host pointer, copied). */
int kws_load_cnn_trad(kws_ctx* ctx, const float* blob, size_t n_floats, int num_classes);
/* float32 [B,1,99,10] features -> logits float32 [B,C] and labels int32 [B] (d_label may be NULL).  The
 * convolution output (76 KB per clip) goes through a context workspace that grows on demand. */
int kws_forward_cnn_trad_f32(kws_ctx* ctx, const float* d_feat, int B, float* d_logits, int32_t* d_label);
/* Arithmetic of cnn-trad-fpool3's three GEMM layers on the matrix cores.  KWS_CT_F16_PAIR (default): every f32 operand as an
 * f16 pair hi + lo' 2^-11 (22 bits) after an exact power-of-two scaling into f16's range -- per layer for the weights, per clip
 * for the activations from rigorous bounds, so no input can overflow -- three f16 MFMAs per k-block; logits differ from a
 * float64 evaluation by what a plain f32 evaluation differs by.  KWS_CT_BF16_TRIPLE: the exact three-way bf16 split the DS-CNN
 * uses (six MFMAs per k-block, twice the matrix time). */
#define KWS_CT_F16_PAIR 0
#define KWS_CT_BF16_TRIPLE 1
int kws_set_cnn_trad_math(kws_ctx* ctx, int math);

/* Fused wav -> label for this model (BASELINE.json configs[2]): kws_mfcc_i16 into the context workspace, then
 * kws_forward_cnn_trad_f32, on the context's stream.  Same arguments and errors as kws_infer_i16. */
int kws_infer_cnn_trad_i16(kws_ctx* ctx, const int16_t* d_wav, int B, float* d_logits, int32_t* d_label);

/* ---- posteriors (build-defined: the reference's scripts stop at argmax of the logits,
 * kws/libs/training.py:371) ------------------------------------------------------------------------------ */

/* Softmax over the C logits of every row: float32 [B,C] -> float32 [B,C] (device pointers, C <= 64). */
int kws_softmax_f32(kws_ctx* ctx, const float* d_logits, int B, int C, float* d_prob);

/* Streaming posterior smoothing for the streams opened with kws_stream_open: softmax of this hop's logits
 * [n_streams, C], then the mean over the last `window` hops per stream (fewer while the history is shorter),
 * written to d_smoothed [n_streams, C]; d_label (may be NULL) = argmax of the smoothed vector, first maximum
 * wins.  The history lives in the context and is reset by kws_stream_open / a change of window or C. */
int kws_stream_smooth_f32(kws_ctx* ctx, const float* d_logits, int C, int window, float* d_smoothed, int32_t* d_label);

/* ---- augmentation of the training transform (kws/libs/audio_processor.py:151-159,172-233) ---------- */

/* out[b][i] = (silence[b] ? 0 : wav[b][i - shift[b]] / 32768, 0 outside the clip) + bg_vol[b] * bg[bg_off[b] + i]
 * as float32 [B, n_samples], ready for kws_mfcc_f32.  d_shift int32 [B] (NULL: no shift), d_bg float32
 * [bg_len] background pool (NULL: no mix) with per-clip offsets int32 [B] and volumes float32 [B],
 * d_silence uint8 [B] (NULL: none).  The random draws stay with the caller (host), as in the reference. */
int kws_augment_i16(kws_ctx* ctx, const int16_t* d_wav, int B, const int32_t* d_shift, const float* d_bg, int bg_len,
                    const int32_t* d_bg_off, const float* d_bg_vol, const uint8_t* d_silence, float* d_out);

/* Diagnostics: the same forward with per-clip shader-clock stamps (s_memtime of thread 0) at the phase
 * boundaries of the DS-CNN kernel, uint64 [B, KWS_DSCNN_STAMPS]: 0 start, 1 features staged, 2/3 conv1
 * done / barrier, 4/5 .. 10/11 blocks 1..4 done / barrier, 12 end; [14], [15] = 100 MHz real-time
 * counter at start / end.  mode: KWS_PW_SPLIT_BF16 (4) / KWS_PW_F32 (1) = the matrix-core kernels, 0 = VALU
 * cross-check variant, 2 / 3 = timing ablations of the f32 kernel (matrix core only / stencil only), 6 = of
 * the split kernel (no stencil): wrong results by construction.  Never used on the product path. */
#define KWS_DSCNN_STAMPS 16
int kws_forward_stamps_f32(kws_ctx* ctx, const float* d_feat, int B, float* d_logits, uint64_t* d_stamps, int mode);

/* ---- sigproc operators (kws/libs/speech_features/sigproc.py) -------------------------------- */

/* preemphasis(signal, coeff) (sigproc.py:93-103), float32 [n] -> float32 [n]. */
int kws_preemphasis_f32(kws_ctx* ctx, const float* d_signal, int n, float coeff, float* d_out);
/* framesig(signal, frame_len, frame_step, winfunc) (sigproc.py:14-52): d_window float32 [frame_len]
 * or NULL for rectangular; d_frames float32 [num_frames, frame_len], num_frames as kws_frontend_shape. */
int kws_framesig_f32(kws_ctx* ctx, const float* d_signal, int n, int frame_len, int frame_step,
                     const float* d_window, float* d_frames);
/* magspec / powspec (sigproc.py:55-90) with NFFT = 512: d_frames float32 [num_frames, frame_len]
 * (frame_len <= 512 is zero-padded, > 512 truncated) -> float32 [num_frames, 257];
 * power != 0 gives 1/NFFT * |X|^2, power == 0 gives |X|. */
int kws_spec512_f32(kws_ctx* ctx, const float* d_frames, int num_frames, int frame_len, int power,
                    float* d_spec);

/* The same for any NFFT (sigproc.py:55-90 takes any): NFFT == 512 runs the kernel above, any other length -- a power of
 * two in [64, 4096] or any value in [2, 2048] -- a float64 transform (FFT / direct DFT) with float32 output. */
int kws_spec_f32(kws_ctx* ctx, const float* d_frames, int num_frames, int frame_len, int nfft, int power, float* d_spec);

/* ---- measurement --------------------------------------------------------------------------- */

/* Per-kernel device timing with HIP events on the context's stream.  kws_prof_enable(ctx, on): 0 = off, 1 = every
 * kernel launch is bracketed by events, n > 1 = every n-th launch of each kernel id is (a pair of events costs the
 * stream ~7 us: 2 % of a 4096-clip step, 8 % of a 1024-clip one, when every launch carries one); kws_prof_read
 * synchronises and returns the summed milliseconds and the number of TIMED launches per kernel id since the last
 * kws_prof_reset.  The kernels of an eager kws_stream_push_i16 are timed
 * too (KWS_K_DSCNN for the one-launch push; KWS_K_STREAM_FRAME + KWS_K_DSCNN for the two-launch routes); a push replayed
 * as a hipGraph is not (events cannot bracket a node). */
enum { KWS_K_MFCC = 0, KWS_K_DSCNN = 1, KWS_K_CNNTRAD_CONV = 2, KWS_K_CNNTRAD_DENSE = 3, KWS_K_STREAM_FRAME = 4, KWS_K_MFCC_F64 = 5, KWS_K_MFCC_REFINE = 6, KWS_K_COUNT = 7 };
int kws_prof_enable(kws_ctx* ctx, int on);
int kws_prof_reset(kws_ctx* ctx);
int kws_prof_read(kws_ctx* ctx, int kernel_id, double* total_ms, int* launches);
/* Name of the device kernel behind a kernel id (as it appears in rocprofv3 traces). */
const char* kws_kernel_name(int kernel_id);

/* ---- host-only helpers (no GPU needed; used by the CPU test-suite) -------------------------- */

/* nfilt+2 mel bin edges exactly as psf get_filterbanks computes them. */
int kws_host_mel_edges(int nfilt, int nfft, int sample_rate, int* edges_out);
/* Dense float32 [nfilt, nfft/2+1] filterbank expanded from the sparse per-lane tables the kernel
 * uses (so the sparse decomposition can be checked against the oracle's dense matrix on CPU). */
int kws_host_mel_dense(int nfilt, int nfft, int sample_rate, float* fb_out);
/* Lane layout of the sparse mel evaluation: for each of the nfilt+1 inter-edge segments the first lane and the number
 * of lanes (chunks of 8 bins) it occupies; *lanes_used = lanes in use including idle padding; *row_safe = 1 when no
 * segment straddles a 16-lane DPP row (the kernel then shifts with row_shl operands).  first_lane_out / n_lanes_out
 * hold nfilt+1 ints; lanes_used / row_safe may be NULL. */
int kws_host_mel_layout(int nfilt, int nfft, int sample_rate, int* first_lane_out, int* n_lanes_out, int* lanes_used, int* row_safe);
/* float32 [numcep, nfilt] DCT-II(ortho) x lifter table the kernel uses. */
int kws_host_dct_lifter(int nfilt, int numcep, int ceplifter, float* out);

#ifdef __cplusplus
}
#endif
#endif /* KWS_HIP_H */
