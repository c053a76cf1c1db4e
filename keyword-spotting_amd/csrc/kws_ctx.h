// The context behind the C ABI (include/kws_hip.h) and the error plumbing shared by the translation units that
// implement it (kws_api.hip, kws_ingest.hip).
#pragma once
#include <new>

#include "kws_internal.h"

namespace kws {
struct Ingest;  // host-ingest pipeline state (kws_ingest.hip)
void ingest_free(kws_ctx* c);
}  // namespace kws

using kws::FrontendParams; using kws::FrontendTables; using kws::DscnnWeights; using kws::CnnTradWeights; using kws::NFFT;

struct kws_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t order_ev = nullptr;  // orders the new stream behind the old one in kws_set_stream
    std::string err;

    // front end
    int sample_rate = 16000, nfft = NFFT, ceplifter = 22;
    FrontendParams fp{};
    bool fe_ready = false;
    bool fe_fast_ok = false;      // the float32 kernel covers this geometry (nfft 512, frame_len <= 512, sparse mel layout fits)
    int fe_math = KWS_FE_F32;     // requested arithmetic (kws_set_frontend_math); geometries without a fast kernel run in float64 anyway
    void* d_fe = nullptr;  // one allocation holding all front-end tables
    FrontendTables ft{};
    double* d_spec_tw64 = nullptr;  // float64 twiddles of kws_spec_f32's last transform length
    int spec_nfft = 0;
    // selective float64 refinement of the float32 front end (kws_set_frontend_refine): worklist + counters
    float refine_span = KWS_FE_REFINE_SPAN_DEFAULT;
    int* d_refine = nullptr;          // int[8] counters followed by the list, one allocation
    int refine_cap = 0;               // frames the list holds
    unsigned long long frames_seen = 0;  // frames through the float32 kernels since kws_create

    // model
    float* d_model = nullptr;
    DscnnWeights mw{};
    bool model_ready = false;
    int pw_math = KWS_PW_PAIR_F16;    // kernel variant of the product entry points
    // cnn-trad-fpool3
    void* d_cnntrad = nullptr;
    CnnTradWeights tw{};
    bool cnntrad_ready = false;
    int cnntrad_math = KWS_CT_F16_PAIR;
    float* d_conv_ws = nullptr;
    size_t conv_ws_floats = 0;

    // workspace (MFCC features between the two kernels of kws_infer_i16)
    float* d_feat_ws = nullptr;
    size_t feat_ws_floats = 0;

    // streaming state (kws_stream_*): per-stream PCM ring, feature ring, hop counter, optional graph
    int n_streams = 0, ring_len = 0;
    int16_t* d_pcm_ring = nullptr;
    float* d_feat_ring = nullptr;
    int* d_hops = nullptr;
    int stream_cluster = 0;           // workgroups per stream of the fused push (kws_stream_cluster; 0 = by stream count)
    float* d_cl_part = nullptr;       // [n_streams][4][64] pooled partial sums of a stream's time tiles
    int* d_cl_count = nullptr;        // [n_streams]
    // zero-copy result delivery of the one-launch push (kws_stream_host_results): pinned, device-mapped host memory
    float* h_stream_logits = nullptr;  // [n_streams][num_classes at enable time]
    int32_t* h_stream_label = nullptr;
    int* h_stream_flag = nullptr;
    int16_t* h_stream_hop = nullptr;   // [n_streams][frame_step]: the hop of kws_stream_push_host_i16, read by the kernel over PCIe
    float* d_hr_logits = nullptr;      // device-side outputs of kws_stream_push_host_i16 (the kernel writes both copies)
    int32_t* d_hr_label = nullptr;
    int host_results_classes = 0;
    int pushes_enqueued = 0;           // pushes since kws_stream_open = the device's hop counter once the stream has drained
    int host_push = 0;                 // the newest push that delivers to host memory (the flag reads this when it is done)
    bool last_push_host = false;
    hipGraphExec_t stream_graph = nullptr;
    const void* graph_key[3] = {nullptr, nullptr, nullptr};
    // posterior smoothing history (kws_stream_smooth_f32): ring [n_streams][window][C], sum [n_streams][C], hop count
    float* d_post_ring = nullptr;
    float* d_post_sum = nullptr;
    int* d_post_count = nullptr;
    int post_window = 0, post_classes = 0;
    // energy endpointer (kws_stream_vad_f32): voiced flags [n_streams][off_window], (cursor, triggered) [n_streams][2]
    unsigned char* d_vad_flags = nullptr;
    int* d_vad_state = nullptr;
    int vad_on = 0, vad_off = 0;

    // host ingest (kws_infer_host_i16): staging rings, copy streams, pack threads -- created on first use
    kws::Ingest* ingest = nullptr;

    // profiling
    bool prof = false;
    int prof_every = 1;                    // bracket every prof_every-th launch of a kernel id (kws_prof_enable)
    unsigned prof_seen[KWS_K_COUNT] = {};  // launches per kernel id since kws_prof_reset, timed or not
    struct EvPair {
        hipEvent_t a, b;
    };
    std::vector<EvPair> ev[KWS_K_COUNT];
    size_t ev_used[KWS_K_COUNT] = {};
    double ms_total[KWS_K_COUNT] = {};
    long launches[KWS_K_COUNT] = {};
};

inline thread_local std::string g_create_err;

inline int fail(kws_ctx* c, int code, const std::string& msg) {
    if (c)
        c->err = msg;
    else
        g_create_err = msg;
    return code;
}
inline int fail_hip(kws_ctx* c, hipError_t e, const char* what) {
    return fail(c, KWS_EHIP, std::string(what) + ": " + hipGetErrorString(e));
}
// No exception may cross the C ABI (ctypes would terminate the process): entry points that allocate host containers or
// start threads run their body between these two.
#define KWS_GUARD_BEGIN try {
#define KWS_GUARD_END(c, fn)                                                          \
    }                                                                                 \
    catch (const std::bad_alloc&) { return fail(c, KWS_ENOMEM, fn ": out of host memory"); } \
    catch (...) { return fail(c, KWS_EHIP, fn ": unexpected C++ exception"); }
#define HIP_TRY(c, expr)                                   \
    do {                                                   \
        hipError_t _e = (expr);                            \
        if (_e != hipSuccess) return fail_hip(c, _e, #expr); \
    } while (0)

