// C ABI of libkws_hip.so (declared in include/kws_hip.h): context, host-side table construction,
// weight repacking, workspace and per-kernel event timing.  No torch types, no exceptions across the
// boundary, no CPU fallback.
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>

#include "kws_ctx.h"

namespace kws {

const char* const kKernelNames[KWS_K_COUNT] = {"kws_mfcc_i16_kernel", "kws_dscnn_fwd_kernel", "kws_cnntrad_conv_kernel",
                                               "kws_cnntrad_dense_kernel", "kws_stream_frame_kernel", "kws_mfcc_f64_kernel",
                                               "kws_mfcc_refine_kernel"};

// ------------------------------------------------------------------------------------------------
// Host tables (double precision, then rounded once to float32).

static double hz2mel(double hz) { return 2595.0 * std::log10(1.0 + hz / 700.0); }
static double mel2hz(double mel) { return 700.0 * (std::pow(10.0, mel / 2595.0) - 1.0); }

// psf get_filterbanks: nfilt+2 points equally spaced in mel (numpy.linspace arithmetic: i*step + start,
// last point = stop), converted to FFT bins by floor((nfft+1)*hz/samplerate).
static void mel_edges(int nfilt, int nfft, int sample_rate, std::vector<int>& edges) {
    const double lowmel = hz2mel(0.0), highmel = hz2mel(sample_rate / 2.0);
    const int num = nfilt + 2;
    const double step = (highmel - lowmel) / (double)(num - 1);
    edges.resize(num);
    for (int i = 0; i < num; ++i) {
        volatile double prod = (double)i * step;  // two roundings, as numpy does (no fused multiply-add)
        double mel = prod + lowmel;
        if (i == num - 1) mel = highmel;
        edges[i] = (int)std::floor((nfft + 1) * mel2hz(mel) / sample_rate);
    }
}

bool build_mel_host(int nfilt, int nfft, int sample_rate, MelHost& out, std::string& err) {
    if (nfilt < 1 || nfilt > MAX_NFILT) {
        err = "nfilt must be in [1, 64]";
        return false;
    }
    mel_edges(nfilt, nfft, sample_rate, out.edges);
    for (int i = 0; i + 1 < (int)out.edges.size(); ++i) {
        if (out.edges[i + 1] < out.edges[i] || out.edges[i] < 0 || out.edges[i + 1] > nfft / 2) {
            err = "mel edges are not monotone inside [0, nfft/2]";
            return false;
        }
    }
    out.k0.assign(64, nfft / 2);
    out.rw.assign(MEL_CHUNK * 64, 0.f);
    out.fw.assign(MEL_CHUNK * 64, 0.f);
    out.gather.assign(64, 0u);
    out.slot.assign(nfft / 2, 0);
    std::vector<int> seg_first(nfilt + 2, 0), seg_count(nfilt + 2, 0);
    // segment s = bins [edge_s, edge_s+1): rising side of filter s (s < nfilt), falling side of filter s-1 (s >= 1);
    // it is cut into chunks of MEL_CHUNK bins, one chunk per lane, the chunks of a segment on adjacent lanes.
    int dense_lanes = 0;
    for (int s = 0; s <= nfilt; ++s) {
        seg_count[s] = (out.edges[s + 1] - out.edges[s] + MEL_CHUNK - 1) / MEL_CHUNK;
        dense_lanes += seg_count[s];
    }
    if (dense_lanes > 64) {
        err = "mel filterbank needs more than 64 chunks of 8 bins";
        return false;
    }
    // Lane layout: no segment straddles a 16-lane DPP row (idle lanes pad the rows), so the segmented sums can shift
    // with row_shl:1/2/4 fused into v_fmac_f32_dpp.  Every filterbank that fits 64 dense chunks at nfft = 512 and
    // the sample rates tried also fits this way (many filters = short segments); one that does not is refused.
    {
        int cursor = 0;
        for (int s = 0; s <= nfilt; ++s) {
            if (cursor % 16 + seg_count[s] > 16) cursor = (cursor + 15) / 16 * 16;
            seg_first[s] = cursor;
            cursor += seg_count[s];
            if (seg_count[s] > 8 || cursor > 64) {
                err = "mel filterbank does not fit 64 lanes with every segment (<= 8 chunks) inside one 16-lane row";
                return false;
            }
        }
    }
    int nchunks = 0;  // lanes in use (idle padding lanes included)
    for (int s = 0; s <= nfilt; ++s) {
        const int lo = out.edges[s], hi = out.edges[s + 1];
        int c = seg_first[s];
        for (int k0 = lo; k0 < hi; k0 += MEL_CHUNK, ++c) {
            out.k0[c] = k0;
            for (int i = 0; i < MEL_CHUNK && k0 + i < hi; ++i) out.slot[k0 + i] = MEL_STRIDE * c + i;
            for (int i = 0; i < MEL_CHUNK && k0 + i < hi; ++i) {
                const double k = k0 + i, width = (double)(hi - lo);
                if (s < nfilt) out.rw[i * 64 + c] = (float)((k - lo) / width);
                if (s >= 1) out.fw[i * 64 + c] = (float)((hi - k) / width);
            }
        }
        nchunks = std::max(nchunks, c);
    }
    // which neighbours (chunk + 1, + 2, + 4) share a chunk's segment: drives the in-register segmented sums
    out.seg.assign(64, 0);
    bool deep = false;
    for (int s = 0; s <= nfilt; ++s) {
        if (seg_count[s] > 8) {
            err = "a mel segment spans more than 8 chunks of 8 bins";
            return false;
        }
        deep = deep || seg_count[s] > 4;
        for (int i = 0; i < seg_count[s]; ++i)
            for (int d = 0; d < 3; ++d)
                if (i + (1 << d) < seg_count[s]) out.seg[seg_first[s] + i] |= 1 << d;
    }
    for (int c = 0; c < 64; ++c) out.seg[c] |= (deep ? 128 : 0) | 64;  // bit 6: row-safe layout (always)
    for (int j = 0; j < nfilt; ++j) {
        if (seg_count[j] > 255 || seg_count[j + 1] > 255) {
            err = "mel segment too long";
            return false;
        }
        out.gather[j] = (uint32_t)seg_first[j] | ((uint32_t)seg_count[j] << 8) | ((uint32_t)seg_first[j + 1] << 16) |
                        ((uint32_t)seg_count[j + 1] << 24);
    }
    out.n_chunks = nchunks;
    return true;
}

void build_dct_lifter_host(int nfilt, int numcep, int ceplifter, std::vector<float>& out) {
    out.assign((size_t)numcep * nfilt, 0.f);
    const double pi = 3.14159265358979323846;
    for (int i = 0; i < numcep; ++i) {
        const double lift = ceplifter > 0 ? 1.0 + (ceplifter / 2.0) * std::sin(pi * i / ceplifter) : 1.0;
        for (int j = 0; j < nfilt; ++j) {
            const double d = (i == 0) ? std::sqrt(1.0 / nfilt)
                                      : std::sqrt(2.0 / nfilt) * std::cos(pi * i * (2 * j + 1) / (2.0 * nfilt));
            out[(size_t)i * nfilt + j] = (float)(lift * d);
        }
    }
}

void build_twiddle_host(std::vector<float2>& out) {
    out.resize(NFFT);
    const double pi = 3.14159265358979323846;
    for (int k = 0; k < NFFT; ++k) {
        const double a = 2.0 * pi * k / NFFT;
        out[k] = make_float2((float)std::cos(a), (float)(-std::sin(a)));
    }
}

}  // namespace kws

using namespace kws;

static int frames_for(int n_samples, int frame_len, int frame_step) {
    if (n_samples <= frame_len) return 1;
    return 1 + (n_samples - frame_len + frame_step - 1) / frame_step;  // 1 + ceil((n - L)/step)
}

// Bracket a kernel launch with events when profiling is on.
struct ProfScope {
    kws_ctx* c;
    int id;
    hipEvent_t stop = nullptr;
    ProfScope(kws_ctx* c_, int id_) : c(c_), id(id_) {
        if (!c->prof) return;
        if (c->prof_seen[id]++ % (unsigned)c->prof_every != 0) return;  // sampling: events around every launch cost ~7 us of stream time each
        if (c->ev_used[id] == c->ev[id].size()) {
            kws_ctx::EvPair p{};
            if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return;
            c->ev[id].push_back(p);
        }
        kws_ctx::EvPair& p = c->ev[id][c->ev_used[id]++];
        (void)hipEventRecord(p.a, c->stream);
        stop = p.b;
    }
    ~ProfScope() {
        if (stop) (void)hipEventRecord(stop, c->stream);
    }
};

static void stream_free_fwd(kws_ctx* c);
static void drop_stream_graph(kws_ctx* c);

#pragma GCC visibility push(default)
extern "C" {

int kws_abi_version(void) { return KWS_ABI_VERSION; }

const char* kws_last_error(kws_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

const char* kws_kernel_name(int id) { return (id >= 0 && id < KWS_K_COUNT) ? kKernelNames[id] : ""; }

int kws_create(kws_ctx** out, int device_id) {
    KWS_GUARD_BEGIN
    if (!out) return fail(nullptr, KWS_EINVAL, "kws_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, KWS_EHIP, std::string("kws_create: no HIP device (") + hipGetErrorString(e) + "); there is no CPU fallback");
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, KWS_EINVAL, "kws_create: device_id out of range");
    kws_ctx* c = new (std::nothrow) kws_ctx();
    if (!c) return fail(nullptr, KWS_ENOMEM, "kws_create: out of host memory");
    c->device = device_id;
    e = hipSetDevice(device_id);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = dscnn_init_device();
    if (e == hipSuccess) e = cnntrad_init_device();
    if (e != hipSuccess) {
        int rc = fail_hip(nullptr, e, "kws_create");
        if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
        delete c;
        return rc;
    }
    c->stream = c->own_stream;
    int rc = kws_set_frontend(c, 16000, 16000, 400, 160, 512, 26, 10, 0.97f, 22);
    if (rc != KWS_OK) {
        g_create_err = c->err;
        kws_destroy(c);
        return rc;
    }
    *out = c;
    return KWS_OK;
    KWS_GUARD_END(nullptr, "kws_create")
}

void kws_destroy(kws_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (int k = 0; k < KWS_K_COUNT; ++k)
        for (auto& p : c->ev[k]) {
            (void)hipEventDestroy(p.a);
            (void)hipEventDestroy(p.b);
        }
    if (c->d_fe) (void)hipFree(c->d_fe);
    if (c->d_spec_tw64) (void)hipFree(c->d_spec_tw64);
    if (c->d_refine) (void)hipFree(c->d_refine);
    if (c->d_model) (void)hipFree(c->d_model);
    if (c->d_cnntrad) (void)hipFree(c->d_cnntrad);
    if (c->d_conv_ws) (void)hipFree(c->d_conv_ws);
    if (c->d_feat_ws) (void)hipFree(c->d_feat_ws);
    stream_free_fwd(c);  // rings, hop counter, captured graph, smoothing and endpointer history
    ingest_free(c);      // staging rings, copy streams, pack threads
    if (c->order_ev) (void)hipEventDestroy(c->order_ev);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

// The workspaces and tables of a context are shared by everything it enqueues, so work enqueued on the new stream must
// not overtake work still pending on the old one: the new stream waits on an event recorded on the old stream.  (A
// context is still single-threaded and serves one stream at a time; this only makes the hand-over safe.)
int kws_set_stream(kws_ctx* c, void* hip_stream, int external) {
    if (!c) return KWS_EINVAL;
    hipStream_t next = external ? (hipStream_t)hip_stream : c->own_stream;
    if (next == c->stream) return KWS_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->order_ev) HIP_TRY(c, hipEventCreateWithFlags(&c->order_ev, hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->order_ev, c->stream));
    HIP_TRY(c, hipStreamWaitEvent(next, c->order_ev, 0));
    if (c->stream_graph) {  // a captured push replays on the stream it was captured for: retire it once it is idle
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        drop_stream_graph(c);
    }
    c->stream = next;
    return KWS_OK;
}

int kws_sync(kws_ctx* c) {
    if (!c) return KWS_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return KWS_OK;
}

int kws_set_frontend(kws_ctx* c, int sample_rate, int n_samples, int frame_len, int frame_step, int nfft, int nfilt,
                     int numcep, float preemph, int ceplifter) {
    KWS_GUARD_BEGIN
    if (!c) return KWS_EINVAL;
    if (sample_rate <= 0 || n_samples <= 0 || frame_len <= 0 || frame_step <= 0 || nfilt <= 0 || numcep <= 0 || nfft < 2)
        return fail(c, KWS_EINVAL, "kws_set_frontend: sizes must be positive");
    if (nfilt > MAX_NFILT || numcep > MAX_NUMCEP || numcep > nfilt)
        return fail(c, KWS_EUNSUPPORTED, "kws_set_frontend: need nfilt <= 64 and numcep <= min(nfilt, 32)");
    int log2n = 0;
    if ((nfft & (nfft - 1)) == 0)
        for (int v = nfft; v > 1; v >>= 1) ++log2n;
    if (nfft > 4096 || (log2n == 0 && nfft > 2048) || (log2n > 0 && nfft < 64))
        return fail(c, KWS_EUNSUPPORTED, "kws_set_frontend: nfft must be a power of two in [64, 4096] or any value in [2, 2048]");
    std::vector<int> edges;
    mel_edges(nfilt, nfft, sample_rate, edges);
    for (size_t i = 0; i + 1 < edges.size(); ++i)
        if (edges[i + 1] < edges[i] || edges[i] < 0 || edges[i + 1] > nfft / 2)
            return fail(c, KWS_EUNSUPPORTED, "kws_set_frontend: mel edges are not monotone inside [0, nfft/2]");
    // The float32 kernel is built for nfft = 512, frames of at most 512 samples and filterbanks its sparse lane layout can
    // hold; every other geometry runs on the float64 kernel (kws_mfcc_f64.hip).
    MelHost mel;
    std::string err;
    bool fast = nfft == NFFT && frame_len <= NFFT && build_mel_host(nfilt, nfft, sample_rate, mel, err) &&
                mel.edges.front() == 0 && mel.edges.back() == nfft / 2;
    if (!fast) {
        mel = MelHost();
        mel.k0.assign(64, 0);
        mel.rw.assign(MEL_CHUNK * 64, 0.f);
        mel.fw.assign(MEL_CHUNK * 64, 0.f);
        mel.gather.assign(64, 0u);
        mel.slot.assign(NFFT / 2, 0);
        mel.seg.assign(64, 0);
    }
    std::vector<float> dct;
    build_dct_lifter_host(nfilt, numcep, ceplifter, dct);
    std::vector<float2> tw;
    build_twiddle_host(tw);
    // float64 tables: twiddles (cos, -sin)(2 pi k / nfft), DCT-II(ortho) x lifter
    const double pi = 3.14159265358979323846;
    std::vector<double> tw64(2 * (size_t)nfft), dct64((size_t)numcep * nfilt);
    for (int k = 0; k < nfft; ++k) {
        const double a = 2.0 * pi * k / nfft;
        tw64[2 * k] = std::cos(a);
        tw64[2 * k + 1] = -std::sin(a);
    }
    for (int i = 0; i < numcep; ++i) {
        const double lift = ceplifter > 0 ? 1.0 + (ceplifter / 2.0) * std::sin(pi * i / ceplifter) : 1.0;
        for (int j = 0; j < nfilt; ++j)
            dct64[(size_t)i * nfilt + j] = lift * (i == 0 ? std::sqrt(1.0 / nfilt) : std::sqrt(2.0 / nfilt) * std::cos(pi * i * (2 * j + 1) / (2.0 * nfilt)));
    }

    // per-bin mel weights exactly as psf's get_filterbanks forms them (float64 divisions): bin i in [e_j, e_j+1) rises in
    // filter j with (i - e_j)/(e_j+1 - e_j) and falls in filter j-1 with (e_j+1 - i)/(e_j+1 - e_j)
    const int nb64 = nfft / 2 + 1;
    std::vector<double> melw(2 * (size_t)nb64, 0.0);
    for (int j = 0; j <= nfilt; ++j)
        for (int i = edges[j]; i < edges[j + 1]; ++i) {
            const double width = (double)(edges[j + 1] - edges[j]);
            melw[i] = (double)(i - edges[j]) / width;
            melw[(size_t)nb64 + i] = (double)(edges[j + 1] - i) / width;
        }

    const int nfp = (nfilt + 3) & ~3;
    std::vector<float> dct_pad(((size_t)numcep * nfp + 3) & ~(size_t)3, 0.f);
    for (int i = 0; i < numcep; ++i)
        for (int j = 0; j < nfilt; ++j) dct_pad[(size_t)i * nfp + j] = dct[(size_t)i * nfilt + j];

    // one device allocation, every table 256-byte aligned
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_tw = 0, o_k0 = al(o_tw + sizeof(float2) * NFFT), o_rw = al(o_k0 + sizeof(int) * 64),
                 o_fw = al(o_rw + sizeof(float) * MEL_CHUNK * 64), o_g = al(o_fw + sizeof(float) * MEL_CHUNK * 64),
                 o_dct = al(o_g + sizeof(uint32_t) * 64), o_slot = al(o_dct + sizeof(float) * dct.size()),
                 o_seg = al(o_slot + sizeof(int) * (NFFT / 2)), o_tw64 = al(o_seg + sizeof(int) * 64),
                 o_edges = al(o_tw64 + sizeof(double) * tw64.size()), o_dct64 = al(o_edges + sizeof(int) * edges.size()),
                 o_melw = al(o_dct64 + sizeof(double) * dct64.size()), o_dctp = al(o_melw + sizeof(double) * melw.size()),
                 total = al(o_dctp + sizeof(float) * dct_pad.size());
    std::vector<unsigned char> host(total, 0);
    memcpy(&host[o_tw], tw.data(), sizeof(float2) * NFFT);
    memcpy(&host[o_k0], mel.k0.data(), sizeof(int) * 64);
    memcpy(&host[o_rw], mel.rw.data(), sizeof(float) * MEL_CHUNK * 64);
    memcpy(&host[o_fw], mel.fw.data(), sizeof(float) * MEL_CHUNK * 64);
    memcpy(&host[o_g], mel.gather.data(), sizeof(uint32_t) * 64);
    memcpy(&host[o_dct], dct.data(), sizeof(float) * dct.size());
    memcpy(&host[o_slot], mel.slot.data(), sizeof(int) * (NFFT / 2));
    memcpy(&host[o_seg], mel.seg.data(), sizeof(int) * 64);
    memcpy(&host[o_tw64], tw64.data(), sizeof(double) * tw64.size());
    memcpy(&host[o_edges], edges.data(), sizeof(int) * edges.size());
    memcpy(&host[o_dct64], dct64.data(), sizeof(double) * dct64.size());
    memcpy(&host[o_melw], melw.data(), sizeof(double) * melw.size());
    memcpy(&host[o_dctp], dct_pad.data(), sizeof(float) * dct_pad.size());

    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));  // tables of the previous configuration may be in use
    if (c->n_streams) stream_free_fwd(c);         // ring geometry depends on the front end
    void* d = nullptr;
    if (hipMalloc(&d, total) != hipSuccess) return fail(c, KWS_ENOMEM, "kws_set_frontend: device allocation failed");
    hipError_t e = hipMemcpy(d, host.data(), total, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(d);
        return fail_hip(c, e, "kws_set_frontend: hipMemcpy");
    }
    if (c->d_fe) (void)hipFree(c->d_fe);
    c->d_fe = d;
    unsigned char* b = static_cast<unsigned char*>(d);
    c->ft.twiddle = reinterpret_cast<const float2*>(b + o_tw);
    c->ft.mel_k0 = reinterpret_cast<const int*>(b + o_k0);
    c->ft.mel_rw = reinterpret_cast<const float*>(b + o_rw);
    c->ft.mel_fw = reinterpret_cast<const float*>(b + o_fw);
    c->ft.mel_gather = reinterpret_cast<const uint32_t*>(b + o_g);
    c->ft.dct = reinterpret_cast<const float*>(b + o_dct);
    c->ft.mel_slot = reinterpret_cast<const int*>(b + o_slot);
    c->ft.mel_seg = reinterpret_cast<const int*>(b + o_seg);
    c->ft.tw64 = reinterpret_cast<const double*>(b + o_tw64);
    c->ft.mel_edges = reinterpret_cast<const int*>(b + o_edges);
    c->ft.dct64 = reinterpret_cast<const double*>(b + o_dct64);
    c->ft.mel_w64 = reinterpret_cast<const double*>(b + o_melw);
    c->ft.dct_pad = reinterpret_cast<const float*>(b + o_dctp);

    FrontendParams& p = c->fp;
    p.n_samples = n_samples;
    p.frame_len = frame_len;
    p.frame_step = frame_step;
    p.num_frames = frames_for(n_samples, frame_len, frame_step);
    p.nfilt = nfilt;
    p.numcep = numcep;
    p.append_energy = 1;
    p.preemph = preemph;
    p.chunk_samples = (MFCC_FRAMES_PER_WG - 1) * frame_step + frame_len;
    // 16-byte PCM loads: every clip base and every workgroup's first sample must be multiples of 8 samples
    // (the pointer itself is checked per call; the tail of a clip falls back to guarded scalar loads)
    p.vec_ok = (n_samples % 8 == 0) && ((MFCC_FRAMES_PER_WG * frame_step) % 8 == 0);
    p.nfft = nfft;
    p.log2_nfft = log2n;
    p.refine_span = c->refine_span;
    c->sample_rate = sample_rate;
    c->nfft = nfft;
    c->ceplifter = ceplifter;
    c->fe_fast_ok = fast;
    c->fe_ready = true;
    return KWS_OK;
    KWS_GUARD_END(c, "kws_set_frontend")
}

int kws_set_frontend_math(kws_ctx* c, int math) {
    if (!c) return KWS_EINVAL;
    if (math != KWS_FE_F32 && math != KWS_FE_F64) return fail(c, KWS_EINVAL, "kws_set_frontend_math: math must be KWS_FE_F32 or KWS_FE_F64");
    c->fe_math = math;
    return KWS_OK;
}

int kws_frontend_math(kws_ctx* c) {
    if (!c || !c->fe_ready) return KWS_EINVAL;
    return (c->fe_math == KWS_FE_F64 || !c->fe_fast_ok) ? KWS_FE_F64 : KWS_FE_F32;
}

int kws_set_frontend_refine(kws_ctx* c, float log_span) {
    if (!c) return KWS_EINVAL;
    if (!(log_span == log_span)) return fail(c, KWS_EINVAL, "kws_set_frontend_refine: log_span is NaN");
    c->refine_span = log_span > 0.f ? log_span : 0.f;
    c->fp.refine_span = c->refine_span;
    if (c->stream_graph) {  // a captured push holds the front-end parameters by value
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        drop_stream_graph(c);
    }
    return KWS_OK;
}

int kws_frontend_stats(kws_ctx* c, uint64_t* frames_total, uint64_t* frames_refined, int* last_call_refined) {
    if (!c) return KWS_EINVAL;
    int ctr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->d_refine) HIP_TRY(c, hipMemcpy(ctr, c->d_refine, sizeof ctr, hipMemcpyDeviceToHost));
    if (frames_total) *frames_total = c->frames_seen;
    if (frames_refined) *frames_refined = ((uint64_t)(uint32_t)ctr[3] << 32 | (uint32_t)ctr[2]) + (uint64_t)(uint32_t)ctr[5];
    if (last_call_refined) *last_call_refined = ctr[4];
    return KWS_OK;
}

int kws_frontend_shape(kws_ctx* c, int* num_frames, int* numcep) {
    if (!c) return KWS_EINVAL;
    if (!c->fe_ready) return fail(c, KWS_ESTATE, "front end not configured");
    if (num_frames) *num_frames = c->fp.num_frames;
    if (numcep) *numcep = c->fp.numcep;
    return KWS_OK;
}

int kws_load_dscnn(kws_ctx* c, const float* blob, size_t n_floats, int num_classes) {
    return kws_load_dscnn_ex(c, blob, n_floats, num_classes, 1);
}

static float pow2_weight_scale(const float* w, size_t n);
static float max_row_abs_sum(const float* w, int rows, size_t row_len);
// PLAIN f16-pair pieces of one weight already multiplied by its layer's scale: hi = f16(x), lo = f16(x - hi) (kws_split_mfma.h)
static void pair_plain(float x, uint16_t& hb, uint16_t& lb) {
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)(x - (float)h);
    memcpy(&hb, &h, 2);
    memcpy(&lb, &l, 2);
}

int kws_load_dscnn_ex(kws_ctx* c, const float* blob, size_t n_floats, int num_classes, int input_channels) {
    KWS_GUARD_BEGIN
    if (!c) return KWS_EINVAL;
    if (!blob) return fail(c, KWS_EINVAL, "kws_load_dscnn: blob is NULL");
    if (num_classes < 1 || num_classes > MAX_CLASSES) return fail(c, KWS_EUNSUPPORTED, "kws_load_dscnn: num_classes must be in [1, 64]");
    if (input_channels < 1 || input_channels > 64) return fail(c, KWS_EUNSUPPORTED, "kws_load_dscnn: input_channels must be in [1, 64]");
    const size_t c1_floats = (size_t)6400 * input_channels;
    const size_t expect = c1_floats + 64 + 4 * (576 + 64 + 4096 + 64) + (size_t)num_classes * 64 + num_classes;
    if (n_floats != expect) {
        char msg[200];
        snprintf(msg, sizeof msg, "kws_load_dscnn: expected %zu floats for %d classes and %d input channel(s), got %zu", expect,
                 num_classes, input_channels, n_floats);
        return fail(c, KWS_EINVAL, msg);
    }
    // repack: c1_w [100][64] | c1_b [64] | dw [4][64][12] | pw_w [4][cin][cout] | pw_b [4][64] | fc_w | fc_b | splits |
    // (input_channels > 1) conv1 as [ci][tap][cout] for kws_conv1_general_kernel
    const size_t o_c1w = 0, o_c1b = o_c1w + 6400, o_dw = o_c1b + 64, o_pww = o_dw + 4 * 64 * 12, o_pwb = o_pww + 4 * 4096,
                 o_fcw = o_pwb + 4 * 64, o_fcb = o_fcw + (size_t)num_classes * 64,
                 o_split = (o_fcb + num_classes + 3) & ~(size_t)3, o_c1s = o_split + 4 * 2 * 4 * 3 * 64 * 4,
                 o_c1g = o_c1s + 2 * 7 * 3 * 64 * 4, o_raw = o_c1g + c1_floats,
                 o_pwp = (o_raw + n_floats + 3) & ~(size_t)3, o_c1p = o_pwp + 4 * 2 * 4 * 2 * 64 * 4, total = o_c1p + 2 * 7 * 2 * 64 * 4;
    std::vector<float> h(total, 0.f);
    const float* src = blob;
    memcpy(&h[o_raw], blob, n_floats * sizeof(float));  // torch layouts, for the composed any-map path (kws_forward_map_f32)
    // conv1.weight [64][C][10][10] -> [ci][tap][cout] (kws_conv1_general_kernel for C > 1, kws_conv1_any_kernel for any map)
    {
        for (int co = 0; co < 64; ++co)
            for (int ci = 0; ci < input_channels; ++ci)
                for (int k = 0; k < 100; ++k) h[o_c1g + ((size_t)ci * 100 + k) * 64 + co] = src[((size_t)co * input_channels + ci) * 100 + k];
    }
    for (int co = 0; co < 64 && input_channels == 1; ++co)  // conv1.weight [64][1][10][10] -> [k][cout]
        for (int k = 0; k < 100; ++k) h[o_c1w + (size_t)k * 64 + co] = src[co * 100 + k];
    if (input_channels == 1) {
        // conv1 as bf16x3 MFMA A operands (32x32x16).  The 100 taps are split between the half-waves: lanes
        // 32..63 take kernel rows 5..9, so both halves walk the same 50 (+6 zero) offsets f = 10*(kh%5) + kw and
        // their LDS addresses differ by a constant.  Lane l of (ct, kb): cout = 32ct + (l&31), f = 8kb + j.
        uint32_t* sp = reinterpret_cast<uint32_t*>(&h[o_c1s]);
        for (int ct = 0; ct < 2; ++ct)
            for (int kb = 0; kb < 7; ++kb)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 8; ++j) {
                        const int co = 32 * ct + (l & 31), f = 8 * kb + j;
                        float r = f < 50 ? src[co * 100 + (f / 10 + 5 * (l >> 5)) * 10 + f % 10] : 0.f;
                        for (int p = 0; p < 3; ++p) {
                            uint32_t u;
                            memcpy(&u, &r, 4);
                            u &= 0xffff0000u;
                            float t;
                            memcpy(&t, &u, 4);
                            r -= t;
                            sp[(((size_t)(ct * 7 + kb) * 3 + p) * 64 + l) * 4 + (j >> 1)] |= (u >> 16) << (16 * (j & 1));
                        }
                    }
    }
    // the f16-pair images and the bounds behind the per-clip activation scales (kws_dscnn.hip, KWS_PW_PAIR_F16)
    DscnnWeights mw{};  // (a local: the context keeps its old model if the upload below fails)
    if (input_channels == 1) {
        const float sw = pow2_weight_scale(src, 6400);
        int ke;
        (void)std::frexp(sw, &ke);
        mw.k_c1 = ke - 1;  // sw = 2^(ke - 1)
        uint32_t* sp = reinterpret_cast<uint32_t*>(&h[o_c1p]);
        for (int ct = 0; ct < 2; ++ct)
            for (int kb = 0; kb < 7; ++kb)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 8; ++j) {
                        // K order of the pre-split windows (kws_dscnn.hip, conv1_unit_pairwin): half-wave h = l >> 5 takes kernel
                        // rows 5h .. 5h + 4; kb < 5: kernel row 5h + kb, taps kw = j; kb = 5: taps kw = 8 + (j & 1) of kernel row
                        // 5h + (j >> 1); kb = 6: taps kw = 8 + j (j < 2) of kernel row 5h + 4, then zeros
                        const int co = 32 * ct + (l & 31), h5 = 5 * (l >> 5);
                        int kh = -1, kw = 0;
                        if (kb < 5) {
                            kh = h5 + kb;
                            kw = j;
                        } else if (kb == 5) {
                            kh = h5 + (j >> 1);
                            kw = 8 + (j & 1);
                        } else if (j < 2) {
                            kh = h5 + 4;
                            kw = 8 + j;
                        }
                        const float v = kh >= 0 ? src[co * 100 + kh * 10 + kw] : 0.f;
                        uint16_t hb, lb;
                        pair_plain(v * sw, hb, lb);
                        sp[(((size_t)(ct * 7 + kb) * 2 + 0) * 64 + l) * 4 + (j >> 1)] |= (uint32_t)hb << (16 * (j & 1));
                        sp[(((size_t)(ct * 7 + kb) * 2 + 1) * 64 + l) * 4 + (j >> 1)] |= (uint32_t)lb << (16 * (j & 1));
                    }
        mw.c1_abs = max_row_abs_sum(src, 64, 100);
        mw.c1_bmax = 0.f;
        for (int i = 0; i < 64; ++i) mw.c1_bmax = std::max(mw.c1_bmax, std::fabs(src[6400 + i]));
    } else {
        mw.k_c1 = 0;
        mw.c1_abs = mw.c1_bmax = 0.f;
    }
    src += c1_floats;
    memcpy(&h[o_c1b], src, 64 * sizeof(float));
    src += 64;
    for (int b = 0; b < 4; ++b) {
        const float *dw_w = src, *dw_b = src + 576, *pw_w = src + 640, *pw_b = src + 640 + 4096;
        for (int ch = 0; ch < 64; ++ch) {  // channel PAIRS interleaved, 24 floats per pair: (tap t of ch, of ch + 1) at 2t, the biases at 18, 19
            float* q = &h[o_dw + ((size_t)b * 32 + ch / 2) * 24 + (ch & 1)];
            for (int t = 0; t < 9; ++t) q[2 * t] = dw_w[ch * 9 + t];
            q[18] = dw_b[ch];
        }
        for (int co = 0; co < 64; ++co)  // pointwise.weight [cout][cin][1][1] -> [cin][cout]
            for (int ci = 0; ci < 64; ++ci) h[o_pww + (size_t)b * 4096 + (size_t)ci * 64 + co] = pw_w[co * 64 + ci];
        memcpy(&h[o_pwb + (size_t)b * 64], pw_b, 64 * sizeof(float));
        // the same weights as bf16x3 MFMA A operands (32x32x16): lane l of (ct, m) holds cin = 16m + 8(l>>5) + j,
        // j = 0..7, of cout = 32ct + (l&31); piece 0/1/2 = top 16 bits of the value / first / second remainder
        uint32_t* sp = reinterpret_cast<uint32_t*>(&h[o_split]) + (size_t)b * (2 * 4 * 3 * 64 * 4);
        for (int ct = 0; ct < 2; ++ct)
            for (int m = 0; m < 4; ++m)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 8; ++j) {
                        const int co = 32 * ct + (l & 31), ci = 16 * m + 8 * (l >> 5) + j;
                        float r = pw_w[co * 64 + ci];
                        for (int p = 0; p < 3; ++p) {
                            uint32_t u;
                            memcpy(&u, &r, 4);
                            u &= 0xffff0000u;
                            float t;
                            memcpy(&t, &u, 4);
                            r -= t;
                            uint32_t& dst = sp[(((size_t)(ct * 4 + m) * 3 + p) * 64 + l) * 4 + (j >> 1)];
                            dst |= (u >> 16) << (16 * (j & 1));
                        }
                    }
        {
            const float sw = pow2_weight_scale(pw_w, 4096);
            int ke;
            (void)std::frexp(sw, &ke);
            mw.k_pw[b] = ke - 1;
            uint32_t* pp = reinterpret_cast<uint32_t*>(&h[o_pwp]) + (size_t)b * (2 * 4 * 2 * 64 * 4);
            for (int ct = 0; ct < 2; ++ct)
                for (int m = 0; m < 4; ++m)
                    for (int l = 0; l < 64; ++l)
                        for (int j = 0; j < 8; ++j) {
                            const int co = 32 * ct + (l & 31), ci = 16 * m + 8 * (l >> 5) + j;
                            uint16_t hb, lb;
                            pair_plain(pw_w[co * 64 + ci] * sw, hb, lb);
                            pp[(((size_t)(ct * 4 + m) * 2 + 0) * 64 + l) * 4 + (j >> 1)] |= (uint32_t)hb << (16 * (j & 1));
                            pp[(((size_t)(ct * 4 + m) * 2 + 1) * 64 + l) * 4 + (j >> 1)] |= (uint32_t)lb << (16 * (j & 1));
                        }
            mw.dw_abs[b] = max_row_abs_sum(dw_w, 64, 9);
            mw.pw_abs[b] = max_row_abs_sum(pw_w, 64, 64);
            mw.dw_bmax[b] = mw.pw_bmax[b] = 0.f;
            for (int i = 0; i < 64; ++i) {
                mw.dw_bmax[b] = std::max(mw.dw_bmax[b], std::fabs(dw_b[i]));
                mw.pw_bmax[b] = std::max(mw.pw_bmax[b], std::fabs(pw_b[i]));
            }
        }
        src += 576 + 64 + 4096 + 64;
    }
    memcpy(&h[o_fcw], src, (size_t)num_classes * 64 * sizeof(float));
    src += (size_t)num_classes * 64;
    memcpy(&h[o_fcb], src, (size_t)num_classes * sizeof(float));

    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    float* d = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d), total * sizeof(float)) != hipSuccess)
        return fail(c, KWS_ENOMEM, "kws_load_dscnn: device allocation failed");
    hipError_t e = hipMemcpy(d, h.data(), total * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(d);
        return fail_hip(c, e, "kws_load_dscnn: hipMemcpy");
    }
    drop_stream_graph(c);  // a captured push holds the old weight pointers by value
    if (c->d_model) (void)hipFree(c->d_model);
    c->d_model = d;
    c->mw.c1_w = d + o_c1w;
    c->mw.c1_b = d + o_c1b;
    c->mw.dw_w = d + o_dw;
    c->mw.pw_w = d + o_pww;
    c->mw.pw_b = d + o_pwb;
    c->mw.pw_split = reinterpret_cast<const uint32_t*>(d + o_split);
    c->mw.c1_split = reinterpret_cast<const uint32_t*>(d + o_c1s);
    c->mw.pw_pair = reinterpret_cast<const uint32_t*>(d + o_pwp);
    c->mw.c1_pair = reinterpret_cast<const uint32_t*>(d + o_c1p);
    c->mw.k_c1 = mw.k_c1;
    c->mw.c1_abs = mw.c1_abs;
    c->mw.c1_bmax = mw.c1_bmax;
    for (int b = 0; b < 4; ++b) {
        c->mw.k_pw[b] = mw.k_pw[b];
        c->mw.dw_abs[b] = mw.dw_abs[b];
        c->mw.dw_bmax[b] = mw.dw_bmax[b];
        c->mw.pw_abs[b] = mw.pw_abs[b];
        c->mw.pw_bmax[b] = mw.pw_bmax[b];
    }
    c->mw.fc_w = d + o_fcw;
    c->mw.fc_b = d + o_fcb;
    c->mw.num_classes = num_classes;
    c->mw.in_channels = input_channels;
    c->mw.c1_general = d + o_c1g;
    c->mw.raw = d + o_raw;
    c->model_ready = true;
    return KWS_OK;
    KWS_GUARD_END(c, "kws_load_dscnn")
}

// Worklist of the selective refinement for batches of up to B clips: int[8] counters + one entry per frame.  The counters
// (running totals included) move to the new allocation.
static int ensure_refine(kws_ctx* c, int B) {
    const size_t frames = (size_t)B * ((c->fp.num_frames + 1) / 2);  // one entry per frame pair
    if (frames > 0x1fffffffu) return fail(c, KWS_EUNSUPPORTED, "refinement worklist: more than 2^29 frame pairs in one call");
    if (c->d_refine && (int)frames <= c->refine_cap) return KWS_OK;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    int* d = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d), sizeof(int) * (8 + frames)) != hipSuccess)
        return fail(c, KWS_ENOMEM, "refinement worklist: device allocation failed");
    hipError_t e = c->d_refine ? hipMemcpy(d, c->d_refine, sizeof(int) * 8, hipMemcpyDeviceToDevice) : hipMemset(d, 0, sizeof(int) * 8);
    if (e != hipSuccess) {
        (void)hipFree(d);
        return fail_hip(c, e, "refinement worklist");
    }
    if (c->d_refine) (void)hipFree(c->d_refine);
    c->d_refine = d;
    c->refine_cap = (int)frames;
    return KWS_OK;
}

int kws_reserve(kws_ctx* c, int max_batch) {
    if (!c) return KWS_EINVAL;
    if (max_batch <= 0) return fail(c, KWS_EINVAL, "kws_reserve: max_batch must be positive");
    if (!c->fe_ready) return fail(c, KWS_ESTATE, "front end not configured");
    if (c->refine_span > 0.f && c->fe_fast_ok) {
        int rc = ensure_refine(c, max_batch);
        if (rc) return rc;
    }
    const size_t need = (size_t)max_batch * c->fp.num_frames * c->fp.numcep;
    if (need <= c->feat_ws_floats) return KWS_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    float* d = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d), need * sizeof(float)) != hipSuccess)
        return fail(c, KWS_ENOMEM, "kws_reserve: device allocation failed");
    if (c->d_feat_ws) (void)hipFree(c->d_feat_ws);
    c->d_feat_ws = d;
    c->feat_ws_floats = need;
    return KWS_OK;
}

static int check_batch(kws_ctx* c, const void* in, int B, const char* fn) {
    if (!c) return KWS_EINVAL;
    if (!in) return fail(c, KWS_EINVAL, std::string(fn) + ": input pointer is NULL");
    if (B <= 0) return fail(c, KWS_EINVAL, std::string(fn) + ": B must be positive");
    return KWS_OK;
}

int kws_mfcc_i16(kws_ctx* c, const int16_t* d_wav, int B, float* d_out) {
    int rc = check_batch(c, d_wav, B, "kws_mfcc_i16");
    if (rc) return rc;
    if (!d_out) return fail(c, KWS_EINVAL, "kws_mfcc_i16: d_out is NULL");
    if (!c->fe_ready) return fail(c, KWS_ESTATE, "kws_mfcc_i16: front end not configured");
    HIP_TRY(c, hipSetDevice(c->device));
    FrontendParams p = c->fp;
    if ((reinterpret_cast<uintptr_t>(d_wav) & 15) != 0) p.vec_ok = 0;
    if (c->fe_math == KWS_FE_F64 || !c->fe_fast_ok) {
        ProfScope ps(c, KWS_K_MFCC_F64);
        HIP_TRY(c, launch_mfcc_f64(c->stream, p, c->ft, d_wav, B, d_out));
        return KWS_OK;
    }
    c->frames_seen += (unsigned long long)B * p.num_frames;
    if (p.refine_span > 0.f) {
        rc = ensure_refine(c, B);
        if (rc) return rc;
        const RefineList rl = {c->d_refine, c->d_refine + 8, c->refine_cap, 0};
        {
            ProfScope ps(c, KWS_K_MFCC);
            HIP_TRY(c, launch_mfcc_flag(c->stream, p, c->ft, d_wav, B, d_out, rl));
        }
        ProfScope ps(c, KWS_K_MFCC_REFINE);
        HIP_TRY(c, launch_mfcc_refine(c->stream, p, c->ft, d_wav, d_out, rl, B));
        return KWS_OK;
    }
    ProfScope ps(c, KWS_K_MFCC);
    HIP_TRY(c, launch_mfcc(c->stream, p, c->ft, d_wav, B, d_out));
    return KWS_OK;
}

int kws_mfcc_f32(kws_ctx* c, const float* d_wav, int B, float* d_out) {
    int rc = check_batch(c, d_wav, B, "kws_mfcc_f32");
    if (rc) return rc;
    if (!d_out) return fail(c, KWS_EINVAL, "kws_mfcc_f32: d_out is NULL");
    if (!c->fe_ready) return fail(c, KWS_ESTATE, "kws_mfcc_f32: front end not configured");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->fe_math == KWS_FE_F64 || !c->fe_fast_ok) {
        ProfScope ps(c, KWS_K_MFCC_F64);
        HIP_TRY(c, launch_mfcc_f64_f32in(c->stream, c->fp, c->ft, d_wav, B, d_out));
        return KWS_OK;
    }
    c->frames_seen += (unsigned long long)B * c->fp.num_frames;
    if (c->fp.refine_span > 0.f) {
        rc = ensure_refine(c, B);
        if (rc) return rc;
        const RefineList rl = {c->d_refine, c->d_refine + 8, c->refine_cap, 0};
        {
            ProfScope ps(c, KWS_K_MFCC);
            HIP_TRY(c, launch_mfcc_f32_flag(c->stream, c->fp, c->ft, d_wav, B, d_out, rl));
        }
        ProfScope ps(c, KWS_K_MFCC_REFINE);
        HIP_TRY(c, launch_mfcc_refine_f32in(c->stream, c->fp, c->ft, d_wav, d_out, rl, B));
        return KWS_OK;
    }
    ProfScope ps(c, KWS_K_MFCC);
    HIP_TRY(c, launch_mfcc_f32(c->stream, c->fp, c->ft, d_wav, B, d_out));
    return KWS_OK;
}

static int grow_conv_ws(kws_ctx* c, size_t need, const std::string& fn);

static int forward_impl(kws_ctx* c, const float* d_feat, int B, float* d_logits, int32_t* d_label, float* d_act,
                        int mode, const char* fn, unsigned long long* d_stamps = nullptr) {
    int rc = check_batch(c, d_feat, B, fn);
    if (rc) return rc;
    if (!d_logits) return fail(c, KWS_EINVAL, std::string(fn) + ": d_logits is NULL");
    if (!c->model_ready) return fail(c, KWS_ESTATE, std::string(fn) + ": no model loaded (kws_load_dscnn)");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->mw.in_channels > 1) {
        // multi-channel input (models.py:125,135): conv1 in its own kernel through the context scratch, then the fused
        // kernel from block 1 on; the diagnostics variants exist for the single-channel model only
        if (d_act || d_stamps || (mode != KWS_PW_SPLIT_BF16 && mode != KWS_PW_PAIR_F16))
            return fail(c, KWS_EUNSUPPORTED, std::string(fn) + ": input_channels > 1 runs on the product kernels only");
        rc = grow_conv_ws(c, (size_t)B * 64 * 141, fn);
        if (rc) return rc;
        HIP_TRY(c, launch_conv1_general(c->stream, d_feat, B, c->mw.in_channels, c->mw.c1_general, c->mw.c1_b, c->d_conv_ws));
        ProfScope ps(c, KWS_K_DSCNN);
        HIP_TRY(c, launch_dscnn(c->stream, c->mw, c->d_conv_ws, B, d_logits, d_label, nullptr, mode, nullptr, nullptr, true));
        return KWS_OK;
    }
    ProfScope ps(c, KWS_K_DSCNN);
    HIP_TRY(c, launch_dscnn(c->stream, c->mw, d_feat, B, d_logits, d_label, d_act, mode, d_stamps));
    return KWS_OK;
}

int kws_forward_f32(kws_ctx* c, const float* d_feat, int B, float* d_logits, int32_t* d_label) {
    return forward_impl(c, d_feat, B, d_logits, d_label, nullptr, c ? c->pw_math : 0, "kws_forward_f32");
}

// DepthwiseSeparableConv.forward on a feature map of any size (models.py:160-183): T x F == 99 x 10 takes the fused
// LDS-resident kernel, anything else runs composed through HBM -- conv1 -> four depthwise-separable blocks (the standalone
// block kernels, each adding its relu(bias) ring) -> global average pool + fc + argmax.
static int forward_map_impl(kws_ctx* c, const float* d_feat, int B, int T, int F, float* d_logits, int32_t* d_label, float* d_layers,
                            const char* fn) {
    int rc = check_batch(c, d_feat, B, fn);
    if (rc) return rc;
    if (!d_logits) return fail(c, KWS_EINVAL, std::string(fn) + ": d_logits is NULL");
    if (!c->model_ready) return fail(c, KWS_ESTATE, std::string(fn) + ": no model loaded (kws_load_dscnn)");
    if (T == IN_T && F == IN_F && !d_layers) return forward_impl(c, d_feat, B, d_logits, d_label, nullptr, c->pw_math, fn);
    if (T < 6 || F < 6) return fail(c, KWS_EINVAL, std::string(fn) + ": the 10 x 10 first convolution (padding 2) needs T >= 6 and F >= 6");
    if ((size_t)(T + 4) * (F + 4) * sizeof(float) > 160 * 1024)
        return fail(c, KWS_EUNSUPPORTED, std::string(fn) + ": the padded feature map must fit 160 KB of LDS ((T + 4) * (F + 4) <= 40960)");
    HIP_TRY(c, hipSetDevice(c->device));
    const int H1 = (T - 6) / 2 + 1, W1 = (F - 6) / 2 + 1, Cin = c->mw.in_channels, C = c->mw.num_classes;
    const size_t per_clip_max = (size_t)CH * (H1 + 8) * (W1 + 8);   // block 4's output, ring included
    const size_t dw_max = (size_t)CH * (H1 + 6) * (W1 + 6);        // block 4's depthwise output
    const int chunk = B < 16384 ? B : 16384;                        // grid.z of the pointwise kernel, and a bounded workspace
    rc = grow_conv_ws(c, (size_t)chunk * (2 * per_clip_max + dw_max), fn);
    if (rc) return rc;
    float* bufs[2] = {c->d_conv_ws, c->d_conv_ws + (size_t)chunk * per_clip_max};
    float* dw_ws = c->d_conv_ws + 2 * (size_t)chunk * per_clip_max;
    const float* raw = c->mw.raw + (size_t)6400 * Cin + 64;        // first block's parameters in state_dict order
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int nb = B - b0 < chunk ? B - b0 : chunk;
        HIP_TRY(c, launch_conv1_any(c->stream, d_feat + (size_t)b0 * Cin * T * F, nb, Cin, T, F, c->mw.c1_general, c->mw.c1_b, bufs[0]));
        int H = H1, W = W1, cur = 0;
        size_t lo = 0;
        auto dump = [&](const float* src, size_t per_clip) -> hipError_t {  // diagnostics: stage outputs, stage-major, batch inside
            if (!d_layers) return hipSuccess;
            hipError_t e = hipMemcpyAsync(d_layers + lo * B + (size_t)b0 * per_clip, src, sizeof(float) * per_clip * nb, hipMemcpyDeviceToDevice, c->stream);
            lo += per_clip;
            return e;
        };
        HIP_TRY(c, dump(bufs[0], (size_t)CH * H * W));
        for (int blk = 0; blk < N_BLOCKS; ++blk) {
            const float* prm = raw + (size_t)blk * (576 + 64 + 4096 + 64);
            HIP_TRY(c, launch_dsblock(c->stream, bufs[cur], nb, CH, H, W, prm, prm + 576, prm + 640, prm + 640 + 4096, CH, 3, 1, 1, dw_ws,
                                      bufs[cur ^ 1]));
            cur ^= 1;
            H += 2;
            W += 2;
            HIP_TRY(c, dump(bufs[cur], (size_t)CH * H * W));
        }
        HIP_TRY(c, launch_pool_fc(c->stream, bufs[cur], nb, H * W, c->mw.fc_w, c->mw.fc_b, C, d_logits + (size_t)b0 * C,
                                  d_label ? d_label + b0 : nullptr));
    }
    return KWS_OK;
}

int kws_forward_map_f32(kws_ctx* c, const float* d_feat, int B, int T, int F, float* d_logits, int32_t* d_label) {
    return forward_map_impl(c, d_feat, B, T, F, d_logits, d_label, nullptr, "kws_forward_map_f32");
}

int kws_forward_map_debug_f32(kws_ctx* c, const float* d_feat, int B, int T, int F, float* d_logits, int32_t* d_label, float* d_layers) {
    if (c && !d_layers) return fail(c, KWS_EINVAL, "kws_forward_map_debug_f32: d_layers is NULL");
    return forward_map_impl(c, d_feat, B, T, F, d_logits, d_label, d_layers, "kws_forward_map_debug_f32");
}

int kws_set_pointwise_math(kws_ctx* c, int math) {
    if (!c) return KWS_EINVAL;
    if (math != KWS_PW_F32 && math != KWS_PW_SPLIT_BF16 && math != KWS_PW_PAIR_F16)
        return fail(c, KWS_EINVAL, "kws_set_pointwise_math: math must be KWS_PW_F32, KWS_PW_SPLIT_BF16 or KWS_PW_PAIR_F16");
    if (math != c->pw_math && c->stream_graph) {  // the captured graph holds the other kernel
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        drop_stream_graph(c);
    }
    c->pw_math = math;
    return KWS_OK;
}

int kws_forward_debug_f32(kws_ctx* c, const float* d_feat, int B, float* d_logits, int32_t* d_label, float* d_act,
                          int use_mfma) {
    if (c && use_mfma != 0 && use_mfma != KWS_PW_F32 && use_mfma != KWS_PW_SPLIT_BF16 && use_mfma != KWS_PW_PAIR_F16)
        return fail(c, KWS_EINVAL, "kws_forward_debug_f32: use_mfma must be 0, KWS_PW_F32, KWS_PW_SPLIT_BF16 or KWS_PW_PAIR_F16");
    return forward_impl(c, d_feat, B, d_logits, d_label, d_act, use_mfma, "kws_forward_debug_f32");
}

int kws_forward_stamps_f32(kws_ctx* c, const float* d_feat, int B, float* d_logits, uint64_t* d_stamps, int mode) {
    if (!d_stamps) return fail(c, KWS_EINVAL, "kws_forward_stamps_f32: d_stamps is NULL");
    if (mode != 0 && mode != 1 && mode != 2 && mode != 3 && mode != 4 && mode != 5 && mode != 6)
        return fail(c, KWS_EINVAL, "kws_forward_stamps_f32: unknown kernel variant");
    return forward_impl(c, d_feat, B, d_logits, nullptr, nullptr, mode, "kws_forward_stamps_f32",
                        reinterpret_cast<unsigned long long*>(d_stamps));
}

int kws_infer_i16(kws_ctx* c, const int16_t* d_wav, int B, float* d_logits, int32_t* d_label) {
    int rc = check_batch(c, d_wav, B, "kws_infer_i16");
    if (rc) return rc;
    if (!c->fe_ready || !c->model_ready) return fail(c, KWS_ESTATE, "kws_infer_i16: front end or model not configured");
    if (c->mw.in_channels != 1) return fail(c, KWS_EUNSUPPORTED, "kws_infer_i16: the MFCC front end yields one channel; the model was loaded with more");
    rc = kws_reserve(c, B);
    if (rc) return rc;
    rc = kws_mfcc_i16(c, d_wav, B, c->d_feat_ws);
    if (rc) return rc;
    // 99 x 10 (the reference geometry): the fused kernel; any other map kws_frontend_shape yields: the composed path
    return forward_map_impl(c, c->d_feat_ws, B, c->fp.num_frames, c->fp.numcep, d_logits, d_label, nullptr, "kws_infer_i16");
}

int kws_infer_f32(kws_ctx* c, const float* d_wav, int B, float* d_logits, int32_t* d_label) {
    int rc = check_batch(c, d_wav, B, "kws_infer_f32");
    if (rc) return rc;
    if (!c->fe_ready || !c->model_ready) return fail(c, KWS_ESTATE, "kws_infer_f32: front end or model not configured");
    if (c->mw.in_channels != 1) return fail(c, KWS_EUNSUPPORTED, "kws_infer_f32: the MFCC front end yields one channel; the model was loaded with more");
    rc = kws_reserve(c, B);
    if (rc) return rc;
    rc = kws_mfcc_f32(c, d_wav, B, c->d_feat_ws);
    if (rc) return rc;
    return forward_map_impl(c, c->d_feat_ws, B, c->fp.num_frames, c->fp.numcep, d_logits, d_label, nullptr, "kws_infer_f32");
}

// ---- streaming ------------------------------------------------------------------------------------
static void smooth_free(kws_ctx* c) {
    if (c->d_post_ring) (void)hipFree(c->d_post_ring);
    if (c->d_post_sum) (void)hipFree(c->d_post_sum);
    if (c->d_post_count) (void)hipFree(c->d_post_count);
    c->d_post_ring = c->d_post_sum = nullptr;
    c->d_post_count = nullptr;
    c->post_window = c->post_classes = 0;
}

static void vad_free(kws_ctx* c) {
    if (c->d_vad_flags) (void)hipFree(c->d_vad_flags);
    if (c->d_vad_state) (void)hipFree(c->d_vad_state);
    c->d_vad_flags = nullptr;
    c->d_vad_state = nullptr;
    c->vad_on = c->vad_off = 0;
}

static void drop_stream_graph(kws_ctx* c) {
    if (c->stream_graph) (void)hipGraphExecDestroy(c->stream_graph);
    c->stream_graph = nullptr;
    c->graph_key[0] = c->graph_key[1] = c->graph_key[2] = nullptr;
}

static void host_results_free(kws_ctx* c) {
    if (c->h_stream_logits) (void)hipHostFree(c->h_stream_logits);
    if (c->h_stream_label) (void)hipHostFree(c->h_stream_label);
    if (c->h_stream_flag) (void)hipHostFree(c->h_stream_flag);
    if (c->h_stream_hop) (void)hipHostFree(c->h_stream_hop);
    if (c->d_hr_logits) (void)hipFree(c->d_hr_logits);
    if (c->d_hr_label) (void)hipFree(c->d_hr_label);
    c->h_stream_hop = nullptr;
    c->d_hr_logits = nullptr;
    c->d_hr_label = nullptr;
    c->h_stream_logits = nullptr;
    c->h_stream_label = nullptr;
    c->h_stream_flag = nullptr;
    c->host_results_classes = 0;
}

static void stream_free(kws_ctx* c) {
    smooth_free(c);
    vad_free(c);
    host_results_free(c);
    if (c->stream_graph) (void)hipGraphExecDestroy(c->stream_graph);
    if (c->d_pcm_ring) (void)hipFree(c->d_pcm_ring);
    if (c->d_feat_ring) (void)hipFree(c->d_feat_ring);
    if (c->d_hops) (void)hipFree(c->d_hops);
    if (c->d_cl_part) (void)hipFree(c->d_cl_part);
    if (c->d_cl_count) (void)hipFree(c->d_cl_count);
    c->d_cl_part = nullptr;
    c->d_cl_count = nullptr;
    c->stream_graph = nullptr;
    c->d_pcm_ring = nullptr;
    c->d_feat_ring = nullptr;
    c->d_hops = nullptr;
    c->n_streams = 0;
}

int kws_stream_open(kws_ctx* c, int n_streams) {
    if (!c) return KWS_EINVAL;
    if (n_streams <= 0) return fail(c, KWS_EINVAL, "kws_stream_open: n_streams must be positive");
    if (!c->fe_ready) return fail(c, KWS_ESTATE, "kws_stream_open: front end not configured");
    if (!c->fe_fast_ok)
        return fail(c, KWS_EUNSUPPORTED, "kws_stream_open: the streaming frame kernel is built for nfft == 512 and frames of at most 512 samples");
    if (c->fp.frame_step > 512) return fail(c, KWS_EUNSUPPORTED, "kws_stream_open: hops of more than 512 samples are not supported");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    stream_free(c);
    const FrontendParams& p = c->fp;
    c->ring_len = ((p.frame_len + p.frame_step - 1) / p.frame_step + 1) * p.frame_step;
    const size_t pcm_b = sizeof(int16_t) * (size_t)n_streams * c->ring_len;
    const size_t feat_b = sizeof(float) * (size_t)n_streams * p.num_frames * p.numcep;
    if (hipMalloc(reinterpret_cast<void**>(&c->d_pcm_ring), pcm_b) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&c->d_feat_ring), feat_b) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&c->d_hops), 2 * sizeof(int)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&c->d_cl_part), sizeof(float) * (size_t)n_streams * 4 * 64) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&c->d_cl_count), sizeof(int) * (size_t)n_streams) != hipSuccess) {
        stream_free(c);
        return fail(c, KWS_ENOMEM, "kws_stream_open: device allocation failed");
    }
    c->n_streams = n_streams;
    c->pushes_enqueued = 0;
    c->host_push = 0;
    c->last_push_host = false;
    if (c->refine_span > 0.f) {  // the pushes count the frames they redo in float64 in the refinement counters
        int rc = ensure_refine(c, 1);
        if (rc) return rc;
    }
    HIP_TRY(c, hipMemsetAsync(c->d_pcm_ring, 0, pcm_b, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_feat_ring, 0, feat_b, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_hops, 0, 2 * sizeof(int), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_cl_count, 0, sizeof(int) * (size_t)n_streams, c->stream));
    return KWS_OK;
}

int kws_stream_host_results(kws_ctx* c, int enable) {
    if (!c) return KWS_EINVAL;
    if (!c->n_streams) return fail(c, KWS_ESTATE, "kws_stream_host_results: call kws_stream_open first");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    host_results_free(c);
    if (!enable) return KWS_OK;
    if (!c->model_ready) return fail(c, KWS_ESTATE, "kws_stream_host_results: no model loaded (kws_load_dscnn)");
    const int C = c->mw.num_classes;
    const unsigned flags = hipHostMallocMapped | hipHostMallocCoherent;  // fine-grained: device stores are visible to the host as they land
    if (hipHostMalloc(reinterpret_cast<void**>(&c->h_stream_logits), sizeof(float) * (size_t)c->n_streams * C, flags) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&c->h_stream_label), sizeof(int32_t) * (size_t)c->n_streams, flags) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&c->h_stream_flag), 64, flags) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&c->h_stream_hop), sizeof(int16_t) * (size_t)c->n_streams * c->fp.frame_step, flags) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&c->d_hr_logits), sizeof(float) * (size_t)c->n_streams * C) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&c->d_hr_label), sizeof(int32_t) * (size_t)c->n_streams) != hipSuccess) {
        host_results_free(c);
        return fail(c, KWS_ENOMEM, "kws_stream_host_results: pinned host allocation failed");
    }
    memset(c->h_stream_logits, 0, sizeof(float) * (size_t)c->n_streams * C);
    memset(c->h_stream_label, 0, sizeof(int32_t) * (size_t)c->n_streams);
    // the flag holds the device's push count: start it where the device stands
    int hops = 0;
    HIP_TRY(c, hipMemcpy(&hops, c->d_hops, sizeof(int), hipMemcpyDeviceToHost));
    *c->h_stream_flag = hops;
    c->pushes_enqueued = hops;
    c->host_push = hops;
    c->host_results_classes = C;
    return KWS_OK;
}

int kws_stream_wait_host(kws_ctx* c, const float** h_logits, const int32_t** h_label) {
    if (!c) return KWS_EINVAL;
    if (!c->h_stream_flag) return fail(c, KWS_ESTATE, "kws_stream_wait_host: call kws_stream_host_results(ctx, 1) first");
    // spin on the flag the last workgroup of the newest push raises; bounded: after ~2 ms without it, fall back to the stream
    if (c->host_push != c->pushes_enqueued)
        return fail(c, KWS_ESTATE, "kws_stream_wait_host: the newest push did not deliver to host memory (it asked for no logits, or took a multi-launch route)");
    volatile int* flag = c->h_stream_flag;
    const int want = c->host_push;
    bool seen = false;
    const auto t0 = std::chrono::steady_clock::now();
    for (long spin = 1;; ++spin) {
        if (*flag - want >= 0) {
            seen = true;
            break;
        }
        __builtin_ia32_pause();
        if ((spin & 1023) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    }
    if (!seen) {
        HIP_TRY(c, hipSetDevice(c->device));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (*flag - want < 0) return fail(c, KWS_EHIP, "kws_stream_wait_host: the stream drained but the results flag never arrived");
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (h_logits) *h_logits = c->h_stream_logits;
    if (h_label) *h_label = c->h_stream_label;
    return KWS_OK;
}

int kws_stream_push_host_i16(kws_ctx* c, const int16_t* h_hop, const float** h_logits, const int32_t** h_label) {
    if (!c) return KWS_EINVAL;
    if (!c->n_streams) return fail(c, KWS_ESTATE, "kws_stream_push_host_i16: call kws_stream_open first");
    if (!h_hop) return fail(c, KWS_EINVAL, "kws_stream_push_host_i16: h_hop is NULL");
    if (!c->h_stream_flag) {
        int rc = kws_stream_host_results(c, 1);
        if (rc) return rc;
    }
    // the hop goes into pinned, device-mapped memory and the kernel reads it from there (one PCIe read of 320 bytes per stream
    // on the frame wavefront's path) -- no H2D submission in front of the launch.  One slot: this call returns after the
    // kernel's last workgroup has raised the flag, i.e. after every read of it.
    memcpy(c->h_stream_hop, h_hop, sizeof(int16_t) * (size_t)c->n_streams * c->fp.frame_step);
    int rc = kws_stream_push_i16(c, c->h_stream_hop, c->d_hr_logits, c->d_hr_label, 0);
    if (rc) return rc;
    return kws_stream_wait_host(c, h_logits, h_label);
}

int kws_stream_cluster(kws_ctx* c, int workgroups_per_stream) {
    if (!c) return KWS_EINVAL;
    if (workgroups_per_stream != 0 && workgroups_per_stream != 1 && workgroups_per_stream != 2 && workgroups_per_stream != 4)
        return fail(c, KWS_EINVAL, "kws_stream_cluster: workgroups_per_stream must be 0 (automatic), 1, 2 or 4");
    if (workgroups_per_stream != c->stream_cluster && c->stream_graph) {  // the captured graph holds the other launch shape
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        drop_stream_graph(c);
    }
    c->stream_cluster = workgroups_per_stream;
    return KWS_OK;
}

// Exact bf16 hi/mid/lo pieces of eight weights, OR-ed into four dwords per piece (MFMA operand fragment of one lane).
static void pack_split8(const float (&v)[8], uint32_t* hi, uint32_t* mid, uint32_t* lo) {
    uint32_t* dst[3] = {hi, mid, lo};
    for (int j = 0; j < 8; ++j) {
        float r = v[j];
        for (int p = 0; p < 3; ++p) {
            uint32_t u;
            memcpy(&u, &r, 4);
            u &= 0xffff0000u;
            float t;
            memcpy(&t, &u, 4);
            r -= t;
            dst[p][j >> 1] |= (u >> 16) << (16 * (j & 1));
        }
    }
}

// f16-pair pieces of eight weights already multiplied by the layer's power-of-two scale: hi = f16(v) (round to nearest),
// lo' = f16((v - hi) * 2^11); OR-ed into four dwords per piece like pack_split8 (kws_cnntrad.hip, split_pair).
static void pack_pair8(const float (&v)[8], float scale, uint32_t* hi, uint32_t* lo) {
    for (int j = 0; j < 8; ++j) {
        const float x = v[j] * scale;
        const _Float16 h = (_Float16)x;
        const _Float16 l = (_Float16)((x - (float)h) * 2048.f);
        uint16_t hb, lb;
        memcpy(&hb, &h, 2);
        memcpy(&lb, &l, 2);
        hi[j >> 1] |= (uint32_t)hb << (16 * (j & 1));
        lo[j >> 1] |= (uint32_t)lb << (16 * (j & 1));
    }
}
// the power of two s with max|w| * s < 2^15, and the layer's bound terms
static float pow2_weight_scale(const float* w, size_t n) {
    float m = 0.f;
    for (size_t i = 0; i < n; ++i) m = std::max(m, std::fabs(w[i]));
    if (!(m > 0.f) || !std::isfinite(m)) return 1.f;
    int e;
    (void)std::frexp(m, &e);  // m = f * 2^e, f in [0.5, 1): m < 2^e
    return std::ldexp(1.f, std::max(-100, std::min(100, 15 - e)));
}
static float max_row_abs_sum(const float* w, int rows, size_t row_len) {
    double best = 0.0;
    for (int r = 0; r < rows; ++r) {
        double a = 0.0;
        for (size_t i = 0; i < row_len; ++i) a += std::fabs((double)w[(size_t)r * row_len + i]);
        best = std::max(best, a);
    }
    return (float)(best * 1.0000002);  // rounded up
}

int kws_load_cnn_trad(kws_ctx* c, const float* blob, size_t n_floats, int num_classes) {
    KWS_GUARD_BEGIN
    if (!c) return KWS_EINVAL;
    if (!blob) return fail(c, KWS_EINVAL, "kws_load_cnn_trad: blob is NULL");
    if (num_classes < 1 || num_classes > MAX_CLASSES) return fail(c, KWS_EUNSUPPORTED, "kws_load_cnn_trad: num_classes must be in [1, 64]");
    const size_t FLAT = 64 * 297;
    const size_t n_c1 = 64 * 160, n_c2 = 64 * 64 * 40, n_lin = 32 * FLAT, n_dnn = 128 * 32, n_fc = (size_t)num_classes * 128;
    const size_t expect = n_c1 + 64 + n_c2 + 64 + n_lin + 32 + n_dnn + 128 + n_fc + num_classes;
    if (n_floats != expect) {
        char msg[160];
        snprintf(msg, sizeof msg, "kws_load_cnn_trad: expected %zu floats for %d classes, got %zu", expect, num_classes, n_floats);
        return fail(c, KWS_EINVAL, msg);
    }
    const float *w1 = blob, *b1 = w1 + n_c1, *w2 = b1 + 64, *b2 = w2 + n_c2, *wl = b2 + 64, *bl = wl + n_lin, *wd = bl + 32,
                *bd = wd + n_dnn, *wf = bd + 128, *bf = wf + n_fc;
    // device image (units: 32-bit words): c1_split | c2_split | c1_b | c2_b | lin_split | lin_b | dnn_w | dnn_b | fc_w | fc_b
    const size_t o_c1s = 0, o_c2s = o_c1s + 10 * 2 * 3 * 64 * 4, o_c1b = o_c2s + (size_t)40 * 4 * 2 * 3 * 64 * 4, o_c2b = o_c1b + 64,
                 o_lin = o_c2b + 64, o_linb = o_lin + (size_t)(FLAT / 16) * 3 * 64 * 4, o_dnn = o_linb + 32, o_dnnb = o_dnn + n_dnn, o_fc = o_dnnb + 128,
                 o_fcb = o_fc + n_fc,
                 // f16-pair images (two pieces), 16-byte aligned
                 o_c1h = (o_fcb + num_classes + 3) / 4 * 4, o_c2h = o_c1h + 10 * 2 * 2 * 64 * 4, o_linh = o_c2h + (size_t)40 * 4 * 2 * 2 * 64 * 4,
                 total = o_linh + (size_t)(FLAT / 16) * 2 * 64 * 4;
    std::vector<uint32_t> h(total, 0u);
    const float sw1 = pow2_weight_scale(w1, n_c1), sw2 = pow2_weight_scale(w2, n_c2), swl = pow2_weight_scale(wl, n_lin);
    auto put = [&](size_t off, const float* src, size_t n) { memcpy(&h[off], src, n * sizeof(float)); };
    // conv1: lane l of (kb, ct): cout = 32ct + (l&31), kernel row 2kb + (l>>5), kernel columns j = 0..7
    for (int kb = 0; kb < 10; ++kb)
        for (int ct = 0; ct < 2; ++ct)
            for (int l = 0; l < 64; ++l) {
                float v[8];
                const int co = 32 * ct + (l & 31), kh = 2 * kb + (l >> 5);
                for (int j = 0; j < 8; ++j) v[j] = w1[(co * 20 + kh) * 8 + j];
                uint32_t* base = &h[o_c1s + ((size_t)(kb * 2 + ct) * 3 * 64 + l) * 4];
                pack_split8(v, base, base + 64 * 4, base + 2 * 64 * 4);
                uint32_t* b2 = &h[o_c1h + ((size_t)(kb * 2 + ct) * 2 * 64 + l) * 4];
                pack_pair8(v, sw1, b2, b2 + 64 * 4);
            }
    // conv2: lane l of (kk = kh*4 + kw, cb, ct): cout = 32ct + (l&31), input channels 16cb + 8(l>>5) + j
    for (int kk = 0; kk < 40; ++kk)
        for (int cb = 0; cb < 4; ++cb)
            for (int ct = 0; ct < 2; ++ct)
                for (int l = 0; l < 64; ++l) {
                    float v[8];
                    const int co = 32 * ct + (l & 31), kh = kk >> 2, kw = kk & 3;
                    for (int j = 0; j < 8; ++j) v[j] = w2[((co * 64 + 16 * cb + 8 * (l >> 5) + j) * 10 + kh) * 4 + kw];
                    uint32_t* base = &h[o_c2s + ((((size_t)kk * 4 + cb) * 2 + ct) * 3 * 64 + l) * 4];
                    pack_split8(v, base, base + 64 * 4, base + 2 * 64 * 4);
                    uint32_t* b2 = &h[o_c2h + ((((size_t)kk * 4 + cb) * 2 + ct) * 2 * 64 + l) * 4];
                    pack_pair8(v, sw2, b2, b2 + 64 * 4);
                }
    put(o_c1b, b1, 64);
    put(o_c2b, b2, 64);
    // first dense layer as MFMA B operands (32x32x16): lane l of k-block kb: output l&31, inputs 16kb + 8(l>>5) + j
    for (size_t kb = 0; kb < FLAT / 16; ++kb)
        for (int l = 0; l < 64; ++l) {
            float v[8];
            for (int j = 0; j < 8; ++j) v[j] = wl[(size_t)(l & 31) * FLAT + 16 * kb + 8 * (l >> 5) + j];
            uint32_t* base = &h[o_lin + (kb * 3 * 64 + l) * 4];
            pack_split8(v, base, base + 64 * 4, base + 2 * 64 * 4);
            uint32_t* b2 = &h[o_linh + (kb * 2 * 64 + l) * 4];
            pack_pair8(v, swl, b2, b2 + 64 * 4);
        }
    put(o_linb, bl, 32);
    put(o_dnn, wd, n_dnn);
    put(o_dnnb, bd, 128);
    put(o_fc, wf, n_fc);
    put(o_fcb, bf, num_classes);

    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    uint32_t* d = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d), total * sizeof(uint32_t)) != hipSuccess)
        return fail(c, KWS_ENOMEM, "kws_load_cnn_trad: device allocation failed");
    hipError_t e = hipMemcpy(d, h.data(), total * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(d);
        return fail_hip(c, e, "kws_load_cnn_trad: hipMemcpy");
    }
    if (c->d_cnntrad) (void)hipFree(c->d_cnntrad);
    c->d_cnntrad = d;
    const float* df = reinterpret_cast<const float*>(d);
    c->tw.c1_split = d + o_c1s;
    c->tw.c2_split = d + o_c2s;
    c->tw.c1_b = df + o_c1b;
    c->tw.c2_b = df + o_c2b;
    c->tw.lin_split = d + o_lin;
    c->tw.lin_b = df + o_linb;
    c->tw.dnn_w = df + o_dnn;
    c->tw.dnn_b = df + o_dnnb;
    c->tw.fc_w = df + o_fc;
    c->tw.fc_b = df + o_fcb;
    c->tw.num_classes = num_classes;
    c->tw.c1_h2 = d + o_c1h;
    c->tw.c2_h2 = d + o_c2h;
    c->tw.lin_h2 = d + o_linh;
    c->tw.inv_sw1 = 1.f / sw1;
    c->tw.inv_sw2 = 1.f / sw2;
    c->tw.inv_swl = 1.f / swl;
    c->tw.w1_abs = max_row_abs_sum(w1, 64, 160);
    c->tw.w2_abs = max_row_abs_sum(w2, 64, 2560);
    c->tw.b1_max = 0.f;
    c->tw.b2_max = 0.f;
    for (int i = 0; i < 64; ++i) {
        c->tw.b1_max = std::max(c->tw.b1_max, std::fabs(b1[i]));
        c->tw.b2_max = std::max(c->tw.b2_max, std::fabs(b2[i]));
    }
    c->cnntrad_ready = true;
    return KWS_OK;
    KWS_GUARD_END(c, "kws_load_cnn_trad")
}

// Grow the context's float scratch (convolution outputs between two kernels of one call) to at least `need` floats.
static int grow_conv_ws(kws_ctx* c, size_t need, const std::string& fn) {
    if (need <= c->conv_ws_floats) return KWS_OK;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    float* d = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d), need * sizeof(float)) != hipSuccess)
        return fail(c, KWS_ENOMEM, fn + ": workspace allocation failed");
    if (c->d_conv_ws) (void)hipFree(c->d_conv_ws);
    c->d_conv_ws = d;
    c->conv_ws_floats = need;
    return KWS_OK;
}

int kws_dsblock_forward_f32(kws_ctx* c, const float* d_x, int B, int C_in, int H, int W, const float* d_dw_w, const float* d_dw_b,
                            const float* d_pw_w, const float* d_pw_b, int C_out, int kernel_size, int stride, int padding,
                            float* d_out) {
    int rc = check_batch(c, d_x, B, "kws_dsblock_forward_f32");
    if (rc) return rc;
    if (!d_dw_w || !d_dw_b || !d_pw_w || !d_pw_b || !d_out) return fail(c, KWS_EINVAL, "kws_dsblock_forward_f32: NULL pointer");
    if (C_in < 1 || C_out < 1 || H < 1 || W < 1 || kernel_size < 1 || stride < 1 || padding < 0)
        return fail(c, KWS_EINVAL, "kws_dsblock_forward_f32: sizes must be positive (padding >= 0)");
    if (H + 2 * padding < kernel_size || W + 2 * padding < kernel_size)
        return fail(c, KWS_EINVAL, "kws_dsblock_forward_f32: the kernel is larger than the padded input");
    if (B > 65535) return fail(c, KWS_EUNSUPPORTED, "kws_dsblock_forward_f32: B must be <= 65535 per call");
    HIP_TRY(c, hipSetDevice(c->device));
    const int Ho = (H + 2 * padding - kernel_size) / stride + 1, Wo = (W + 2 * padding - kernel_size) / stride + 1;
    rc = grow_conv_ws(c, (size_t)B * C_in * Ho * Wo, "kws_dsblock_forward_f32");
    if (rc) return rc;
    HIP_TRY(c, launch_dsblock(c->stream, d_x, B, C_in, H, W, d_dw_w, d_dw_b, d_pw_w, d_pw_b, C_out, kernel_size, stride, padding,
                              c->d_conv_ws, d_out));
    return KWS_OK;
}

int kws_forward_cnn_trad_f32(kws_ctx* c, const float* d_feat, int B, float* d_logits, int32_t* d_label) {
    int rc = check_batch(c, d_feat, B, "kws_forward_cnn_trad_f32");
    if (rc) return rc;
    if (!d_logits) return fail(c, KWS_EINVAL, "kws_forward_cnn_trad_f32: d_logits is NULL");
    if (!c->cnntrad_ready) return fail(c, KWS_ESTATE, "kws_forward_cnn_trad_f32: no model loaded (kws_load_cnn_trad)");
    HIP_TRY(c, hipSetDevice(c->device));
    rc = grow_conv_ws(c, (size_t)B * 64 * 297 + (size_t)B, "kws_forward_cnn_trad_f32");  // + one scale per clip (f16-pair arithmetic)
    if (rc) return rc;
    const bool pair = c->cnntrad_math == KWS_CT_F16_PAIR;
    {
        ProfScope ps(c, KWS_K_CNNTRAD_CONV);
        HIP_TRY(c, launch_cnntrad_conv(c->stream, c->tw, d_feat, B, c->d_conv_ws, pair));
    }
    {
        ProfScope ps(c, KWS_K_CNNTRAD_DENSE);
        HIP_TRY(c, launch_cnntrad_dense(c->stream, c->tw, c->d_conv_ws, B, d_logits, d_label, pair));
    }
    return KWS_OK;
}

int kws_set_cnn_trad_math(kws_ctx* c, int math) {
    if (!c) return KWS_EINVAL;
    if (math != KWS_CT_F16_PAIR && math != KWS_CT_BF16_TRIPLE) return fail(c, KWS_EINVAL, "kws_set_cnn_trad_math: unknown arithmetic");
    c->cnntrad_math = math;
    return KWS_OK;
}

int kws_infer_cnn_trad_i16(kws_ctx* c, const int16_t* d_wav, int B, float* d_logits, int32_t* d_label) {
    int rc = check_batch(c, d_wav, B, "kws_infer_cnn_trad_i16");
    if (rc) return rc;
    if (!c->fe_ready || !c->cnntrad_ready)
        return fail(c, KWS_ESTATE, "kws_infer_cnn_trad_i16: front end or model not configured (kws_load_cnn_trad)");
    if (c->fp.num_frames != IN_T || c->fp.numcep != IN_F)
        return fail(c, KWS_EUNSUPPORTED, "kws_infer_cnn_trad_i16: the kernels are built for a 99 x 10 feature map");
    rc = kws_reserve(c, B);
    if (rc) return rc;
    rc = kws_mfcc_i16(c, d_wav, B, c->d_feat_ws);
    if (rc) return rc;
    return kws_forward_cnn_trad_f32(c, c->d_feat_ws, B, d_logits, d_label);
}

int kws_stream_vad_f32(kws_ctx* c, float log_energy_threshold, int on_window, int off_window, int32_t* d_state) {
    if (!c) return KWS_EINVAL;
    if (!c->n_streams) return fail(c, KWS_ESTATE, "kws_stream_vad_f32: call kws_stream_open first");
    if (!d_state) return fail(c, KWS_EINVAL, "kws_stream_vad_f32: d_state is NULL");
    if (on_window < 1 || off_window < on_window || off_window > 1024)
        return fail(c, KWS_EINVAL, "kws_stream_vad_f32: need 1 <= on_window <= off_window <= 1024");
    if (!c->fp.append_energy) return fail(c, KWS_EUNSUPPORTED, "kws_stream_vad_f32: needs cepstrum 0 = log frame energy (appendEnergy)");
    HIP_TRY(c, hipSetDevice(c->device));
    if (on_window != c->vad_on || off_window != c->vad_off) {  // (re)start the history
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        vad_free(c);
        const size_t fb = (size_t)c->n_streams * off_window, sb = sizeof(int) * 2 * (size_t)c->n_streams;
        if (hipMalloc(reinterpret_cast<void**>(&c->d_vad_flags), fb) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&c->d_vad_state), sb) != hipSuccess) {
            vad_free(c);
            return fail(c, KWS_ENOMEM, "kws_stream_vad_f32: device allocation failed");
        }
        HIP_TRY(c, hipMemsetAsync(c->d_vad_flags, 0, fb, c->stream));
        HIP_TRY(c, hipMemsetAsync(c->d_vad_state, 0, sb, c->stream));
        c->vad_on = on_window;
        c->vad_off = off_window;
    }
    HIP_TRY(c, launch_stream_vad(c->stream, c->d_feat_ring, c->d_hops, c->n_streams, c->fp.num_frames, c->fp.numcep,
                                 (c->fp.frame_len + c->fp.frame_step - 1) / c->fp.frame_step, log_energy_threshold, on_window, off_window, c->d_vad_flags, c->d_vad_state, d_state));
    return KWS_OK;
}

int kws_softmax_f32(kws_ctx* c, const float* d_logits, int B, int C, float* d_prob) {
    int rc = check_batch(c, d_logits, B, "kws_softmax_f32");
    if (rc) return rc;
    if (!d_prob) return fail(c, KWS_EINVAL, "kws_softmax_f32: d_prob is NULL");
    if (C < 1 || C > MAX_CLASSES) return fail(c, KWS_EUNSUPPORTED, "kws_softmax_f32: C must be in [1, 64]");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, launch_softmax(c->stream, d_logits, B, C, d_prob));
    return KWS_OK;
}

int kws_stream_smooth_f32(kws_ctx* c, const float* d_logits, int C, int window, float* d_smoothed, int32_t* d_label) {
    if (!c) return KWS_EINVAL;
    if (!c->n_streams) return fail(c, KWS_ESTATE, "kws_stream_smooth_f32: call kws_stream_open first");
    if (!d_logits || !d_smoothed) return fail(c, KWS_EINVAL, "kws_stream_smooth_f32: d_logits / d_smoothed is NULL");
    if (C < 1 || C > MAX_CLASSES) return fail(c, KWS_EUNSUPPORTED, "kws_stream_smooth_f32: C must be in [1, 64]");
    if (window < 1 || window > 4096) return fail(c, KWS_EINVAL, "kws_stream_smooth_f32: window must be in [1, 4096]");
    HIP_TRY(c, hipSetDevice(c->device));
    if (window != c->post_window || C != c->post_classes) {  // (re)start the history
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        smooth_free(c);
        const size_t ring_b = sizeof(float) * (size_t)c->n_streams * window * C, sum_b = sizeof(float) * (size_t)c->n_streams * C;
        if (hipMalloc(reinterpret_cast<void**>(&c->d_post_ring), ring_b) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&c->d_post_sum), sum_b) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&c->d_post_count), 2 * sizeof(int)) != hipSuccess) {
            smooth_free(c);
            return fail(c, KWS_ENOMEM, "kws_stream_smooth_f32: device allocation failed");
        }
        HIP_TRY(c, hipMemsetAsync(c->d_post_ring, 0, ring_b, c->stream));
        HIP_TRY(c, hipMemsetAsync(c->d_post_sum, 0, sum_b, c->stream));
        HIP_TRY(c, hipMemsetAsync(c->d_post_count, 0, 2 * sizeof(int), c->stream));
        c->post_window = window;
        c->post_classes = C;
    }
    HIP_TRY(c, launch_smooth_posteriors(c->stream, d_logits, c->n_streams, C, window, c->d_post_ring, c->d_post_sum,
                                        c->d_post_count, d_smoothed, d_label));
    return KWS_OK;
}

int kws_stream_close(kws_ctx* c) {
    if (!c) return KWS_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    stream_free(c);
    return KWS_OK;
}

// timed: bracket the launches with profiling events (eager pushes only; never inside a stream capture).
// A push that asks for logits from the product DS-CNN is ONE launch: each stream's workgroup computes the stream's new
// frame in its prologue (launch_dscnn_stream).  Features-only pushes, and the diagnostic pointwise variants, take the
// frame kernel followed -- if logits are wanted -- by the DS-CNN kernel over the advanced ring.
static hipError_t stream_enqueue(kws_ctx* c, const int16_t* d_hop, float* d_logits, int32_t* d_label, bool timed) {
    hipError_t e;
    if (d_logits && (c->pw_math == KWS_PW_SPLIT_BF16 || c->pw_math == KWS_PW_PAIR_F16)) {
        const bool was = c->prof;
        c->prof = was && timed;
        ProfScope ps(c, KWS_K_DSCNN);
        c->prof = was;
        // time-tile clusters while there are CUs to spare: 4 workgroups per stream up to 64 streams, 2 up to 128 (256 CUs)
        const int cluster = c->stream_cluster ? c->stream_cluster : (c->n_streams <= 64 ? 4 : (c->n_streams <= 128 ? 2 : 1));
        const bool host = c->h_stream_flag && c->host_results_classes == c->mw.num_classes;
        const StreamPush sp = {c->fp, c->ft, d_hop, c->d_pcm_ring, c->ring_len, c->d_hops, c->d_refine, 0, cluster, c->d_cl_part, c->d_cl_count,
                               host ? c->h_stream_logits : nullptr, host ? c->h_stream_label : nullptr, host ? c->h_stream_flag : nullptr};
        c->last_push_host = host;
        return launch_dscnn_stream(c->stream, c->mw, sp, c->d_feat_ring, c->n_streams, d_logits, d_label, c->pw_math == KWS_PW_PAIR_F16);
    }
    c->last_push_host = false;
    {
        const bool was = c->prof;
        c->prof = was && timed;
        ProfScope ps(c, KWS_K_STREAM_FRAME);
        c->prof = was;
        e = launch_stream_frame(c->stream, c->fp, c->ft, d_hop, c->n_streams, c->d_pcm_ring, c->ring_len, c->d_feat_ring,
                                c->d_hops, c->d_refine);
    }
    if (e != hipSuccess) return e;
    if (d_logits) {
        const bool was = c->prof;
        c->prof = was && timed;
        ProfScope ps(c, KWS_K_DSCNN);
        c->prof = was;
        e = launch_dscnn(c->stream, c->mw, c->d_feat_ring, c->n_streams, d_logits, d_label, nullptr, c->pw_math, nullptr, c->d_hops, false,
                         (c->fp.frame_len + c->fp.frame_step - 1) / c->fp.frame_step);
    }
    return e;
}

int kws_stream_push_i16(kws_ctx* c, const int16_t* d_hop, float* d_logits, int32_t* d_label, int use_graph) {
    if (!c) return KWS_EINVAL;
    if (!c->n_streams) return fail(c, KWS_ESTATE, "kws_stream_push_i16: call kws_stream_open first");
    if (!d_hop) return fail(c, KWS_EINVAL, "kws_stream_push_i16: d_hop is NULL");
    if (d_logits) {
        if (!c->model_ready) return fail(c, KWS_ESTATE, "kws_stream_push_i16: no model loaded (kws_load_dscnn)");
        if (c->mw.in_channels != 1) return fail(c, KWS_EUNSUPPORTED, "kws_stream_push_i16: the model was loaded with more than one input channel");
        if (c->fp.num_frames != IN_T || c->fp.numcep != IN_F)
            return fail(c, KWS_EUNSUPPORTED, "kws_stream_push_i16: the DS-CNN kernel is built for a 99 x 10 feature map");
    }
    HIP_TRY(c, hipSetDevice(c->device));
    // A push that is ONE kernel launch gains nothing from a graph -- on this stack it loses: hipGraphLaunch of a one-node graph
    // takes 6.9 us of host time against 3.0 us for the plain launch, and completion is observed 8 us later in all
    // (tools/graph_overhead.hip: 20.6 vs 12.3 us launch -> hipStreamSynchronize for a trivial kernel; still 23.6 vs 20.3 us at
    // four kernels).  use_graph is honoured for the multi-launch routes only, where it saves host time per push.
    if (use_graph && d_logits && (c->pw_math == KWS_PW_SPLIT_BF16 || c->pw_math == KWS_PW_PAIR_F16)) use_graph = 0;
    if (use_graph) {
        // one hipGraph per (hop, logits, label) pointer triple: the two launches replay as one submission
        if (!c->stream_graph || c->graph_key[0] != d_hop || c->graph_key[1] != d_logits || c->graph_key[2] != d_label) {
            if (c->stream_graph) {  // other buffers than the captured ones: retire the old graph once it is idle
                HIP_TRY(c, hipStreamSynchronize(c->stream));
                drop_stream_graph(c);
            }
            hipGraph_t g = nullptr;
            HIP_TRY(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
            hipError_t e = stream_enqueue(c, d_hop, d_logits, d_label, false);
            hipError_t e2 = hipStreamEndCapture(c->stream, &g);
            if (e != hipSuccess || e2 != hipSuccess || !g) return fail_hip(c, e != hipSuccess ? e : e2, "kws_stream_push_i16: graph capture");
            e = hipGraphInstantiate(&c->stream_graph, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            if (e != hipSuccess) return fail_hip(c, e, "kws_stream_push_i16: hipGraphInstantiate");
            c->graph_key[0] = d_hop;
            c->graph_key[1] = d_logits;
            c->graph_key[2] = d_label;
        }
        HIP_TRY(c, hipGraphLaunch(c->stream_graph, c->stream));
        c->pushes_enqueued += 1;
        return KWS_OK;
    }
    HIP_TRY(c, stream_enqueue(c, d_hop, d_logits, d_label, true));
    c->pushes_enqueued += 1;
    if (c->last_push_host) c->host_push = c->pushes_enqueued;
    return KWS_OK;
}

int kws_stream_state(kws_ctx* c, const float** d_feat_ring, int* hops) {
    if (!c) return KWS_EINVAL;
    if (!c->n_streams) return fail(c, KWS_ESTATE, "kws_stream_state: no open stream set");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (d_feat_ring) *d_feat_ring = c->d_feat_ring;
    if (hops) HIP_TRY(c, hipMemcpy(hops, c->d_hops, sizeof(int), hipMemcpyDeviceToHost));
    return KWS_OK;
}

int kws_stream_copy_features(kws_ctx* c, float* d_out) {
    if (!c) return KWS_EINVAL;
    if (!c->n_streams) return fail(c, KWS_ESTATE, "kws_stream_copy_features: no open stream set");
    if (!d_out) return fail(c, KWS_EINVAL, "kws_stream_copy_features: d_out is NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(d_out, c->d_feat_ring, sizeof(float) * (size_t)c->n_streams * c->fp.num_frames * c->fp.numcep,
                              hipMemcpyDeviceToDevice, c->stream));
    return KWS_OK;
}

// ---- augmentation ---------------------------------------------------------------------------------
int kws_augment_i16(kws_ctx* c, const int16_t* d_wav, int B, const int32_t* d_shift, const float* d_bg, int bg_len,
                    const int32_t* d_bg_off, const float* d_bg_vol, const uint8_t* d_silence, float* d_out) {
    int rc = check_batch(c, d_wav, B, "kws_augment_i16");
    if (rc) return rc;
    if (!d_out) return fail(c, KWS_EINVAL, "kws_augment_i16: d_out is NULL");
    if (d_bg && (bg_len <= 0 || !d_bg_off || !d_bg_vol)) return fail(c, KWS_EINVAL, "kws_augment_i16: background pool needs length, offsets and volumes");
    if (!c->fe_ready) return fail(c, KWS_ESTATE, "kws_augment_i16: front end not configured");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, launch_augment(c->stream, d_wav, B, c->fp.n_samples, d_shift, d_bg, bg_len, d_bg_off, d_bg_vol, d_silence, d_out));
    return KWS_OK;
}

// ---- sigproc operators --------------------------------------------------------------------------
int kws_preemphasis_f32(kws_ctx* c, const float* d_signal, int n, float coeff, float* d_out) {
    int rc = check_batch(c, d_signal, n, "kws_preemphasis_f32");
    if (rc) return rc;
    if (!d_out) return fail(c, KWS_EINVAL, "kws_preemphasis_f32: d_out is NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, launch_preemphasis(c->stream, d_signal, n, coeff, d_out));
    return KWS_OK;
}

int kws_framesig_f32(kws_ctx* c, const float* d_signal, int n, int frame_len, int frame_step, const float* d_window,
                     float* d_frames) {
    int rc = check_batch(c, d_signal, n, "kws_framesig_f32");
    if (rc) return rc;
    if (!d_frames || frame_len <= 0 || frame_step <= 0) return fail(c, KWS_EINVAL, "kws_framesig_f32: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, launch_framesig(c->stream, d_signal, n, frame_len, frame_step, frames_for(n, frame_len, frame_step), d_window, d_frames));
    return KWS_OK;
}

int kws_spec512_f32(kws_ctx* c, const float* d_frames, int num_frames, int frame_len, int power, float* d_spec) {
    int rc = check_batch(c, d_frames, num_frames, "kws_spec512_f32");
    if (rc) return rc;
    if (!d_spec || frame_len <= 0) return fail(c, KWS_EINVAL, "kws_spec512_f32: bad argument");
    if (!c->fe_ready) return fail(c, KWS_ESTATE, "kws_spec512_f32: front end tables not built");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, launch_spec512(c->stream, c->ft, d_frames, num_frames, frame_len, power, d_spec));
    return KWS_OK;
}

int kws_spec_f32(kws_ctx* c, const float* d_frames, int num_frames, int frame_len, int nfft, int power, float* d_spec) {
    KWS_GUARD_BEGIN
    int rc = check_batch(c, d_frames, num_frames, "kws_spec_f32");
    if (rc) return rc;
    if (!d_spec || frame_len <= 0 || nfft < 2) return fail(c, KWS_EINVAL, "kws_spec_f32: bad argument");
    if (nfft == NFFT) return kws_spec512_f32(c, d_frames, num_frames, frame_len, power, d_spec);
    int log2n = 0;
    if ((nfft & (nfft - 1)) == 0)
        for (int v = nfft; v > 1; v >>= 1) ++log2n;
    if (nfft > 4096 || (log2n == 0 && nfft > 2048) || (log2n > 0 && nfft < 64))
        return fail(c, KWS_EUNSUPPORTED, "kws_spec_f32: NFFT must be a power of two in [64, 4096] or any value in [2, 2048]");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->spec_nfft != nfft) {  // float64 twiddles of this transform length, kept until another length is asked for
        std::vector<double> tw(2 * (size_t)nfft);
        const double pi = 3.14159265358979323846;
        for (int k = 0; k < nfft; ++k) {
            tw[2 * k] = std::cos(2.0 * pi * k / nfft);
            tw[2 * k + 1] = -std::sin(2.0 * pi * k / nfft);
        }
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        double* d = nullptr;
        if (hipMalloc(reinterpret_cast<void**>(&d), sizeof(double) * tw.size()) != hipSuccess)
            return fail(c, KWS_ENOMEM, "kws_spec_f32: device allocation failed");
        hipError_t e = hipMemcpy(d, tw.data(), sizeof(double) * tw.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(d);
            return fail_hip(c, e, "kws_spec_f32: hipMemcpy");
        }
        if (c->d_spec_tw64) (void)hipFree(c->d_spec_tw64);
        c->d_spec_tw64 = d;
        c->spec_nfft = nfft;
    }
    HIP_TRY(c, launch_spec_f64(c->stream, c->d_spec_tw64, d_frames, num_frames, frame_len, nfft, log2n, power, d_spec));
    return KWS_OK;
    KWS_GUARD_END(c, "kws_spec_f32")
}

// ---- measurement ---------------------------------------------------------------------------------
int kws_prof_enable(kws_ctx* c, int on) {
    if (!c) return KWS_EINVAL;
    c->prof = on != 0;
    c->prof_every = on > 1 ? on : 1;
    return KWS_OK;
}

static int prof_drain(kws_ctx* c) {
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int k = 0; k < KWS_K_COUNT; ++k) {
        for (size_t i = 0; i < c->ev_used[k]; ++i) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, c->ev[k][i].a, c->ev[k][i].b) == hipSuccess) {
                c->ms_total[k] += ms;
                c->launches[k] += 1;
            }
        }
        c->ev_used[k] = 0;
    }
    return KWS_OK;
}

int kws_prof_reset(kws_ctx* c) {
    if (!c) return KWS_EINVAL;
    int rc = prof_drain(c);
    for (int k = 0; k < KWS_K_COUNT; ++k) {
        c->ms_total[k] = 0;
        c->launches[k] = 0;
        c->prof_seen[k] = 0;
    }
    return rc;
}

int kws_prof_read(kws_ctx* c, int kernel_id, double* total_ms, int* launches) {
    if (!c) return KWS_EINVAL;
    if (kernel_id < 0 || kernel_id >= KWS_K_COUNT) return fail(c, KWS_EINVAL, "kws_prof_read: bad kernel id");
    int rc = prof_drain(c);
    if (rc) return rc;
    if (total_ms) *total_ms = c->ms_total[kernel_id];
    if (launches) *launches = (int)c->launches[kernel_id];
    return KWS_OK;
}

// ---- host-only helpers -----------------------------------------------------------------------------
int kws_host_mel_edges(int nfilt, int nfft, int sample_rate, int* edges_out) {
    if (!edges_out || nfilt < 1 || nfft < 2 || sample_rate < 1) return KWS_EINVAL;
    std::vector<int> e;
    mel_edges(nfilt, nfft, sample_rate, e);
    memcpy(edges_out, e.data(), sizeof(int) * e.size());
    return KWS_OK;
}

int kws_host_mel_dense(int nfilt, int nfft, int sample_rate, float* fb_out) {
    if (!fb_out) return KWS_EINVAL;
    MelHost mel;
    std::string err;
    if (!build_mel_host(nfilt, nfft, sample_rate, mel, err)) return KWS_EUNSUPPORTED;
    const int nb = nfft / 2 + 1;
    std::fill(fb_out, fb_out + (size_t)nfilt * nb, 0.f);
    // expand exactly what the kernel evaluates: filter j = rising weights of its chunks + falling weights
    // of the next segment's chunks
    for (int j = 0; j < nfilt; ++j) {
        const uint32_t g = mel.gather[j];
        const int r0 = g & 255, nr = (g >> 8) & 255, q0 = (g >> 16) & 255, nq = g >> 24;
        for (int c = r0; c < r0 + nr; ++c)
            for (int i = 0; i < MEL_CHUNK; ++i) {
                const int k = mel.k0[c] + i;
                if (k < nb) fb_out[(size_t)j * nb + k] += mel.rw[i * 64 + c];
            }
        for (int c = q0; c < q0 + nq; ++c)
            for (int i = 0; i < MEL_CHUNK; ++i) {
                const int k = mel.k0[c] + i;
                if (k < nb) fb_out[(size_t)j * nb + k] += mel.fw[i * 64 + c];
            }
    }
    return KWS_OK;
}

int kws_host_mel_layout(int nfilt, int nfft, int sample_rate, int* first_lane_out, int* n_lanes_out, int* lanes_used, int* row_safe) {
    if (!first_lane_out || !n_lanes_out) return KWS_EINVAL;
    MelHost mel;
    std::string err;
    if (!build_mel_host(nfilt, nfft, sample_rate, mel, err)) return KWS_EUNSUPPORTED;
    // segment s = filter s's rising side; the last segment is the falling side of the last filter
    for (int j = 0; j < nfilt; ++j) {
        first_lane_out[j] = (int)(mel.gather[j] & 255);
        n_lanes_out[j] = (int)((mel.gather[j] >> 8) & 255);
    }
    first_lane_out[nfilt] = (int)((mel.gather[nfilt - 1] >> 16) & 255);
    n_lanes_out[nfilt] = (int)(mel.gather[nfilt - 1] >> 24);
    if (lanes_used) *lanes_used = mel.n_chunks;
    if (row_safe) *row_safe = (mel.seg[0] & 64) ? 1 : 0;
    return KWS_OK;
}

int kws_host_dct_lifter(int nfilt, int numcep, int ceplifter, float* out) {
    if (!out || nfilt < 1 || numcep < 1) return KWS_EINVAL;
    std::vector<float> t;
    build_dct_lifter_host(nfilt, numcep, ceplifter, t);
    memcpy(out, t.data(), sizeof(float) * t.size());
    return KWS_OK;
}

}  // extern "C"
#pragma GCC visibility pop

static void stream_free_fwd(kws_ctx* c) { stream_free(c); }
