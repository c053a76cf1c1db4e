// Host ingest for the fused path: kws_infer_host_i16 takes a HOST batch of PCM clips and
// returns HOST logits / labels.  It replaces what the reference does between the decoded audio and the model input --
// DataLoader workers collating batches into pinned memory and `inputs.to(device)` (kws/libs/data_loader.py:96-105,
// train.py:108-121, kws/libs/training.py:286) -- with a three-stage pipeline inside the library:
//
//   pack   the batch is cut into chunks of `chunk` clips; a pool of host threads copies chunk k+1 from the caller's
//          (pageable) memory into a pinned staging slot while
//   H2D    chunk k travels to the device on a copy stream (hipMemcpyAsync from pinned memory = one DMA), while
//   run    chunk k-1 runs MFCC + DS-CNN on the context's stream, and
//   D2H    the 52 bytes per clip of results of chunk k-2 return on a second copy stream into pinned memory.
//
// Events order the streams; the host thread blocks only when it needs a slot that is still in flight.  With S slots up
// to S chunks are in flight.  A caller buffer that is already pinned (hipHostMalloc / hipHostRegister / torch
// pin_memory) skips the pack stage: the DMA reads it directly.  The PCIe-inclusive rate of this path is reported by
// tools/bench_ingest.py, never as bench.py's `value`.
#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>
#include <system_error>
#include <thread>

#include "kws_ctx.h"

namespace kws {

// A fixed set of host threads that copy one chunk in parallel (a single memcpy moves ~10-20 GB/s; the PCIe DMA behind it
// takes ~50): job = (dst, src, bytes), cut into equal 4 KiB-aligned slices.
class PackPool {
  public:
    explicit PackPool(int n) {
        // a host that refuses more threads (cgroup pids limit, RLIMIT_NPROC) still gets a pool: of the threads that did start,
        // or of none -- copy() then runs on the calling thread.  Nothing is thrown across the C ABI.
        try {
            for (int i = 0; i < n; ++i) workers_.emplace_back([this, i] { run(i); });
        } catch (const std::system_error&) {
        }
        n_started_ = workers_.size();
    }
    ~PackPool() {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    int size() const { return (int)workers_.size(); }
    // blocks until the whole copy is done (the caller enqueues the DMA right after)
    void copy(void* dst, const void* src, size_t bytes) {
        if (workers_.empty() || bytes < (1u << 20)) {
            memcpy(dst, src, bytes);
            return;
        }
        std::unique_lock<std::mutex> g(m_);
        dst_ = static_cast<unsigned char*>(dst);
        src_ = static_cast<const unsigned char*>(src);
        bytes_ = bytes;
        pending_ = (int)n_started_;
        ++gen_;
        cv_.notify_all();
        done_.wait(g, [this] { return pending_ == 0; });
    }

  private:
    void run(int idx) {
        unsigned long seen = 0;
        for (;;) {
            unsigned char* dst;
            const unsigned char* src;
            size_t bytes;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
                dst = dst_, src = src_, bytes = bytes_;
            }
            const size_t n = n_started_;
            const size_t slice = ((bytes + n - 1) / n + 4095) & ~(size_t)4095;
            const size_t lo = std::min(bytes, slice * idx), hi = std::min(bytes, lo + slice);
            if (hi > lo) memcpy(dst + lo, src + lo, hi - lo);
            {
                std::lock_guard<std::mutex> g(m_);
                if (--pending_ == 0) done_.notify_one();
            }
        }
    }
    std::vector<std::thread> workers_;
    size_t n_started_ = 0;  // fixed before any job is posted (workers_.size() must not be read while the vector may still grow)
    std::mutex m_;
    std::condition_variable cv_, done_;
    unsigned char* dst_ = nullptr;
    const unsigned char* src_ = nullptr;
    size_t bytes_ = 0;
    int pending_ = 0;
    unsigned long gen_ = 0;
    bool stop_ = false;
};

struct IngestSlot {
    int16_t* h_in = nullptr;      // pinned [chunk][n_samples]
    int16_t* d_in = nullptr;      // device
    float* d_logits = nullptr;    // device [chunk][C]
    int32_t* d_label = nullptr;   // device [chunk]
    float* h_logits = nullptr;    // pinned
    int32_t* h_label = nullptr;   // pinned
    hipEvent_t staged = nullptr, computed = nullptr, drained = nullptr;
    int count = 0;                // clips this slot currently carries (0: free)
    float* dst_logits = nullptr;  // where its results go in the submitting call's host arrays (already offset to the chunk)
    int32_t* dst_label = nullptr;
    uint64_t ticket = 0;          // the submit call it belongs to
};

struct Ingest {
    int chunk = 0, n_slots = 0, n_samples = 0, classes = 0;
    hipStream_t copy_in[2] = {nullptr, nullptr}, copy_out = nullptr;  // two H2D streams: consecutive chunks ride different DMA engines
    std::vector<IngestSlot> slots;
    PackPool* pool = nullptr;
    int pool_threads = 0;
    uint64_t next_ticket = 1;
    int next_slot = 0, lane = 0;   // ring position and H2D stream of the next chunk (they persist across submit calls)
    // configuration requested through kws_ingest_config (0 = default)
    int want_chunk = 0, want_slots = 0, want_threads = 0;
};

static void ingest_release(Ingest* g) {
    if (!g) return;
    for (auto& s : g->slots) {
        if (s.h_in) (void)hipHostFree(s.h_in);
        if (s.d_in) (void)hipFree(s.d_in);
        if (s.d_logits) (void)hipFree(s.d_logits);
        if (s.d_label) (void)hipFree(s.d_label);
        if (s.h_logits) (void)hipHostFree(s.h_logits);
        if (s.h_label) (void)hipHostFree(s.h_label);
        if (s.staged) (void)hipEventDestroy(s.staged);
        if (s.computed) (void)hipEventDestroy(s.computed);
        if (s.drained) (void)hipEventDestroy(s.drained);
    }
    g->slots.clear();
    for (auto& st : g->copy_in) {
        if (st) (void)hipStreamDestroy(st);
        st = nullptr;
    }
    if (g->copy_out) (void)hipStreamDestroy(g->copy_out);
    g->copy_out = nullptr;
    g->chunk = g->n_slots = 0;
}

void ingest_free(kws_ctx* c) {
    if (!c || !c->ingest) return;
    ingest_release(c->ingest);
    delete c->ingest->pool;
    delete c->ingest;
    c->ingest = nullptr;
}

// (Re)build the staging rings for the current front end / model geometry and the requested configuration.
static int ingest_prepare(kws_ctx* c) {
    if (!c->ingest) c->ingest = new (std::nothrow) Ingest();
    Ingest* g = c->ingest;
    if (!g) return fail(c, KWS_ENOMEM, "kws_infer_host_i16: out of host memory");
    const int chunk = g->want_chunk > 0 ? g->want_chunk : 1024;
    const int n_slots = g->want_slots > 0 ? g->want_slots : 3;
    int threads = g->want_threads > 0 ? g->want_threads : (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency() / 2));
    if (g->want_threads < 0) threads = 0;  // pack on the calling thread
    const int n_samples = c->fp.n_samples, classes = c->mw.num_classes;
    if (!g->pool || g->pool_threads != threads) {
        delete g->pool;
        g->pool = new (std::nothrow) PackPool(threads);
        g->pool_threads = threads;
        if (!g->pool) return fail(c, KWS_ENOMEM, "kws_infer_host_i16: out of host memory");
    }
    if (g->chunk == chunk && g->n_slots == n_slots && g->n_samples == n_samples && g->classes == classes) return KWS_OK;
    for (auto& s : g->slots)
        if (s.count) return fail(c, KWS_ESTATE, "kws_infer_host_i16: the ingest geometry changed while batches are in flight (kws_infer_host_wait first)");
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    ingest_release(g);
    g->next_slot = g->lane = 0;
    HIP_TRY(c, hipStreamCreateWithFlags(&g->copy_in[0], hipStreamNonBlocking));
    HIP_TRY(c, hipStreamCreateWithFlags(&g->copy_in[1], hipStreamNonBlocking));
    HIP_TRY(c, hipStreamCreateWithFlags(&g->copy_out, hipStreamNonBlocking));
    g->slots.resize(n_slots);
    const size_t in_b = sizeof(int16_t) * (size_t)chunk * n_samples, lg_b = sizeof(float) * (size_t)chunk * classes,
                 lb_b = sizeof(int32_t) * (size_t)chunk;
    for (auto& s : g->slots) {
        if (hipHostMalloc(reinterpret_cast<void**>(&s.h_in), in_b, hipHostMallocDefault) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&s.d_in), in_b) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&s.d_logits), lg_b) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&s.d_label), lb_b) != hipSuccess ||
            hipHostMalloc(reinterpret_cast<void**>(&s.h_logits), lg_b, hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc(reinterpret_cast<void**>(&s.h_label), lb_b, hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&s.staged, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s.computed, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s.drained, hipEventDisableTiming) != hipSuccess) {
            ingest_release(g);
            return fail(c, KWS_ENOMEM, "kws_infer_host_i16: staging allocation failed");
        }
    }
    g->chunk = chunk;
    g->n_slots = n_slots;
    g->n_samples = n_samples;
    g->classes = classes;
    return kws_reserve(c, chunk);
}

// Results of the chunk a slot carries: wait for its D2H, hand them to the caller, mark the slot free.
static int ingest_collect(kws_ctx* c, IngestSlot& s) {
    if (s.count == 0) return KWS_OK;
    HIP_TRY(c, hipEventSynchronize(s.drained));
    const int C = c->ingest->classes;
    memcpy(s.dst_logits, s.h_logits, sizeof(float) * (size_t)s.count * C);
    if (s.dst_label) memcpy(s.dst_label, s.h_label, sizeof(int32_t) * (size_t)s.count);
    s.count = 0;
    return KWS_OK;
}

// 0: ordinary (pageable) host memory, 1: pinned / registered host memory (the DMA can read it), -1: not host memory at all
static int host_pointer_kind(const void* p) {
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();  // an ordinary malloc'ed pointer is "invalid value" for the query: not an error here
        return 0;
    }
    if (a.type == hipMemoryTypeHost) return 1;
    if (a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeArray) return -1;
    return 0;  // unregistered / managed: treated as pageable
}

}  // namespace kws

using namespace kws;

#pragma GCC visibility push(default)
extern "C" {

int kws_ingest_config(kws_ctx* c, int chunk_clips, int n_slots, int pack_threads) {
    if (!c) return KWS_EINVAL;
    if (c->ingest)
        for (auto& s : c->ingest->slots)
            if (s.count) return fail(c, KWS_ESTATE, "kws_ingest_config: batches are in flight (kws_infer_host_wait first)");
    if (chunk_clips < 0 || n_slots < 0 || (n_slots > 0 && n_slots < 2) || n_slots > 16 || pack_threads > 64)
        return fail(c, KWS_EINVAL, "kws_ingest_config: chunk_clips >= 0, n_slots 0 or 2..16, pack_threads <= 64");
    if (!c->ingest) c->ingest = new (std::nothrow) Ingest();
    if (!c->ingest) return fail(c, KWS_ENOMEM, "kws_ingest_config: out of host memory");
    c->ingest->want_chunk = chunk_clips;
    c->ingest->want_slots = n_slots;
    c->ingest->want_threads = pack_threads;
    return KWS_OK;
}

// Enqueue one host batch: every chunk is packed (pageable input), sent, computed and its results started on their way back;
// slots still carrying an older chunk are collected (results copied to THAT chunk's destination) as the ring comes round.
static int ingest_submit(kws_ctx* c, const int16_t* h_wav, int B, float* h_logits, int32_t* h_label, uint64_t* ticket_out) {
    if (!h_wav || !h_logits) return fail(c, KWS_EINVAL, "kws_infer_host_submit_i16: h_wav and h_logits must not be NULL");
    if (B <= 0) return fail(c, KWS_EINVAL, "kws_infer_host_submit_i16: B must be positive");
    if (!c->fe_ready || !c->model_ready) return fail(c, KWS_ESTATE, "kws_infer_host_submit_i16: front end or model not configured");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = ingest_prepare(c);
    if (rc) return rc;
    Ingest* g = c->ingest;
    const int n = g->n_samples, C = g->classes;
    const int kind = host_pointer_kind(h_wav);
    if (kind < 0 || host_pointer_kind(h_logits) < 0 || (h_label && host_pointer_kind(h_label) < 0))
        return fail(c, KWS_EINVAL, "kws_infer_host_submit_i16: h_wav / h_logits / h_label must be HOST pointers (device tensors go to kws_infer_i16)");
    const bool direct = kind == 1;  // the DMA can read the caller's buffer: no pack stage
    // Chunk size of THIS batch: a batch smaller than the ring (the reference's own batch_size = 1028, train.py:110, against
    // 3 x 1024) would otherwise be one chunk plus a tail, its pack / H2D / compute / D2H strictly one after the other; cut
    // into as many chunks as there are slots (not below 128 clips: a launch pair per chunk costs ~0.1 ms of fixed time).
    int chunk = (B + g->n_slots - 1) / g->n_slots;
    chunk = std::min(g->chunk, std::max(chunk, std::min(B, 128)));  // never above what the staging slots hold
    const uint64_t ticket = g->next_ticket++;
    for (int first = 0; first < B; first += chunk, g->next_slot = (g->next_slot + 1) % g->n_slots, g->lane ^= 1) {
        IngestSlot& s = g->slots[g->next_slot];
        rc = ingest_collect(c, s);  // blocks only if this slot's previous chunk is still in flight
        if (rc) return rc;
        const int count = std::min(chunk, B - first);
        const size_t bytes = sizeof(int16_t) * (size_t)count * n;
        const int16_t* src = h_wav + (size_t)first * n;
        if (!direct) {
            g->pool->copy(s.h_in, src, bytes);
            src = s.h_in;
        }
        HIP_TRY(c, hipMemcpyAsync(s.d_in, src, bytes, hipMemcpyHostToDevice, g->copy_in[g->lane]));
        HIP_TRY(c, hipEventRecord(s.staged, g->copy_in[g->lane]));
        HIP_TRY(c, hipStreamWaitEvent(c->stream, s.staged, 0));
        rc = kws_infer_i16(c, s.d_in, count, s.d_logits, s.d_label);
        if (rc) return rc;
        HIP_TRY(c, hipEventRecord(s.computed, c->stream));
        HIP_TRY(c, hipStreamWaitEvent(g->copy_out, s.computed, 0));
        HIP_TRY(c, hipMemcpyAsync(s.h_logits, s.d_logits, sizeof(float) * (size_t)count * C, hipMemcpyDeviceToHost, g->copy_out));
        HIP_TRY(c, hipMemcpyAsync(s.h_label, s.d_label, sizeof(int32_t) * (size_t)count, hipMemcpyDeviceToHost, g->copy_out));
        HIP_TRY(c, hipEventRecord(s.drained, g->copy_out));
        s.dst_logits = h_logits + (size_t)first * C;
        s.dst_label = h_label ? h_label + first : nullptr;
        s.ticket = ticket;
        s.count = count;
    }
    if (ticket_out) *ticket_out = ticket;
    return KWS_OK;
}

// Collect every chunk of the batches up to `ticket` (0: everything in flight), oldest first.
static int ingest_wait(kws_ctx* c, uint64_t ticket) {
    Ingest* g = c->ingest;
    if (!g || g->slots.empty()) return KWS_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    for (int i = 0; i < g->n_slots; ++i) {
        IngestSlot& s = g->slots[(g->next_slot + i) % g->n_slots];
        if (s.count && (ticket == 0 || s.ticket <= ticket)) {
            int rc = ingest_collect(c, s);
            if (rc) {  // a failed D2H: let the device finish, then forget what was in flight (the caller's arrays stay partial)
                (void)hipDeviceSynchronize();
                for (auto& t : g->slots) t.count = 0;
                return rc;
            }
        }
    }
    return KWS_OK;
}

int kws_infer_host_submit_i16(kws_ctx* c, const int16_t* h_wav, int B, float* h_logits, int32_t* h_label, uint64_t* ticket) {
    if (!c) return KWS_EINVAL;
    try {
        int rc = ingest_submit(c, h_wav, B, h_logits, h_label, ticket);
        if (rc) {  // a call that failed half-way may have left chunks in flight: let them finish, then forget them
            (void)hipDeviceSynchronize();
            if (c->ingest)
                for (auto& t : c->ingest->slots) t.count = 0;
        }
        return rc;
    } catch (const std::bad_alloc&) {
        return fail(c, KWS_ENOMEM, "kws_infer_host_submit_i16: out of host memory");
    } catch (...) {
        return fail(c, KWS_EHIP, "kws_infer_host_submit_i16: unexpected C++ exception");
    }
}

int kws_infer_host_wait(kws_ctx* c, uint64_t ticket) {
    if (!c) return KWS_EINVAL;
    try {
        return ingest_wait(c, ticket);
    } catch (...) {
        return fail(c, KWS_EHIP, "kws_infer_host_wait: unexpected C++ exception");
    }
}

int kws_infer_host_i16(kws_ctx* c, const int16_t* h_wav, int B, float* h_logits, int32_t* h_label) {
    if (!c) return KWS_EINVAL;
    uint64_t ticket = 0;
    int rc = kws_infer_host_submit_i16(c, h_wav, B, h_logits, h_label, &ticket);
    if (rc) return rc;
    return kws_infer_host_wait(c, ticket);
}

}  // extern "C"
#pragma GCC visibility pop
