// context.hip -- library context, error string, per-kernel HIP-event timing.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <vector>
#include <thread>

#include <chrono>

#include "internal.h"

namespace wsnark {

static thread_local std::string t_last_error;
void set_last_error(const std::string& s) { t_last_error = s; }
const std::string& get_last_error() { return t_last_error; }

static Context* g_ctx = nullptr;
static std::mutex g_ctx_mu;
static std::string g_devinfo;

Context* ctx() { return g_ctx; }

int context_init(int device) {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    if (g_ctx) return WS_OK;
    if (device < 0) {
        const char* lr = getenv("LOCAL_RANK");
        device = lr ? atoi(lr) : 0;
    }
    int count = 0;
    WS_HIP_CHECK(hipGetDeviceCount(&count));
    if (count <= 0) { set_last_error("no HIP device visible"); return WS_ERR_HIP; }
    if (device >= count) device = device % count;
    WS_HIP_CHECK(hipSetDevice(device));
    Context* C = new Context();
    C->device = device;
    hipDeviceProp_t prop;
    WS_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    C->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    g_devinfo = std::string(prop.name) + " " + prop.gcnArchName + " CUs=" + std::to_string(C->num_cu);
    WS_HIP_CHECK(hipStreamCreateWithFlags(&C->stream, hipStreamNonBlocking));
    {
        const char* e = getenv("WSNARK_LANES");
        int nl = e ? atoi(e) : 2;
        C->n_lanes = nl < 1 ? 1 : nl > kMaxLanes ? kMaxLanes : nl;
    }
    // WSNARK_S2_PRIO=1 gives every lane's second queue the device's highest priority (round 1's default).  Since the prover
    // finishes whichever sum is ready first, plain queues released at once are the best schedule on dense and sparse
    // keys alike (profiles/r02_sweep_prove_overlap.txt: 11.26 ms against 11.30-11.68 ms for the other three combinations)
    const char* pe = getenv("WSNARK_S2_PRIO");
    int lo = 0, hi = 0;
    const bool prio = (pe && atoi(pe) == 1) && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi != lo;
    for (int i = 0; i < C->n_lanes; i++) {
        Lane& L = C->lanes[i];
        L.id = i;
        WS_HIP_CHECK(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
        if (prio && hipStreamCreateWithPriority(&L.stream2, hipStreamNonBlocking, hi) != hipSuccess) L.stream2 = nullptr;
        if (!L.stream2) WS_HIP_CHECK(hipStreamCreateWithFlags(&L.stream2, hipStreamNonBlocking));   // (no priorities here)
        WS_HIP_CHECK(hipStreamCreateWithFlags(&L.stream3, hipStreamNonBlocking));
    }
    g_ctx = C;
    return WS_OK;
}

void context_shutdown() {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    if (!g_ctx) return;
    (void)hipSetDevice(g_ctx->device);
    (void)hipStreamSynchronize(g_ctx->stream);
    for (int i = 0; i < g_ctx->n_lanes; i++) {
        (void)hipStreamSynchronize(g_ctx->lanes[i].stream);
        (void)hipStreamSynchronize(g_ctx->lanes[i].stream2);
        (void)hipStreamSynchronize(g_ctx->lanes[i].stream3);
    }
    g_ctx->timer.reset();
    for (hipEvent_t e : g_ctx->timer.pool) (void)hipEventDestroy(e);
    g_ctx->timer.pool.clear();
    g_ctx->ntt_plans.clear();
    if (g_ctx->pin_ring) { (void)hipHostFree(g_ctx->pin_ring); g_ctx->pin_ring = nullptr; }
    for (auto& e : g_ctx->pin_ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    for (int i = 0; i < g_ctx->n_lanes; i++) {
        Lane& L = g_ctx->lanes[i];
        msm_workspace_free(L);
        for (hipEvent_t* e : {&L.ntt_chain.done, &L.calch_chain.done, &L.ev_start, &L.ev_tail, &L.ev_h, &L.ev_plan, &L.ev_g2})
            if (*e) { (void)hipEventDestroy(*e); *e = nullptr; }
        L.ntt_scratch.release();
        for (auto& b : L.calch_buf) b.release();
        for (auto& b : L.dist_buf) b.release();
        L.host_in[0].release(); L.host_in[1].release();
        L.witness.release(); L.h.release();
        (void)hipStreamDestroy(L.stream);
        (void)hipStreamDestroy(L.stream2);
        (void)hipStreamDestroy(L.stream3);
    }
    (void)hipStreamDestroy(g_ctx->stream);
    delete g_ctx;
    g_ctx = nullptr;
}

const std::string& device_info() { return g_devinfo; }

LaneLock acquire_lane(Context* C) {
    LaneLock r;
    r.C = C;
    for (;;) {
        for (int i = 0; i < C->n_lanes; i++) {
            std::unique_lock<std::mutex> lk(C->lanes[i].mu, std::try_to_lock);
            if (lk.owns_lock()) { r.L = &C->lanes[i]; r.lk = std::move(lk); return r; }
        }
        // every lane is busy: sleep until one is released (the timeout only bounds a wake-up lost between the scan above
        // and the wait below)
        std::unique_lock<std::mutex> w(C->lane_mu);
        C->lane_cv.wait_for(w, std::chrono::milliseconds(2));
    }
}
LaneLock::~LaneLock() {
    if (lk.owns_lock()) {
        lk.unlock();
        if (C) C->lane_cv.notify_one();
    }
}

// ---- staged uploads ----
// n persistent worker threads; run(n, fn) executes fn(0) on the caller and fn(1..n-1) on the pool, and returns when all are done.
// One job at a time (upload_staged holds its ring mutex around run()).  The threads live as long as the process.
struct StagePool {
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    std::vector<std::thread> th;
    const std::function<void(int)>* job = nullptr;
    unsigned long generation = 0;
    int pending = 0, active = 0;                  // workers 1 .. active-1 take part in the current job; the others stay parked
    void loop(int w) {
        unsigned long seen = 0;
        for (;;) {
            const std::function<void(int)>* f;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_go.wait(lk, [&] { return generation != seen; });
                seen = generation;
                if (w >= active) continue;          // a job with fewer workers than the pool has threads: not this one's
                f = job;
            }
            (*f)(w);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (--pending == 0) cv_done.notify_one();
            }
        }
    }
    void run(int n, const std::function<void(int)>& fn) {
        if ((int)th.size() < n - 1) {
            for (int w = (int)th.size() + 1; w < n; w++) { th.emplace_back([this, w] { loop(w); }); th.back().detach(); }
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            job = &fn;
            active = n;
            pending = n - 1;
            generation++;
        }
        cv_go.notify_all();
        fn(0);
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return pending == 0; });
    }
};

static const size_t PIN_CHUNK = (size_t)8 << 20;
static const int PIN_WORKERS = 8, PIN_PER_WORKER = 2;
int upload_staged(void* d_dst, const void* h_src, size_t bytes, hipStream_t s) {
    Context* C = ctx();
    if (!C) return WS_ERR_NOINIT;
    if (!s) s = C->stream;
    if (bytes == 0) return WS_OK;
#ifdef WSNARK_EMUL
    WS_HIP_CHECK(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, s));
    return WS_OK;
#else
    if (bytes < PIN_CHUNK / 2) {
        WS_HIP_CHECK(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, s));
        return WS_OK;
    }
    static std::mutex ring_mu;                      // one upload at a time uses the ring
    std::lock_guard<std::mutex> lk(ring_mu);
    if (!C->pin_ring) {
        WS_HIP_CHECK(hipHostMalloc(&C->pin_ring, PIN_CHUNK * PIN_WORKERS * PIN_PER_WORKER, 0));
        for (auto& e : C->pin_ev) WS_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    const int device = C->device;
    std::atomic<int> err(0);
    static StagePool* pool = new StagePool();       // never destroyed: its threads are parked on it until the process ends
    // Mode 1 (default; WSNARK_STAGE_MODE=0 = the per-piece scheme below, for A/B): MANY threads copy, FEW DMAs are issued.
    // Every hipMemcpyAsync costs tens of microseconds of runtime time on the stream's lock, so one DMA per copied piece made
    // the staging API-bound (32 MiB: 0.87 ms on the host whatever the worker count, profiles/r03_s8_stage_sweep.txt).  Here the
    // ring is two halves of 64 MiB; an upload goes through them in super-chunks, each cut into a few GROUPS: all workers copy
    // their slices of group g, the last one to arrive issues ONE DMA for the whole group and everybody moves on to group g + 1
    // while it runs.
    static const int stage_mode = [] { const char* e = getenv("WSNARK_STAGE_MODE"); return e ? atoi(e) : 1; }();
    if (stage_mode == 1) {
        const size_t HALF = PIN_CHUNK * PIN_WORKERS * PIN_PER_WORKER / 2;
        static const int env_groups = [] { const char* e = getenv("WSNARK_STAGE_GROUPS"); int v = e ? atoi(e) : 0; return v < 0 ? 0 : (v > 8 ? 8 : v); }();
        static const int env_w = [] { const char* e = getenv("WSNARK_STAGE_WORKERS"); int v = e ? atoi(e) : 0; return v < 0 ? 0 : (v > PIN_WORKERS ? PIN_WORKERS : v); }();
        int half = 0;
        for (size_t done = 0; done < bytes; half ^= 1) {
            const size_t sz = bytes - done < HALF ? bytes - done : HALF;
            const int G = env_groups ? env_groups : (sz >= ((size_t)48 << 20) ? 4 : sz >= ((size_t)12 << 20) ? 2 : 1);
            const int n_workers = env_w ? env_w : (sz >= ((size_t)8 << 20) ? PIN_WORKERS : 2);
            const size_t gs = ((sz + G - 1) / G + 0xFFFF) & ~(size_t)0xFFFF;          // group size, 64 KiB granules
            char* pin = (char*)C->pin_ring + (size_t)half * HALF;
            const char* src = (const char*)h_src + done;
            char* dst = (char*)d_dst + done;
            WS_HIP_CHECK(hipEventSynchronize(C->pin_ev[half]));                       // the half's previous DMAs have drained
            std::atomic<int> arrived[8];
            for (auto& a : arrived) a.store(0);
            auto worker = [&](int w) {
                if (hipSetDevice(device) != hipSuccess) { err = 1; return; }
                for (int g = 0; g < G; g++) {
                    const size_t g_lo = (size_t)g * gs, g_hi = g_lo + gs < sz ? g_lo + gs : sz;
                    if (g_lo < g_hi) {
                        const size_t per = (((g_hi - g_lo) + n_workers - 1) / n_workers + 4095) & ~(size_t)4095;
                        const size_t lo = g_lo + (size_t)w * per, hi = lo + per < g_hi ? lo + per : g_hi;
                        if (lo < hi) memcpy(pin + lo, src + lo, hi - lo);
                    }
                    if (arrived[g].fetch_add(1) == n_workers - 1 && g_lo < g_hi) {     // last one in: the group is staged
                        if (hipMemcpyAsync(dst + g_lo, pin + g_lo, g_hi - g_lo, hipMemcpyHostToDevice, s) != hipSuccess) err = 1;
                    }
                }
            };
            pool->run(n_workers, worker);
            if (err) { set_last_error("staged upload failed"); return WS_ERR_HIP; }
            WS_HIP_CHECK(hipEventRecord(C->pin_ev[half], s));
            done += sz;
        }
        return WS_OK;
    }
    // Mode 0: one DMA per copied piece.  The ring's slots are 8 MiB, but a 32 MiB witness cut into 8 MiB pieces gives every worker ONE piece --
    // no DMA starts before a whole 8 MiB memcpy is done and nothing overlaps.  Shorter pieces (a 32nd of the buffer,
    // 512 KiB .. 8 MiB) keep four pieces per worker in flight behind each other.
    // Swept on the MI355X box with a 32 MiB witness (profiles/r03_s8_stage_sweep.txt): the proof from a host witness costs
    // 1.0-1.4 ms more than from a resident one whatever the split -- best with two workers and one or two pieces each; a 0.6 GB
    // key section wants all eight (memcpy-bound: 10 ms).
    static const int env_workers = [] { const char* e = getenv("WSNARK_STAGE_WORKERS"); int v = e ? atoi(e) : 0; return v < 0 ? 0 : (v > PIN_WORKERS ? PIN_WORKERS : v); }();
    static const int env_pieces = [] { const char* e = getenv("WSNARK_STAGE_PIECES"); int v = e ? atoi(e) : 2; return v < 1 ? 1 : (v > 16 ? 16 : v); }();
    int n_workers = env_workers;
    if (!n_workers) { n_workers = (int)(bytes >> 24); n_workers = n_workers < 2 ? 2 : (n_workers > PIN_WORKERS ? PIN_WORKERS : n_workers); }   // one per 16 MiB, 2..8
    size_t chunk = (bytes / ((size_t)env_pieces * n_workers) + 0xFFFF) & ~(size_t)0xFFFF;
    chunk = chunk < ((size_t)512 << 10) ? ((size_t)512 << 10) : (chunk > PIN_CHUNK ? PIN_CHUNK : chunk);
    const size_t nchunks = (bytes + chunk - 1) / chunk;
    for (int b = 0; b < 2; b++) WS_HIP_CHECK(hipEventSynchronize(C->pin_ev[b]));     // (mode 1's half events share slots 0, 1)
    auto worker = [&](int w) {
        if (hipSetDevice(device) != hipSuccess) { err = 1; return; }
        int use = 0;
        for (size_t c = (size_t)w; c < nchunks; c += (size_t)n_workers, use++) {
            const int b = w * PIN_PER_WORKER + (use % PIN_PER_WORKER);
            char* pin = (char*)C->pin_ring + (size_t)b * PIN_CHUNK;
            // the buffer's previous DMA (this call or an earlier one) must have drained
            if (hipEventSynchronize(C->pin_ev[b]) != hipSuccess) { err = 1; return; }
            const size_t off = c * chunk, len = off + chunk <= bytes ? chunk : bytes - off;
            memcpy(pin, (const char*)h_src + off, len);
            if (hipMemcpyAsync((char*)d_dst + off, pin, len, hipMemcpyHostToDevice, s) != hipSuccess ||
                hipEventRecord(C->pin_ev[b], s) != hipSuccess) { err = 1; return; }
        }
    };
    // (the copy workers are PERSISTENT threads, created once and parked on a condition variable)
    pool->run(n_workers, worker);
    if (err) { set_last_error("staged upload failed"); return WS_ERR_HIP; }
    return WS_OK;
#endif
}

// ---- KernelTimer ----
static hipEvent_t timer_event(std::vector<hipEvent_t>& pool) {
    hipEvent_t e;
    if (!pool.empty()) { e = pool.back(); pool.pop_back(); return e; }
    (void)hipEventCreate(&e);
    return e;
}
static thread_local long t_rec = -1;      // this thread's open bracket (index into recs), -1 = none / filtered out
static thread_local unsigned long t_gen = 0;   // ... valid only while recs has not been folded away by collect()
void KernelTimer::begin(const char* name, hipStream_t s) {
    if (!enabled) return;
    t_rec = -1;
    if (dominant_only && strncmp(name, "msm_accumulate", 14) != 0) return;
    std::lock_guard<std::mutex> lk(mu);
    Rec r;
    r.name = name;          // string literals only
    r.a = timer_event(pool);
    r.b = timer_event(pool);
    (void)hipEventRecord(r.a, s);
    recs.push_back(r);
    t_rec = (long)recs.size() - 1;
    t_gen = generation;
}
void KernelTimer::end(hipStream_t s) {
    if (!enabled || t_rec < 0) return;
    std::lock_guard<std::mutex> lk(mu);
    if (t_gen == generation && (size_t)t_rec < recs.size()) (void)hipEventRecord(recs[(size_t)t_rec].b, s);
    t_rec = -1;
}
void KernelTimer::collect() {
    std::lock_guard<std::mutex> lk(mu);
    // WSNARK_TIMELINE=1: start/end of every bracket relative to the first one (both queues share the device clock)
    static const bool timeline = [] { const char* e = getenv("WSNARK_TIMELINE"); return e && atoi(e) == 1; }();
    for (auto& r : recs) {
        float ms = 0.f;
        (void)hipEventSynchronize(r.b);
        (void)hipEventElapsedTime(&ms, r.a, r.b);
        if (timeline && !recs.empty()) {
            float t0 = 0.f;
            (void)hipEventElapsedTime(&t0, recs.front().a, r.a);
            fprintf(stderr, "[wsnark timeline] %9.3f -> %9.3f  %s\n", t0, t0 + ms, r.name);
        }
        auto& a = acc[r.name];
        a.first += ms;
        a.second += 1;
        pool.push_back(r.a);
        pool.push_back(r.b);
    }
    recs.clear();
    generation++;
}
void KernelTimer::reset() {
    collect();
    std::lock_guard<std::mutex> lk(mu);
    acc.clear();
}

}  // namespace wsnark
