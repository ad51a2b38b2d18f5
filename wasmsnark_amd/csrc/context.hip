// context.hip -- library context, error string, per-kernel HIP-event timing.
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <vector>
#include <thread>
#include <memory>
#include <unistd.h>

#include <chrono>

#include "internal.h"

namespace wsnark {

static thread_local std::string t_last_error;
void set_last_error(const std::string& s) { t_last_error = s; }
const std::string& get_last_error() { return t_last_error; }

static Context* g_ctx = nullptr;                  // the default context (wsnark_init)
static std::mutex g_ctx_mu;
static thread_local Context* t_ctx = nullptr;      // the calling thread's selection (CtxScope); nullptr = the default context

Context* ctx() { return t_ctx ? t_ctx : g_ctx; }
Context* ctx_set_current(Context* c) { Context* p = t_ctx; t_ctx = c; return p; }

// Switches: an override set through wsnark_tuning_set wins; otherwise the environment variable WSNARK_<name> AS IT WAS WHEN THE
// NAME WAS FIRST ASKED FOR (one getenv per name and process, under the mutex: prover threads never read the environment while a
// host -- Python's os.environ[...] = ..., Node's process.env -- may be writing it, which is undefined behaviour in glibc);
// otherwise the default.  A process that wants to change a switch after its first use calls wsnark_tuning_set.
static std::mutex g_tune_mu;
static std::map<std::string, long> g_tune;                       // overrides
static std::map<std::string, std::pair<bool, long>> g_env;       // name -> (set in the environment, value), filled on first use
// Fast path: the switches are read many times per proof from every prover thread, so a reader first looks into a small
// per-thread table keyed by the NAME'S ADDRESS (every caller passes a string literal) and stamped with the generation of the
// override map; only a miss -- first use on this thread, or any wsnark_tuning_set since -- takes the mutex and the two map lookups.
// Nothing caches a switch in a `static` any more, so wsnark_tuning_set reaches every reader with its next call.
static std::atomic<uint64_t> g_tune_gen{1};
static long tuning_lookup(const char* name, bool* is_set) {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    auto it = g_tune.find(name);
    if (it != g_tune.end()) { *is_set = true; return it->second; }
    auto ie = g_env.find(name);
    if (ie == g_env.end()) {
        const std::string env = std::string("WSNARK_") + name;
        const char* e = getenv(env.c_str());
        ie = g_env.emplace(name, std::make_pair(e != nullptr, e ? atol(e) : 0L)).first;
    }
    *is_set = ie->second.first;
    return ie->second.second;
}
long tuning_get(const char* name, long dflt) {
    struct Slot { const char* name; uint64_t gen; long value; bool is_set; };
    static thread_local Slot cache[64];
    const uint64_t gen = g_tune_gen.load(std::memory_order_acquire);
    Slot& sl = cache[((uintptr_t)name >> 2) & 63];
    if (sl.name != name || sl.gen != gen) {
        bool is_set = false;
        const long v = tuning_lookup(name, &is_set);
        sl = Slot{name, gen, v, is_set};
    }
    return sl.is_set ? sl.value : dflt;
}
void tuning_set(const char* name, long value) {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    if (value == LONG_MIN) g_tune.erase(name);
    else g_tune[name] = value;
    g_tune_gen.fetch_add(1, std::memory_order_release);
}
// WSNARK_<name> as a fraction-capable number (e.g. WSNARK_TABLE_MAX_GB=0.5): the environment value is parsed as a double once; an
// override through wsnark_tuning_set (integers only) wins.  Negative values read as 0.
double tuning_get_real(const char* name, double dflt) {
    {
        std::lock_guard<std::mutex> lk(g_tune_mu);
        auto it = g_tune.find(name);
        if (it != g_tune.end()) return it->second < 0 ? 0.0 : (double)it->second;
    }
    static std::mutex mu;
    static std::map<std::string, std::pair<bool, double>> env;
    std::lock_guard<std::mutex> lk(mu);
    auto ie = env.find(name);
    if (ie == env.end()) {
        const std::string key = std::string("WSNARK_") + name;
        const char* e = getenv(key.c_str());
        ie = env.emplace(name, std::make_pair(e != nullptr, e ? atof(e) : 0.0)).first;
    }
    if (!ie->second.first) return dflt;
    return ie->second.second < 0 ? 0.0 : ie->second.second;
}

static int ensure_ring(Context* C);
// One context on one device: its queues, lanes and helper threads.  device < 0: LOCAL_RANK, else 0.
static int context_create_impl(int device, bool wrap, Context* C);
int context_create(int device, Context** out, bool wrap) {
    Context* C = new Context();
    const int rc = context_create_impl(device, wrap, C);
    if (rc != WS_OK) {                 // (an error return used to delete the Context and leak the queues created so far)
        const std::string why = get_last_error();
        context_destroy(C);
        set_last_error(why);
        return rc;
    }
    *out = C;
    return WS_OK;
}
// device < 0: LOCAL_RANK, else 0.  wrap (the default context only: torchrun-style launches on hosts that show each rank one GPU):
// ordinals beyond the device count wrap around; a GROUP's ordinals must exist (two contexts on one GPU are asked for by repeating
// the ordinal, never by naming a device that is not there).
static int context_create_impl(int device, bool wrap, Context* C) {
    if (device < 0) {
        const char* lr = getenv("LOCAL_RANK");
        device = lr ? atoi(lr) : 0;
    }
    int count = 0;
    WS_HIP_CHECK(hipGetDeviceCount(&count));
    if (count <= 0) { set_last_error("no HIP device visible"); return WS_ERR_HIP; }
    if (device >= count) {
        if (!wrap) { set_last_error("device ordinal " + std::to_string(device) + " does not exist (" + std::to_string(count) + " visible)"); return WS_ERR_ARG; }
        device = device % count;
    }
    WS_HIP_CHECK(hipSetDevice(device));
    C->device = device;
    C->owner_pid = (int)getpid();
    hipDeviceProp_t prop;
    WS_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    C->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    C->devinfo = std::string(prop.name) + " " + prop.gcnArchName + " CUs=" + std::to_string(C->num_cu) + " device=" + std::to_string(device);
    WS_HIP_CHECK(hipStreamCreateWithFlags(&C->stream, hipStreamNonBlocking));
    {
        const long nl = tuning_get("LANES", 2);
        C->n_lanes = nl < 1 ? 1 : nl > kMaxLanes ? kMaxLanes : (int)nl;
    }
    // Plain queues (one priority): high-priority second / third queues were measured in rounds 2 and 5 (profiles/r02_sweep_prove_overlap.txt,
    // r05_schedule_experiments.txt) and change nothing -- a kernel's workgroups get SIMD slots as the other queue's workgroups retire,
    // whatever the queue's priority.
    for (int i = 0; i < C->n_lanes; i++) {
        Lane& L = C->lanes[i];
        L.id = i;
        WS_HIP_CHECK(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
        WS_HIP_CHECK(hipStreamCreateWithFlags(&L.stream2, hipStreamNonBlocking));
        WS_HIP_CHECK(hipStreamCreateWithFlags(&L.stream3, hipStreamNonBlocking));
        WS_HIP_CHECK(hipStreamCreateWithFlags(&L.stream_copy, hipStreamNonBlocking));
    }
    for (auto& q : C->load_q) WS_HIP_CHECK(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
    // (the LOWEST stream priority: the runtime multiplexes a process's streams onto a few hardware queues per priority class, and a
    //  proof whose queue shared one with a normal-priority build would sit behind 130 ms of table kernels -- seen through the Node
    //  addon: first proof 133 ms instead of 13; at the lowest priority the builds have a queue of their own and yield to proofs.
    //  Created HERE, once: no queue is ever created while proofs may be in flight on other threads)
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess || hipStreamCreateWithPriority(&C->build_q, hipStreamNonBlocking, lo) != hipSuccess) {
        (void)hipGetLastError();
        C->build_q = nullptr;
        WS_HIP_CHECK(hipStreamCreateWithFlags(&C->build_q, hipStreamNonBlocking));
    }
#ifndef WSNARK_EMUL
    // Two helper threads do what the first key load and proof on this context would otherwise wait for: one pins the staging ring
    // (8-26 ms), the other launches one no-op kernel per translation unit, which makes the runtime load that unit's code object
    // (7-12 ms for msm.hip alone under ROCm 7.2).  The call returns at once; whoever needs the ring or a kernel first finds the
    // work done or in progress (the ring is created under its mutex, module loads are serialised by the runtime).
    // WSNARK_INIT_WARM=0: on first use.
    if (tuning_get("INIT_WARM", 1)) {
        Context* P = C;
        C->warm_ring = new std::thread([P]() {
            if (hipSetDevice(P->device) != hipSuccess) return;
            std::lock_guard<std::mutex> lk(P->ring_mu);
            if (ensure_ring(P) != WS_OK) (void)hipGetLastError();      // (not fatal: the first upload will say)
        });
        hipStream_t q = C->stream;      // (the utility queue: the launch itself is what loads the code object -- nothing to wait for)
        const int dev = device;
        C->warm = new std::thread([dev, q]() {
            if (hipSetDevice(dev) != hipSuccess) return;
            warm_msm(q); warm_calch(q); warm_ntt(q); warm_fixedbase(q); warm_dist(q);
        });
    }
#endif
    return WS_OK;
}
void context_join_warm(Context* C) {
    std::lock_guard<std::mutex> lk(C->warm_mu);
    for (std::thread** t : {&C->warm, &C->warm_ring}) {
        if (!*t) continue;
        // (a fork()ed child inherits the objects but none of the threads: joining would never return -- they are left behind)
        if (C->owner_pid == (int)getpid()) { if ((*t)->joinable()) (*t)->join(); delete *t; }
        *t = nullptr;
    }
}

void context_destroy(Context* C) {
    if (!C) return;
    context_join_warm(C);
    (void)hipSetDevice(C->device);
    (void)hipStreamSynchronize(C->stream);
    for (auto& q : C->load_q) if (q) { (void)hipStreamSynchronize(q); (void)hipStreamDestroy(q); q = nullptr; }
    if (C->build_q) { (void)hipStreamSynchronize(C->build_q); (void)hipStreamDestroy(C->build_q); C->build_q = nullptr; }
    C->build_tmp.release();
    for (int i = 0; i < C->n_lanes; i++) {
        (void)hipStreamSynchronize(C->lanes[i].stream);
        (void)hipStreamSynchronize(C->lanes[i].stream2);
        (void)hipStreamSynchronize(C->lanes[i].stream3);
        (void)hipStreamSynchronize(C->lanes[i].stream_copy);
    }
    C->timer.reset();
    for (hipEvent_t e : C->timer.pool) (void)hipEventDestroy(e);
    C->timer.pool.clear();
    C->ntt_plans.clear();
    if (C->pin_ring) { (void)hipHostFree(C->pin_ring); C->pin_ring = nullptr; }
    for (auto& e : C->pin_ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    for (int i = 0; i < C->n_lanes; i++) {
        Lane& L = C->lanes[i];
        msm_workspace_free(L);
        for (hipEvent_t* e : {&L.ntt_chain.done, &L.calch_chain.done, &L.ev_start, &L.ev_tail, &L.ev_h, &L.ev_plan, &L.ev_g2, &L.ev_chunk[0], &L.ev_chunk[1]})
            if (*e) { (void)hipEventDestroy(*e); *e = nullptr; }
        L.ntt_scratch.release();
        for (auto& b : L.calch_buf) b.release();
        for (auto& b : L.dist_buf) b.release();
        L.host_in[0].release(); L.host_in[1].release();
        L.witness.release(); L.h.release();
        (void)hipStreamDestroy(L.stream);
        (void)hipStreamDestroy(L.stream2);
        (void)hipStreamDestroy(L.stream3);
        (void)hipStreamDestroy(L.stream_copy);
    }
    (void)hipStreamDestroy(C->stream);
    delete C;
}

int context_init(int device) {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    if (g_ctx) return WS_OK;
    Context* C = nullptr;
    int rc = context_create(device, &C, /*wrap=*/true);
    if (rc) return rc;
    g_ctx = C;
    return WS_OK;
}
void context_shutdown() {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    if (!g_ctx) return;
    Context* C = g_ctx;
    g_ctx = nullptr;
    context_destroy(C);
}

const std::string& device_info() {       // of the calling thread's context (the default one unless a group call selected another)
    static const std::string none;
    Context* C = ctx();
    return C ? C->devinfo : none;
}

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
// Host -> device copies of caller memory that may be pageable and never touched by the runtime before (the witness and key
// bytes of the reference's callers, src/bn128.js:580).  The runtime's own pageable path pins fresh pages at ~10 GB/s; here
// worker threads memcpy CHUNKS into a pinned ring and the calling thread -- the orchestrator -- issues one DMA per chunk as soon
// as the chunk is staged, so that copying chunk g + 1 overlaps the DMA of chunk g.  Round 4: (a) the orchestrator no longer
// copies (a 32 MiB witness used to be staged in 0.87 ms before its last DMA could start: 38 GB/s of memcpy against ~50 GB/s
// of PCIe -- the staging, not the link, set the pace), (b) chunks are 4 MiB instead of 16, (c) every chunk is announced to the
// caller (`on_chunk`, on the orchestrator thread, after its DMA has been queued on `s`), which lets the prover start the first
// pass over the witness -- the digit histogram of the grouping pass -- while the rest is still on its way, and (d) a source that
// is ALREADY pinned (hipHostMalloc / hipHostRegister: the N-API addon's external ArrayBuffers) is DMA'd in place.
// The pool threads never call into the HIP runtime.  Ring, slot events and pool belong to the CONTEXT (round 5): the devices of a
// group stage their uploads side by side.
struct StagePool {
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    std::vector<std::thread> th;
    const std::function<void(int)>* job = nullptr;
    unsigned long generation = 0;
    int pending = 0, active = 0;                  // workers 0 .. active-1 take part in the current job; the others stay parked
    pid_t owner = 0;                              // the process the threads live in: a fork()ed child has the vector but not the threads
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
    // starts fn(0 .. n-1) on the pool and returns at once; wait() returns when all are done
    void start(int n, const std::function<void(int)>& fn) {
        for (int w = (int)th.size(); w < n; w++) { th.emplace_back([this, w] { loop(w); }); th.back().detach(); }
        {
            std::lock_guard<std::mutex> lk(mu);
            job = &fn;
            active = n;
            pending = n;
            generation++;
        }
        cv_go.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return pending == 0; });
    }
};
// One pool per context and process: after fork() (Python multiprocessing with the "fork" start method, Node child workers) the
// child inherits the object -- thread handles, possibly a locked mutex -- but none of the threads, so it gets a pool of its own.
// Never destroyed: the threads are parked on it until the process ends (the library must not be dlclose'd).  Caller holds ring_mu.
static StagePool* stage_pool(Context* C) {
    if (!C->pool || C->pool->owner != getpid()) { C->pool = new StagePool(); C->pool->owner = getpid(); }
    return C->pool;
}

static const int PIN_MAX_WORKERS = 24, PIN_MAX_SLOTS = 128;
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
}

// The pinned staging ring (and its slot events): created by the context's helper thread on a device build -- pinning 128 MiB takes
// 8-26 ms, which used to sit inside the first key load of a process -- and on first use otherwise (the emulator's tests size it
// per test).  Caller holds C->ring_mu.
static int ensure_ring(Context* C) {
    if (C->pin_ring) return WS_OK;
    size_t ring = (size_t)tuning_get("STAGE_RING_KB", 128 << 10) << 10;        // (read once, when the ring is created)
    ring = ring < ((size_t)1 << 20) ? ((size_t)1 << 20) : ring > ((size_t)1 << 30) ? ((size_t)1 << 30) : ring;
    const auto t_ring = std::chrono::steady_clock::now();
    WS_HIP_CHECK(hipHostMalloc(&C->pin_ring, ring, 0));
    if (tuning_get("TRACE", 0) == 1)
        fprintf(stderr, "[wsnark trace] staging ring: %zu MiB of pinned host memory in %.2f ms\n", ring >> 20,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_ring).count());
    C->pin_ring_bytes = ring;
    for (auto& e : C->pin_ev) WS_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return WS_OK;
}

// The ring's slot events are recorded on the queue an upload names, and the runtime keeps that queue's address inside the event
// (hipEventRecord stores it, hipEventQuery / hipEventSynchronize read the queue's capture state through it -- libamdhip64 7.2): an
// event must never outlive the queue it was last recorded on.  Both belong to ONE context here, which holds as long as an upload runs
// on the context whose queue it names -- checked, because a helper thread once did not (prove.hip, the transposition of matrix B).
static bool queue_of_context(const Context* C, hipStream_t s) {
    if (s == C->stream || s == C->build_q) return true;
    for (const auto& q : C->load_q) if (s == q) return true;
    for (int i = 0; i < C->n_lanes; i++) {
        const Lane& L = C->lanes[i];
        if (s == L.stream || s == L.stream2 || s == L.stream3 || s == L.stream_copy) return true;
    }
    return false;
}

int upload_pipelined(void* d_dst, const void* h_src, size_t bytes, hipStream_t s, const ChunkFn& on_chunk, bool* direct_out) {
    Context* C = ctx();
    if (!C) return WS_ERR_NOINIT;
    if (!s) s = C->stream;
    if (!queue_of_context(C, s)) { set_last_error("staged upload: the queue belongs to another context than the one this thread has selected"); return WS_ERR_ARG; }
    if (direct_out) *direct_out = false;
    if (bytes == 0) return WS_OK;
    // chunk size: WSNARK_STAGE_CHUNK_KB (default 4 MiB; 64 KiB granules, 64 KiB .. 16 MiB).  Measured on the MI355X box with a
    // 32 MiB witness (profiles/r04_s3_upload_sweep.txt): 4 MiB chunks and 4 copy threads cost the proof +0.95 ms over a resident
    // witness, 16 MiB chunks +1.17, 8 threads +1.06, a pinned source DMA'd in place +0.97 -- of which 0.63 ms is the transfer
    // itself at the link's 53.5 GB/s (one copy queue: two move no more, the link is the limit); 12 threads +1.6 ms.
    size_t chunk = ((size_t)tuning_get("STAGE_CHUNK_KB", 4096) << 10) & ~(size_t)0xFFFF;
    chunk = chunk < ((size_t)64 << 10) ? ((size_t)64 << 10) : chunk > ((size_t)16 << 20) ? ((size_t)16 << 20) : chunk;
    int rc = WS_OK;
    // small copies, and sources that are ALREADY pinned (hipHostMalloc / hipHostRegister), need no staging
    bool direct = bytes < ((size_t)1 << 20), pinned = false;
#ifdef WSNARK_EMUL
    direct = !tuning_get("STAGE_FORCE_RING", 0);    // (the emulator's "device" memory is host memory; tests force the ring to run its bookkeeping)
#else
    if (!direct || direct_out) {                    // (small copies go up directly whatever their source: no query -- ADVICE r5)
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, h_src) == hipSuccess) pinned = at.type == hipMemoryTypeHost;
        else (void)hipGetLastError();               // pageable memory is "invalid value" to the runtime: not an error here
        direct = direct || pinned;
    }
#endif
    if (direct) {
        // A pinned source goes up in ONE DMA: measured on the MI355X box with the 32 MiB witness (profiles/r04_s6_pinned_sweep.txt),
        // proof from the pinned buffer minus proof from a resident witness: one DMA +0.73-0.87 ms, 4 MiB pieces with the histogram per
        // piece +0.95-0.97 -- every copy command has its own start-up, and the 0.06 ms of histogram it would hide is less than that.
        chunk = bytes;
#ifdef WSNARK_EMUL
        chunk = ((size_t)tuning_get("STAGE_CHUNK_KB", 4096) << 10) & ~(size_t)0xFFFF;       // (tests: the chunked bookkeeping on small inputs)
        if (chunk < ((size_t)64 << 10)) chunk = (size_t)64 << 10;
#endif
        if (direct_out) *direct_out = pinned;       // (a small pageable source has been staged by the runtime when hipMemcpyAsync returns)
        const size_t G = (bytes + chunk - 1) / chunk;
        for (size_t g = 0; g < G; g++) {
            const size_t lo = g * chunk, hi = lo + chunk < bytes ? lo + chunk : bytes;
            WS_HIP_CHECK(hipMemcpyAsync((char*)d_dst + lo, (const char*)h_src + lo, hi - lo, hipMemcpyHostToDevice, s));
            if (on_chunk && (rc = on_chunk(lo, hi, s))) return rc;
        }
        return WS_OK;
    }
    std::lock_guard<std::mutex> lk(C->ring_mu);     // one upload at a time uses the context's ring
    if ((rc = ensure_ring(C))) return rc;
    if (chunk > C->pin_ring_bytes / 2) chunk = (C->pin_ring_bytes / 2) & ~(size_t)0xFFFF;
    size_t nslots = C->pin_ring_bytes / chunk;
    if (nslots > (size_t)PIN_MAX_SLOTS) nslots = PIN_MAX_SLOTS;
    // A slot's event guards the bytes [slot * chunk, (slot + 1) * chunk) of the ring ONLY under the chunk size it was recorded
    // with: an upload with another chunk size lays its slots over different bytes (two 4 MiB slots of the previous upload under one
    // 8 MiB slot of this one), so it first waits for everything the previous geometry still has in flight.  (Found the hard way:
    // a points shard's 8 MiB sections went up in 4 MiB chunks, the 16 MiB one behind them in 8 MiB chunks -- and overwrote the second
    // half of its predecessor before that had left the ring.)
    if (C->pin_chunk_last != chunk) {
        for (int i = 0; i < PIN_MAX_SLOTS; i++)
            if (C->pin_ev_rec[i] && hipEventSynchronize(C->pin_ev[i]) != hipSuccess) { set_last_error("staged upload: a ring slot's DMA failed"); return WS_ERR_HIP; }
        C->pin_chunk_last = chunk;
    }
    const size_t G = (bytes + chunk - 1) / chunk;
    // workers: WSNARK_STAGE_WORKERS, default 4 (more only get in each other's way: the sweep above), 2 below 8 MiB
    const long env_w = tuning_get("STAGE_WORKERS", 0);
    int W = env_w > 0 ? (int)env_w : (bytes < ((size_t)8 << 20) ? 2 : 4);
    W = W < 1 ? 1 : W > PIN_MAX_WORKERS ? PIN_MAX_WORKERS : W;
    const size_t per = ((chunk + W - 1) / W + 4095) & ~(size_t)4095;      // a worker's slice of a chunk, page granules
    // Bookkeeping.  Chunk g lives in slot g % nslots.  Workers may write chunk g once released > g; the orchestrator releases
    // chunk k only when the slot's previous occupant -- chunk k - nslots of THIS upload (whose DMA it has queued itself, so the
    // slot's event is the fresh one), or an earlier upload's -- has drained.
    std::atomic<size_t> released(0);
    std::unique_ptr<std::atomic<int>[]> arrived(new std::atomic<int>[G]);
    for (size_t g = 0; g < G; g++) arrived[g].store(0, std::memory_order_relaxed);
    std::atomic<int> stop(0);
    char* ring = (char*)C->pin_ring;
    const std::function<void(int)> worker = [&](int w) {
        for (size_t g = 0; g < G; g++) {
            unsigned spins = 0;
            while (released.load(std::memory_order_acquire) <= g) {
                if (stop.load(std::memory_order_relaxed)) return;
                if (++spins > 2000) std::this_thread::yield(); else cpu_relax();
            }
            if (stop.load(std::memory_order_relaxed)) return;
            const size_t c_lo = g * chunk, c_hi = c_lo + chunk < bytes ? c_lo + chunk : bytes;
            const size_t lo = c_lo + (size_t)w * per, hi = lo + per < c_hi ? lo + per : c_hi;
            if (lo < hi) memcpy(ring + (g % nslots) * chunk + (lo - c_lo), (const char*)h_src + lo, hi - lo);
            arrived[g].fetch_add(1, std::memory_order_release);
        }
    };
    StagePool* pool = stage_pool(C);
    pool->start(W, worker);
    size_t next = 0;                                // next chunk to release
    // release chunks below `limit`; block = wait for the first one's slot instead of giving up when it is still draining
    auto release_upto = [&](size_t limit, bool block) -> int {
        while (next < limit) {
            hipEvent_t ev = C->pin_ev[next % nslots];
            if (!C->pin_ev_rec[next % nslots]) {           // a slot no DMA has left yet: free
                released.store(++next, std::memory_order_release);
                block = false;
                continue;
            }
            if (block) {
                if (hipEventSynchronize(ev) != hipSuccess) { set_last_error("staged upload: a ring slot's DMA failed"); return WS_ERR_HIP; }
            } else {
                const hipError_t q = hipEventQuery(ev);
                if (q == hipErrorNotReady) { (void)hipGetLastError(); return WS_OK; }
                if (q != hipSuccess) { set_last_error(std::string("staged upload: ") + hipGetErrorString(q)); return WS_ERR_HIP; }
            }
            released.store(++next, std::memory_order_release);
            block = false;
        }
        return WS_OK;
    };
    for (size_t g = 0; g < G && !rc; g++) {
        // chunks < g have their DMA queued, so slots of chunks < g + nslots carry fresh events (or none from this upload)
        const size_t limit = g + nslots < G ? g + nslots : G;
        unsigned spins = 0;
        while (!rc && arrived[g].load(std::memory_order_acquire) < W) {
            if (next <= g) rc = release_upto(limit, true);               // the workers have nothing they may write: wait for the slot
            else if (next < limit && (spins & 31) == 0) rc = release_upto(limit, false);
            if (++spins > 4000) std::this_thread::yield(); else cpu_relax();
        }
        if (rc) break;
        const size_t lo = g * chunk, hi = lo + chunk < bytes ? lo + chunk : bytes;
        const size_t slot = g % nslots;
        if (hipMemcpyAsync((char*)d_dst + lo, ring + slot * chunk, hi - lo, hipMemcpyHostToDevice, s) != hipSuccess ||
            hipEventRecord(C->pin_ev[slot], s) != hipSuccess) { set_last_error("staged upload: DMA failed"); rc = WS_ERR_HIP; break; }
        C->pin_ev_rec[slot] = true;
        if (on_chunk) rc = on_chunk(lo, hi, s);
    }
    if (rc) stop.store(1);                          // the workers leave at their next chunk boundary
    pool->wait();
    return rc;
}

int upload_staged(void* d_dst, const void* h_src, size_t bytes, hipStream_t s) {
    bool direct = false;
    int rc = upload_pipelined(d_dst, h_src, bytes, s, nullptr, &direct);
    // (a source DMA'd in place is still being read: wait, so that the caller may free or rewrite it -- as after a staged upload)
    if (!rc && direct) {
        Context* C = ctx();
        WS_HIP_CHECK(hipStreamSynchronize(s ? s : C->stream));
    }
    return rc;
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
    const bool timeline = tuning_get("TIMELINE", 0) == 1;
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
