// hip_emul.cpp -- coroutine scheduler behind hip_emul.h.  TEST INFRASTRUCTURE ONLY.
#include "hip_emul.h"

#include <ucontext.h>

#include <atomic>
#include <mutex>
#include <vector>

// Sanitizer builds (make -C wasmsnark_amd/csrc emul SAN=address,undefined | SAN=thread): every kernel thread is a ucontext coroutine
// on a heap stack, so each switch is announced to the sanitizer's runtime -- ASan would otherwise take the new stack for a wild
// stack pointer (false stack-buffer-overflow reports, broken unwinding), TSan would attribute a coroutine's accesses to whatever
// ran on the host thread before.
#if defined(__SANITIZE_ADDRESS__)
#include <sanitizer/common_interface_defs.h>
#define EMUL_ASAN 1
#else
#define EMUL_ASAN 0
#endif
#if defined(__SANITIZE_THREAD__)
#include <sanitizer/tsan_interface.h>
#define EMUL_TSAN 1
#else
#define EMUL_TSAN 0
#endif

// queue / event handles (hip_emul.h): distinct, non-null, never dereferenced; the low six bits hold the device they were created on
int emul_device_count() {
    const char* e = getenv("WSNARK_EMUL_DEVICES");
    const int n = e ? atoi(e) : 1;
    return n < 1 ? 1 : n > 16 ? 16 : n;
}
int& emul_current_device() { static thread_local int d = 0; return d; }
hipError_t& emul_last_error() { static thread_local hipError_t e = hipSuccess; return e; }
hipStream_t emul_new_stream_handle() {
    static std::atomic<uintptr_t> next{0};
    return reinterpret_cast<hipStream_t>(((next.fetch_add(1) + 1) << 6) | (uintptr_t)emul_current_device());
}
hipEvent_t emul_new_event_handle() {
    static std::atomic<uintptr_t> next{0};
    return reinterpret_cast<hipEvent_t>(((next.fetch_add(1) + 1) << 6) | (uintptr_t)emul_current_device());
}

namespace hip_emul {

thread_local uint3_emul t_threadIdx, t_blockIdx;
thread_local dim3 t_blockDim, t_gridDim;

namespace {
const size_t STACK_BYTES = 256 * 1024;

struct Thr {
    ucontext_t uc;
    char* stack = nullptr;
    bool done = false;
    bool at_barrier = false;
    bool at_shfl = false;
    uint32_t shfl_val = 0;
    int shfl_src = 0;
    bool at_pair = false;      // waiting in pair_exchange for lane ^ pair_dist
    uint32_t pair_val = 0;
    int pair_dist = 1;
    void* asan_fake = nullptr; // ASan: this coroutine's fake-stack handle while it is switched out
    void* tsan_fiber = nullptr;
};

thread_local std::vector<Thr>* g_thr = nullptr;
thread_local ucontext_t g_sched;
thread_local int g_cur = -1;
thread_local const std::function<void()>* g_body = nullptr;
thread_local std::vector<unsigned char> g_smem;
thread_local std::vector<char*> g_stack_pool;
thread_local const void* g_sched_stack = nullptr;      // ASan: the scheduler's (host thread's) stack, learnt on the first switch
thread_local size_t g_sched_stack_size = 0;
thread_local void* g_sched_fiber = nullptr;            // TSan: the host thread's own fiber

// coroutine -> scheduler (from trampoline's end with `last`: the coroutine never runs again)
void yield_to_scheduler(Thr& me, bool last = false) {
#if EMUL_ASAN
    __sanitizer_start_switch_fiber(last ? nullptr : &me.asan_fake, g_sched_stack, g_sched_stack_size);
#endif
#if EMUL_TSAN
    __tsan_switch_to_fiber(g_sched_fiber, 0);
#endif
    swapcontext(&me.uc, &g_sched);
#if EMUL_ASAN
    __sanitizer_finish_switch_fiber(me.asan_fake, &g_sched_stack, &g_sched_stack_size);      // (back in the coroutine)
#endif
    (void)last;
}
// scheduler -> coroutine
void resume(Thr& T) {
#if EMUL_ASAN
    void* fake = nullptr;
    __sanitizer_start_switch_fiber(&fake, T.stack, STACK_BYTES);
#endif
#if EMUL_TSAN
    __tsan_switch_to_fiber(T.tsan_fiber, 0);
#endif
    swapcontext(&g_sched, &T.uc);
#if EMUL_ASAN
    __sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
#endif
}

void trampoline() {
#if EMUL_ASAN
    __sanitizer_finish_switch_fiber(nullptr, &g_sched_stack, &g_sched_stack_size);      // first entry: nothing to restore
#endif
    (*g_body)();
    (*g_thr)[g_cur].done = true;
    yield_to_scheduler((*g_thr)[g_cur], true);
}
void set_ids(int t, dim3 block) {
    t_threadIdx.x = (unsigned)t % block.x;
    t_threadIdx.y = ((unsigned)t / block.x) % block.y;
    t_threadIdx.z = (unsigned)t / (block.x * block.y);
}
}  // namespace

void* dyn_smem() { return g_smem.data(); }

void sync_threads() {
    Thr& me = (*g_thr)[g_cur];
    me.at_barrier = true;
    yield_to_scheduler(me);
}

uint32_t shfl_exchange(uint32_t v, int src_lane, int /*width*/) {
    Thr& me = (*g_thr)[g_cur];
    me.shfl_val = v;
    me.shfl_src = src_lane;
    me.at_shfl = true;
    yield_to_scheduler(me);          // scheduler resumes us once the whole wave has posted
    return me.shfl_val;
}

// exchange with lane ^ 1 only (a DPP quad_perm swap on the device): unlike the wave shuffles above it may be called from code
// that only SOME lane pairs of the wave execute -- both lanes of a pair always take the same branch
uint32_t pair_exchange(uint32_t v) { return pair_exchange_dist(v, 1); }
uint32_t pair_exchange_dist(uint32_t v, int dist) {
    Thr& me = (*g_thr)[g_cur];
    me.pair_val = v;
    me.pair_dist = dist;
    me.at_pair = true;
    yield_to_scheduler(me);          // the scheduler resumes us once the partner has posted
    return me.pair_val;
}

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
    // __shared__ arrays are process-wide statics here: kernels of different host threads (two lanes) take turns
    static std::mutex one_kernel_at_a_time;
    std::lock_guard<std::mutex> turn(one_kernel_at_a_time);
    const int nthr = (int)(block.x * block.y * block.z);
    std::vector<Thr> thr((size_t)nthr);
    while ((int)g_stack_pool.size() < nthr) g_stack_pool.push_back((char*)malloc(STACK_BYTES));
    g_smem.assign(shmem + 64, 0);
    g_thr = &thr;
    g_body = &body;
    t_blockDim = block;
    t_gridDim = grid;
    for (unsigned bz = 0; bz < grid.z; bz++)
    for (unsigned by = 0; by < grid.y; by++)
    for (unsigned bx = 0; bx < grid.x; bx++) {
        t_blockIdx.x = bx; t_blockIdx.y = by; t_blockIdx.z = bz;
        for (int t = 0; t < nthr; t++) {
            Thr& T = thr[(size_t)t];
            T.done = false; T.at_barrier = false; T.at_shfl = false; T.at_pair = false;
            getcontext(&T.uc);
            T.stack = g_stack_pool[(size_t)t];
            T.uc.uc_stack.ss_sp = T.stack;
            T.uc.uc_stack.ss_size = STACK_BYTES;
            T.uc.uc_link = &g_sched;
            makecontext(&T.uc, trampoline, 0);
#if EMUL_TSAN
            if (!g_sched_fiber) g_sched_fiber = __tsan_get_current_fiber();
            T.tsan_fiber = __tsan_create_fiber(0);
#endif
        }
        int live = nthr;
        while (live > 0) {
            // run every runnable thread until it finishes or blocks
            for (int t = 0; t < nthr; t++) {
                Thr& T = thr[(size_t)t];
                if (T.done || T.at_barrier || T.at_shfl || T.at_pair) continue;
                g_cur = t;
                set_ids(t, block);
                resume(T);
                if (T.done) {
                    live--;
#if EMUL_TSAN
                    __tsan_destroy_fiber(T.tsan_fiber);
                    T.tsan_fiber = nullptr;
#endif
                }
            }
            bool progressed = false;
            // resolve pair exchanges: lanes 2k and 2k + 1 swap once both have posted
            for (int t = 0; t < nthr; t++) {
                Thr& A = thr[(size_t)t];
                if (!A.at_pair || (t & A.pair_dist)) continue;          // (the lower lane of a pair does the swap)
                const int u = t ^ A.pair_dist;
                if (u >= nthr) continue;
                Thr& B = thr[(size_t)u];
                if (B.at_pair && B.pair_dist == A.pair_dist) {
                    const uint32_t x = A.pair_val;
                    A.pair_val = B.pair_val; B.pair_val = x;
                    A.at_pair = B.at_pair = false;
                    progressed = true;
                }
            }
            // resolve wave shuffles: a wave proceeds when all its live lanes posted
            for (int w0 = 0; w0 < nthr; w0 += 64) {
                int w1 = w0 + 64 < nthr ? w0 + 64 : nthr;
                bool any = false, all = true;
                for (int t = w0; t < w1; t++) {
                    if (thr[(size_t)t].done) continue;
                    if (thr[(size_t)t].at_shfl) any = true; else all = false;      // (a lane waiting for its pair partner has not posted either)
                }
                if (any && all) {
                    uint32_t vals[64];
                    for (int t = w0; t < w1; t++) vals[t - w0] = thr[(size_t)t].shfl_val;
                    for (int t = w0; t < w1; t++) {
                        Thr& T = thr[(size_t)t];
                        if (T.done) continue;
                        int src = T.shfl_src & 63;
                        if (w0 + src < w1) T.shfl_val = vals[src];
                        T.at_shfl = false;
                    }
                    progressed = true;
                }
            }
            if (progressed) continue;
            // barrier: release when every live thread is waiting at it
            bool all_bar = live > 0;
            for (int t = 0; t < nthr; t++) if (!thr[(size_t)t].done && !thr[(size_t)t].at_barrier) { all_bar = false; break; }
            if (all_bar) for (int t = 0; t < nthr; t++) thr[(size_t)t].at_barrier = false;
            else if (live > 0) {
                bool runnable = false;
                for (int t = 0; t < nthr; t++) if (!thr[(size_t)t].done && !thr[(size_t)t].at_barrier && !thr[(size_t)t].at_shfl && !thr[(size_t)t].at_pair) runnable = true;
                if (!runnable) abort();   // deadlock: divergent barrier / partial-wave shuffle
            }
        }
    }
    g_thr = nullptr;
    g_body = nullptr;
}

}  // namespace hip_emul
