// hip_emul.cpp -- coroutine scheduler behind hip_emul.h.  TEST INFRASTRUCTURE ONLY.
#include "hip_emul.h"

#include <ucontext.h>

#include <mutex>
#include <vector>

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
    bool at_pair = false;      // waiting in pair_exchange for lane ^ 1
    uint32_t pair_val = 0;
};

thread_local std::vector<Thr>* g_thr = nullptr;
thread_local ucontext_t g_sched;
thread_local int g_cur = -1;
thread_local const std::function<void()>* g_body = nullptr;
thread_local std::vector<unsigned char> g_smem;
thread_local std::vector<char*> g_stack_pool;

void trampoline() {
    (*g_body)();
    (*g_thr)[g_cur].done = true;
    swapcontext(&(*g_thr)[g_cur].uc, &g_sched);
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
    swapcontext(&me.uc, &g_sched);
}

uint32_t shfl_exchange(uint32_t v, int src_lane, int /*width*/) {
    Thr& me = (*g_thr)[g_cur];
    me.shfl_val = v;
    me.shfl_src = src_lane;
    me.at_shfl = true;
    swapcontext(&me.uc, &g_sched);   // scheduler resumes us once the whole wave has posted
    return me.shfl_val;
}

// exchange with lane ^ 1 only (a DPP quad_perm swap on the device): unlike the wave shuffles above it may be called from code
// that only SOME lane pairs of the wave execute -- both lanes of a pair always take the same branch
uint32_t pair_exchange(uint32_t v) {
    Thr& me = (*g_thr)[g_cur];
    me.pair_val = v;
    me.at_pair = true;
    swapcontext(&me.uc, &g_sched);   // the scheduler resumes us once the partner has posted
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
        }
        int live = nthr;
        while (live > 0) {
            // run every runnable thread until it finishes or blocks
            for (int t = 0; t < nthr; t++) {
                Thr& T = thr[(size_t)t];
                if (T.done || T.at_barrier || T.at_shfl || T.at_pair) continue;
                g_cur = t;
                set_ids(t, block);
                swapcontext(&g_sched, &T.uc);
                if (T.done) live--;
            }
            bool progressed = false;
            // resolve pair exchanges: lanes 2k and 2k + 1 swap once both have posted
            for (int t = 0; t + 1 < nthr; t += 2) {
                Thr &A = thr[(size_t)t], &B = thr[(size_t)t + 1];
                if (A.at_pair && B.at_pair) {
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
