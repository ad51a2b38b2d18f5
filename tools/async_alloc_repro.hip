// Stand-alone probe of the pattern under which this library's intermediate round-4 build returned wrong proofs:
// one thread keeps "loading keys" -- plain hipMalloc'd buffers, a scratch slab from hipMallocAsync on a shared low-priority queue,
// many short kernels that stage data through the slab into the key's rows, hipFreeAsync behind them, a second stream-ordered slab on
// another queue (the matrices' temporaries) -- while two other threads run verified kernels on their own queues and buffers and
// now and then allocate and free plain memory.  Every buffer is checked against what its own kernels must have left in it.
//   hipcc --offload-arch=gfx950 -O2 -o tools/alt/async_alloc_repro tools/async_alloc_repro.hip -lpthread
//   tools/alt/async_alloc_repro [seconds] [async=1|0] [reuse=1|0: 0 switches the default pool's cross-queue reuse policies off] [keep=0|1: 1 sets the pool's release threshold to its maximum]
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void fill(uint32_t* p, size_t n, uint32_t seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = seed * 2654435761u + (uint32_t)i;
}
// tmp[slot][i] = f(src[i]) with a little arithmetic in between (a "row step")
__global__ void step(const uint32_t* src, uint32_t* tmp, size_t n, uint32_t k) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t v = src[i];
    for (int r = 0; r < 200; r++) v = v * 1664525u + 1013904223u + k;
    tmp[i] = v;
}
__global__ void copy_out(const uint32_t* tmp, uint32_t* row, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) row[i] = tmp[i] ^ 0x5a5a5a5au;
}
static uint32_t host_step(uint32_t v, uint32_t k) { for (int r = 0; r < 200; r++) v = v * 1664525u + 1013904223u + k; return v; }

static std::atomic<long> bad_rows(0), bad_scratch(0), keys_done(0), checks_done(0);
static std::atomic<int> stop(0);

static void loader(bool use_async, hipStream_t qb, hipStream_t ql) {
    CK(hipSetDevice(0));
    uint32_t seed = 1;
    std::vector<uint32_t> h;
    struct Key { uint32_t* row0; uint32_t* rows; size_t n; uint32_t seed; hipEvent_t done; };
    std::vector<Key> live;
    while (!stop.load()) {
        const size_t n = 64 + (seed * 37u) % 4096;                 // tiny keys, like the Node suite's
        const int nrows = 12;
        Key k; k.n = n; k.seed = seed;
        CK(hipMalloc(&k.row0, n * 4)); CK(hipMalloc(&k.rows, n * 4 * nrows));
        CK(hipEventCreateWithFlags(&k.done, hipEventDisableTiming));
        // "matrices": a stream-ordered slab on the load queue, used and freed at once
        uint32_t* slab = nullptr;
        if (use_async) CK(hipMallocAsync((void**)&slab, n * 4 * 3, ql)); else CK(hipMalloc(&slab, n * 4 * 3));
        fill<<<(n + 255) / 256, 256, 0, ql>>>(k.row0, n, seed);
        fill<<<(n * 3 + 255) / 256, 256, 0, ql>>>(slab, n * 3, seed + 7);
        CK(hipStreamSynchronize(ql));
        if (use_async) CK(hipFreeAsync(slab, ql)); else CK(hipFree(slab));
        // the "table build": scratch from the stream-ordered allocator on the shared build queue
        uint32_t* tmp = nullptr;
        if (use_async) CK(hipMallocAsync((void**)&tmp, n * 4, qb)); else CK(hipMalloc(&tmp, n * 4));
        for (int r = 0; r < nrows; r++) {
            step<<<(n + 255) / 256, 256, 0, qb>>>(r == 0 ? k.row0 : k.rows + (size_t)(r - 1) * n, tmp, n, (uint32_t)r);
            copy_out<<<(n + 255) / 256, 256, 0, qb>>>(tmp, k.rows + (size_t)r * n, n);
        }
        if (use_async) CK(hipFreeAsync(tmp, qb));
        CK(hipEventRecord(k.done, qb));
        if (!use_async) { CK(hipEventSynchronize(k.done)); CK(hipFree(tmp)); }
        live.push_back(k);
        seed++;
        // retire the oldest keys: wait for their build, verify, free (like a GC'd key handle)
        while (live.size() > 6 || (stop.load() && !live.empty())) {
            Key o = live.front(); live.erase(live.begin());
            CK(hipEventSynchronize(o.done));
            h.resize(o.n * nrows);
            CK(hipMemcpy(h.data(), o.rows, o.n * 4 * nrows, hipMemcpyDeviceToHost));
            long bad = 0;
            for (size_t i = 0; i < o.n; i += 7) {
                uint32_t v = o.seed * 2654435761u + (uint32_t)i;
                for (int r = 0; r < nrows; r++) { v = host_step(v, (uint32_t)r) ^ 0x5a5a5a5au; if (h[(size_t)r * o.n + i] != v) { bad++; break; } }
            }
            if (bad) bad_rows++;
            CK(hipFree(o.row0)); CK(hipFree(o.rows)); CK(hipEventDestroy(o.done));
            keys_done++;
        }
    }
}

static void prover(int id) {
    CK(hipSetDevice(0));
    hipStream_t q; CK(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
    size_t n = 1 << 16;
    uint32_t* a; uint32_t* b; CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4));
    std::vector<uint32_t> h(n);
    uint32_t it = 0;
    while (!stop.load()) {
        const uint32_t seed = 1000003u * (uint32_t)id + it;
        fill<<<(n + 255) / 256, 256, 0, q>>>(a, n, seed);
        step<<<(n + 255) / 256, 256, 0, q>>>(a, b, n, 3);
        CK(hipMemcpyAsync(h.data(), b, n * 4, hipMemcpyDeviceToHost, q));
        CK(hipStreamSynchronize(q));
        for (size_t i = 0; i < n; i += 97) if (h[i] != host_step(seed * 2654435761u + (uint32_t)i, 3)) { bad_scratch++; break; }
        checks_done++;
        if (++it % 16 == 0) {        // a lane's scratch grows: plain free + malloc
            CK(hipFree(a)); CK(hipFree(b));
            n = (size_t)1 << (14 + it / 16 % 5);
            CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4));
            h.resize(n);
        }
    }
    CK(hipFree(a)); CK(hipFree(b)); CK(hipStreamDestroy(q));
}

int main(int argc, char** argv) {
    const int seconds = argc > 1 ? atoi(argv[1]) : 10;
    const bool use_async = argc > 2 ? atoi(argv[2]) != 0 : true;
    CK(hipSetDevice(0));
    int lo = 0, hi = 0;
    hipStream_t qb, ql;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithPriority(&qb, hipStreamNonBlocking, lo));
    CK(hipStreamCreateWithFlags(&ql, hipStreamNonBlocking));
    const bool reuse = argc > 3 ? atoi(argv[3]) != 0 : true;
    if (!reuse) {
        hipMemPool_t pool;
        CK(hipDeviceGetDefaultMemPool(&pool, 0));
        int off = 0;
        CK(hipMemPoolSetAttribute(pool, hipMemPoolReuseFollowEventDependencies, &off));
        CK(hipMemPoolSetAttribute(pool, hipMemPoolReuseAllowOpportunistic, &off));
        CK(hipMemPoolSetAttribute(pool, hipMemPoolReuseAllowInternalDependencies, &off));
    }
    const bool keep = argc > 4 ? atoi(argv[4]) != 0 : false;
    if (keep) {
        hipMemPool_t pool;
        CK(hipDeviceGetDefaultMemPool(&pool, 0));
        uint64_t thr = UINT64_MAX;
        CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr));
    }
    std::thread t0(loader, use_async, qb, ql), t1(prover, 1), t2(prover, 2);
    std::this_thread::sleep_for(std::chrono::seconds(seconds));
    stop.store(1);
    t0.join(); t1.join(); t2.join();
    printf("{\"stream_ordered_allocator\": %s, \"cross_queue_reuse_policies\": %s, \"release_threshold\": %s, \"seconds\": %d, \"keys_built_and_verified\": %ld, \"keys_with_wrong_rows\": %ld, \"prover_checks\": %ld, \"prover_checks_wrong\": %ld}\n",
           use_async ? "true" : "false", reuse ? "\"default\"" : "\"off\"", keep ? "\"max\"" : "\"default (0)\"", seconds, keys_done.load(), bad_rows.load(), checks_done.load(), bad_scratch.load());
    return 0;
}
