// prove.hip -- proving-key ingestion and Groth16 proof assembly.
//
// Replaces SURVEY.md section 8a rows a21-a23 (/root/reference src/bn128.js:580-720
// Bn128.groth16GenProof; key layout tools/buildpkey.js:124-240; loadPoint1/2
// src/bn128.js:441-453).  The five MSMs and CALC_H run on the GPU with the key resident in
// HBM; the O(1) tail (5 scalar multiplications, 9 additions, 3 inversions, src/bn128.js:671-712)
// runs on the host with the same field/curve headers.
#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <future>

#include <functional>
#include <thread>

#include "internal.h"

namespace wsnark {

struct ProvingKey {
    Context* owner = nullptr;      // the context (device) the key is resident on: every call on the handle runs there
    uint32_t n_vars = 0, n_public = 0, domain = 0;
    Affine<Fq> alfa1, beta1, delta1;
    Affine<Fq2> beta2, delta2;
    CsrMatrix polsA, polsB;
    DevBuf pointsA, pointsB1, pointsB2, pointsC, pointsH;
    DevBuf maskA, maskB;        // 1 byte per signal: 0 where A (resp. B1 and B2) is infinity: the variable is not in that matrix
    uint32_t infA = 0, infB = 0;   // how many of those there are
    bool sparseA = false, sparseB = false;   // enough of them to give those sums a plan variant that leaves them out
    // Fixed-base window tables: the key's points never change, so each section is kept as rows x n points, row w =
    // 2^(c w) * (the section) -- 13 rows at c = 20 for 2^20 pairs, 5.2 GB for the five sections.  Every window of a scalar
    // then adds into ONE bucket set: fewer, wider windows (13 instead of 16 passes over the points at 2^20), one
    // reduction tail per sum instead of one per window, no doubling chain on the host.  0 = plain sections.
    uint32_t table_cw = 0, table_ch = 0;     // window width of the A/B1/B2/C tables (n_local pairs) and of the H table (h_local pairs)
    // Points shard (multi-GPU, SURVEY.md section 8e split (i) = the reference's own worker split, src/bn128.js:353-361):
    // this handle holds the points of signals [lo, lo + n_local) of every section and h_local of the hExps -- 1/world
    // of the memory and of the additions, all table rows, uniform work whatever the window count.  world 1 = whole key.
    // The hExps slice is the contiguous range [hlo, hlo + h_local) (h computed in full on this rank), or -- h_log_m != 0
    // -- the rank's rows of the m-interleaved layout the distributed CALC_H leaves its slice of h in (element (row r, j)
    // = hExps[(rank*m/world + r) + m*j]).  The two sparse matrices are always complete (CALC_H needs every row).
    uint32_t shard_rank = 0, shard_world = 1;
    uint32_t lo = 0, n_local = 0, hlo = 0, h_local = 0, h_log_m = 0;
    // wall-clock of the load, by phase (ms): pols -> CSR, point sections host -> device, masks + conversion to the device
    // field's domain, fixed-base table build, whole call (what a cold caller pays before its first proof)
    double load_ms[5] = {0, 0, 0, 0, 0};
    // Round 4: the table rows 1.. are built IN THE BACKGROUND (the context's queue `build_q`): the load returns once the sections are resident
    // and converted (row 0 of every table = the plain section), proofs that arrive before `ev_tables` has fired run on the
    // plain sections (per-window plans: the round-1 / 2 path; 10 % slower by itself, about twice the time beside the build), later ones on the tables.  `tables_ready` only ever
    // goes 0 -> 1.  WSNARK_TABLE_ASYNC=0: the load waits for the build as in round 3.
    hipEvent_t ev_build0 = nullptr, ev_tables = nullptr;      // (timing events: their distance is the build's duration)
    // 1 = rows built (or no tables at all), 0 = build queued, `ev_tables` recorded behind it, -1 = the build could not be queued
    // (plain sections for good)
    std::atomic<int> tables_ready{1};
    // Otherwise read-only after load: any number of proofs may use one handle at once (each on its own lane, which holds the
    // per-proof buffers and events).
    ~ProvingKey() {
        // the build writes this key's buffers: it must be over before they go (the queue is the context's; the event is this key's build)
        if (ev_tables && tables_ready.load() == 0 && hipEventSynchronize(ev_tables) != hipSuccess) (void)hipGetLastError();
        if (ev_build0) (void)hipEventDestroy(ev_build0);
        if (ev_tables) (void)hipEventDestroy(ev_tables);
    }
};

// which window width a call that starts NOW may plan with: the tables' once they are built (wait = block until they are: calls
// whose partial results must mean the same on every rank), 0 = the plain sections
// A call that WAITS does so because its partial sums must mean the same on every rank (window shards): if the rows cannot be had,
// it fails instead of falling back to the plain sections' window width behind its peers' backs.
static int pkey_table_state(ProvingKey* K, bool wait, uint32_t* cw, uint32_t* ch) {
    *cw = K->table_cw; *ch = K->table_ch;
    if (!K->table_cw && !K->table_ch) return WS_OK;
    int st = K->tables_ready.load(std::memory_order_acquire);
#ifdef WSNARK_EMUL
    // (the emulator's "queue" has run the build by the time the load returns; tests hold a key in the not-yet-ready state with this
    //  switch to drive the plain-sections path of a table-layout key and the calls that wait)
    if (!wait && tuning_get("EMUL_TABLES_PENDING", 0)) { *cw = *ch = 0; return WS_OK; }
#endif
    if (st == 1) return WS_OK;
    if (st == 0) {
        if (wait ? hipEventSynchronize(K->ev_tables) == hipSuccess : hipEventQuery(K->ev_tables) == hipSuccess) {
            K->tables_ready.store(1, std::memory_order_release);
            return WS_OK;
        }
        (void)hipGetLastError();      // (hipErrorNotReady is not an error here)
    }
    *cw = *ch = 0;
    if (wait) { set_last_error("proving key: the fixed-base table rows could not be built (the key serves whole proofs from its plain sections)"); return WS_ERR_HIP; }
    return WS_OK;
}
Context* pkey_context(const ProvingKey* K) { return K ? K->owner : nullptr; }
int pkey_wait_tables(ProvingKey* K) {
    uint32_t cw, ch;
    return pkey_table_state(K, true, &cw, &ch);
}

const std::string& get_last_error();

static bool range_ok(uint64_t off, uint64_t bytes, size_t len) { return off <= len && bytes <= len - off; }

int pkey_load_sections(const KeySections& S, ProvingKey** out, KeyShard shard) {
    Context* C = ctx();
    if (!C) return WS_ERR_NOINIT;
    const uint32_t nv = S.n_vars, np = S.n_public, dom = S.domain;
    if (shard.world == 0 || shard.rank >= shard.world) return WS_ERR_ARG;
    if (nv == 0 || (uint64_t)np + 1 > nv) { set_last_error("proving key: nPublic + 1 > nVars"); return WS_ERR_FORMAT; }
    if (dom < 2 || (dom & (dom - 1)) || dom > (1u << 27)) { set_last_error("proving key: domainSize must be a power of two in [2, 2^27]"); return WS_ERR_SIZE; }
    const uint64_t nC = (uint64_t)nv - np - 1;
    if (S.lenPA < (uint64_t)nv * 64 || S.lenPB1 < (uint64_t)nv * 64 || S.lenPB2 < (uint64_t)nv * 128 || S.lenPC < nC * 64 ||
        S.lenPH < (uint64_t)dom * 64) {
        set_last_error("proving key: a point section is shorter than its header-implied size");
        return WS_ERR_FORMAT;
    }
    std::unique_ptr<ProvingKey> K(new ProvingKey());
    K->owner = C;
    K->n_vars = nv; K->n_public = np; K->domain = dom;
    // the rank's share: floor(n / world) pairs per rank, the remainder to the last one (src/bn128.js:354-361)
    K->shard_rank = shard.rank; K->shard_world = shard.world;
    {
        const uint32_t per = nv / shard.world, hper = dom / shard.world;
        K->lo = shard.rank * per;
        K->n_local = shard.rank == shard.world - 1 ? nv - K->lo : per;
        K->hlo = shard.rank * hper;
        K->h_local = shard.rank == shard.world - 1 ? dom - K->hlo : hper;
        if (shard.h_log_m) {
            const uint64_t m = shard.h_log_m <= 27 ? (uint64_t)1 << shard.h_log_m : 0;      // (range first: the value comes straight from the C ABI)
            if (m == 0 || m > dom || (shard.world & (shard.world - 1)) || m % shard.world) {
                set_last_error("proving key shard: the interleave 2^h_log_m must divide the domain and be a multiple of the (power-of-two) world size");
                return WS_ERR_ARG;
            }
            K->h_log_m = shard.h_log_m;
            K->hlo = 0;                      // (no contiguous range: see pointsH below)
            K->h_local = dom / shard.world;
        }
    }
    const uint32_t nl = K->n_local, hl = K->h_local, lo = K->lo;
    memcpy(&K->alfa1, S.alfa1, 64);
    memcpy(&K->beta1, S.beta1, 64);
    memcpy(&K->delta1, S.delta1, 64);
    memcpy(&K->beta2, S.beta2, 128);
    memcpy(&K->delta2, S.delta2, 128);
    hipStream_t s = C->stream;
    size_t used = 0;
    // Which sections become fixed-base tables -- what the key's resident memory buys (the sweep: profiles/r05_table_sweep.txt):
    // WSNARK_KEY_TABLE=1 (default) all five, 0 none (plain sections), 2 the hExps only, 3 A / B1 / B2 / C only.  Tables must fit:
    // at most WSNARK_TABLE_MAX_GB (default 160) and half of the memory that is free now.
    {
        const long which = tuning_get("KEY_TABLE", 1);
        if (which != 0) {
            const uint32_t cw = (which == 1 || which == 3) ? msm_table_window(nl ? nl : 1) : 0, ch = (which == 1 || which == 2) ? msm_table_window(hl ? hl : 1) : 0;
            const uint64_t bytes = (uint64_t)nl * 320 * msm_table_rows(cw) + (uint64_t)hl * 64 * msm_table_rows(ch);
            const uint64_t cap = (uint64_t)(tuning_get_real("TABLE_MAX_GB", 160.0) * (double)((uint64_t)1 << 30));   // GB, fractions allowed (0.5 = 512 MiB)
            size_t free_b = 0, total_b = 0;
            WS_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
            if (bytes <= cap && bytes <= free_b / 2 && (uint64_t)msm_table_rows(cw) * nl < ((uint64_t)1 << 31) &&
                (uint64_t)msm_table_rows(ch) * hl < ((uint64_t)1 << 31)) {
                K->table_cw = cw;
                K->table_ch = ch;
            }
        }
    }
    typedef std::chrono::steady_clock Clock;
    const auto t_begin = Clock::now();
    bool lap_failed = false;                       // an asynchronous fault of the uploads / conversion kernels surfaces in these drains
    auto lap = [&](Clock::time_point& from) {      // ms since `from`, after the queue has drained; `from` moves on
        if (hipStreamSynchronize(s) != hipSuccess) { lap_failed = true; (void)hipGetLastError(); }
        const auto now = Clock::now();
        const double ms = std::chrono::duration<double, std::milli>(now - from).count();
        from = now;
        return ms;
    };
    auto t_phase = t_begin;
    int rc;
    // the rank's slice of every section.  C holds points for signals nPublic+1.. only (src/bn128.js:620 slices the scalars
    // instead).  Resident copy: padded in front with infinities (x == 0) for the signals 0..nPublic, so that the C sum uses
    // the SAME scalar vector -- and the same digit/sort plan -- as A, B1 and B2: local index i = signal lo + i everywhere.
    const uint64_t c_front = lo < np + 1 ? std::min<uint64_t>(nl, (uint64_t)np + 1 - lo) : 0;      // leading infinities of the local C array
    const uint64_t c_first = (lo > np + 1 ? lo - (np + 1) : 0);                                     // first C point of the file that is mine
    struct Sec { DevBuf* d; const uint8_t* src; uint64_t bytes, row_bytes, rows; } secs[5] = {
        {&K->pointsA, S.A + (uint64_t)lo * 64, (uint64_t)nl * 64, (uint64_t)nl * 64, 0},
        {&K->pointsB1, S.B1 + (uint64_t)lo * 64, (uint64_t)nl * 64, (uint64_t)nl * 64, 0},
        {&K->pointsB2, S.B2 + (uint64_t)lo * 128, (uint64_t)nl * 128, (uint64_t)nl * 128, 0},
        {&K->pointsC, S.Cpts + c_first * 64, ((uint64_t)nl - c_front) * 64, (uint64_t)nl * 64, 0},
        {&K->pointsH, S.H + (uint64_t)K->hlo * 64, (uint64_t)hl * 64, (uint64_t)hl * 64, 0}};
    // the five buffers: rows x section.  Should the device refuse a table after all (fragmentation, another process), the
    // key falls back to plain sections instead of failing
    for (int attempt = 0; attempt < 2; attempt++) {
        const uint64_t rowsW = msm_table_rows(K->table_cw), rowsH = msm_table_rows(K->table_ch);
        hipError_t err = hipSuccess;
        for (auto& sc : secs) {
            sc.rows = sc.d == &K->pointsH ? rowsH : rowsW;
            if ((err = sc.d->alloc((size_t)std::max<uint64_t>(sc.row_bytes * sc.rows, 64))) != hipSuccess) break;
        }
        if (err == hipSuccess) break;
        (void)hipGetLastError();
        if (!K->table_cw && !K->table_ch) { set_last_error("proving key: device allocation of the point sections failed"); return WS_ERR_HIP; }
        for (auto& sc : secs) sc.d->release();
        K->table_cw = K->table_ch = 0;
    }
    const bool trace_load = tuning_get("TRACE", 0) == 1;
    auto since = [&](Clock::time_point from) { return std::chrono::duration<double, std::milli>(Clock::now() - from).count(); };
    if (trace_load) fprintf(stderr, "[wsnark trace] key load: device buffers allocated at %.2f ms\n", since(t_begin));
    std::vector<uint8_t> h_gather;
    if (K->h_log_m) {
        // the rank's rows of the m-interleaved layout, row-major (m / world) x (domain / m): local (r, j) = hExps[(rank m/world + r) + m j]
        const uint64_t m = (uint64_t)1 << K->h_log_m, per = m / shard.world, cols = dom / m;
        h_gather.resize((size_t)hl * 64);
        for (uint64_t r = 0; r < per; r++)
            for (uint64_t j = 0; j < cols; j++)
                memcpy(&h_gather[(size_t)(r * cols + j) * 64], S.H + ((uint64_t)shard.rank * per + r + m * j) * 64, 64);
        secs[4].src = h_gather.data();
        if (S.release) S.release(S.H, (size_t)dom * 64);      // (the gather touched a 1 / world share of the section's pages)
    }
    for (auto& sc : secs) {
        size_t front = 0;
        if (sc.d == &K->pointsC && c_front) {
            front = (size_t)c_front * 64;
            WS_HIP_CHECK(hipMemsetAsync(sc.d->p, 0, front, s));
        }
        if (sc.bytes && (rc = upload_staged((uint8_t*)sc.d->p + front, sc.src, (size_t)sc.bytes, s))) return rc;
        if (S.release && sc.bytes && sc.src != h_gather.data()) S.release(sc.src, (size_t)sc.bytes);
        if (trace_load) fprintf(stderr, "[wsnark trace] key load: section of %llu bytes handed to the copy queue at %.2f ms\n", (unsigned long long)sc.bytes, since(t_begin));
    }
    K->load_ms[1] = lap(t_phase);
    if (trace_load) fprintf(stderr, "[wsnark trace] key load: sections resident at %.2f ms\n", since(t_begin));
    // Variables that do not occur in matrix A (resp. B) have A (resp. B1 = B2) = infinity -- common: real circuits put
    // far fewer terms on the B side.  Their pairs cost a lane slot each in those sums, so when there are enough of
    // them the sums run on a plan VARIANT that leaves them out (msm_plan_variant: the per-bin sort and the task list
    // are redone, ~0.17 ms at 2^20; the digit extraction and the scatter are shared with the full plan).
    // WSNARK_PROVE_SPARSE: 0 never, 2 always.
    WS_HIP_CHECK(K->maskA.alloc((size_t)std::max<uint32_t>(nl, 1)));
    WS_HIP_CHECK(K->maskB.alloc((size_t)std::max<uint32_t>(nl, 1)));
    if (trace_load) fprintf(stderr, "[wsnark trace] key load: mask buffers allocated at %.2f ms\n", since(t_begin));
    if ((rc = msm_points_mask(K->pointsA.as<Affine<Fq>>(), nullptr, nl, K->maskA.as<uint8_t>(), &K->infA, s))) return rc;
    if ((rc = msm_points_mask(K->pointsB1.as<Affine<Fq>>(), K->pointsB2.as<Affine<Fq2>>(), nl, K->maskB.as<uint8_t>(), &K->infB, s))) return rc;
    {
        const int mode = (int)tuning_get("PROVE_SPARSE", 1);
        const bool big = nl >= (1u << 14);
        K->sparseA = mode == 2 || (mode == 1 && big && (uint64_t)K->infA * 100 >= (uint64_t)nl * 15);   // saves a share of one G1 sum
        K->sparseB = mode == 2 || (mode == 1 && big && (uint64_t)K->infB * 100 >= (uint64_t)nl * 5);    // ... of a G1 and a G2 sum
    }
    if (trace_load) fprintf(stderr, "[wsnark trace] key load: infinity masks taken at %.2f ms\n", since(t_begin));
    // resident keys are kept in the device field's internal domain: no per-proof conversion pass
    if ((rc = msm_prepare_points(0, K->pointsA.p, nl, s))) return rc;
    if ((rc = msm_prepare_points(0, K->pointsB1.p, nl, s))) return rc;
    if ((rc = msm_prepare_points(1, K->pointsB2.p, nl, s))) return rc;
    if ((rc = msm_prepare_points(0, K->pointsC.p, nl, s))) return rc;
    if ((rc = msm_prepare_points(0, K->pointsH.p, hl, s))) return rc;
    K->load_ms[2] = lap(t_phase);
    if (trace_load) fprintf(stderr, "[wsnark trace] key load: sections converted at %.2f ms\n", since(t_begin));
    // The two record streams are transposed side by side (calch.hip: pols_to_csr -- a header walk on the host, upload, three kernels
    // each, on queues of their own; round 3 did the two passes over the records on host threads: 56-65 ms of a 2^20 key's 214).
    // BEFORE the table build is launched: under it the transposition's small kernels wait for issue slots behind kernels that
    // use 0.95 of them (measured: 133 ms instead of 20), and the load must not return before the matrices are in.
    {
        hipStream_t sa = C->load_q[0], sb = C->load_q[1];       // (the context's: creating and destroying two queues per load cost milliseconds under ROCm 7.2)
        size_t used_b = 0;
        const int device = C->device;
        std::string err_b;                      // (the error text is per thread: carried back by hand)
        std::future<int> fb = std::async(std::launch::async, [&, device, sb]() {
            // THIS context on the helper thread, not just its device (round 6): a new thread's selection is the process's DEFAULT
            // context, so the records of matrix B went up through the default context's staging ring whatever context was loading --
            // a group member's load left the default ring's slot events recorded on ITS queue, and once the group was gone the next
            // upload through the default ring queried events whose queue had been destroyed (hipEventQuery dereferences the event's
            // last queue: "operation not permitted on an event last recorded in a capturing stream", one GPU suite in five); on a
            // real second GPU the record itself fails (an event of device 0 on a queue of device 1).
            CtxScope helper_scope(C);
            if (hipSetDevice(device) != hipSuccess) return (int)WS_ERR_HIP;
            const int r = pols_to_csr(S.polsB, (size_t)S.lenB, nv, dom, &K->polsB, &used_b, sb, S.release);
            if (r) err_b = get_last_error();
            return r;
        });
        rc = pols_to_csr(S.polsA, (size_t)S.lenA, nv, dom, &K->polsA, &used, sa, S.release);
        const int rcb = fb.get();
        if (rc) return rc;                      // (~ProvingKey waits for the build queue)
        if (rcb) { set_last_error(err_b); return rcb; }
    }
    K->load_ms[0] = lap(t_phase);               // (also drains `s`: sections resident and converted -- proofs may start)
    if (lap_failed) { set_last_error("proving key: a HIP error surfaced while the sections were uploaded and converted"); return WS_ERR_HIP; }
    if (trace_load) fprintf(stderr, "[wsnark trace] key load: matrices transposed at %.2f ms\n", since(t_begin));
    if (K->table_cw || K->table_ch) {
        // Rows 1.. of the tables, from row 0, in that domain: queued on the CONTEXT's build queue (lowest stream priority; `s` is
        // drained by the lap above), one key's build after the other, in short launches (a row per launch) through the context's
        // scratch slab (the one long kernel per section only if the slab cannot be had).  No stream-ordered allocation anywhere near this: an earlier
        // version of this round took the slab from hipMallocAsync / hipFreeAsync (and the matrices' temporaries likewise), and the Node
        // suite -- many small keys loaded back to back, their builds still running under later proofs -- then produced a WRONG proof in
        // 3 to 36 of 60 runs, depending on how long the builds overlapped later loads and proofs; with plain allocations 0 of 120
        // (tools/async_alloc_repro.hip shows the same corruption without any of this library's code, on ROCm 7.2.0).
        WS_HIP_CHECK(hipEventCreate(&K->ev_build0));
        WS_HIP_CHECK(hipEventCreate(&K->ev_tables));
        std::lock_guard<std::mutex> build_lk(C->build_mu);
        hipStream_t b = C->build_q;
        void* tmp = nullptr;
        size_t tmp_bytes = 0;
        {
            // lanes per slab = how much of the chip the build holds at a time.  Measured at 2^20 (tools/build_slab_ab.sh): 2^18 lanes
            // (1024-2048 workgroups in flight) -- proofs beside the build 19-23 ms, build 166 ms; 2^16 -- 16-17 ms, 189 ms; 2^15 --
            // 14-15 ms, 264 ms (and 8x the launches).  Default 2^16, growing with the key so that the build stays ~2 700 launches
            // (2^24: 2^18 lanes): they all have to fit the queue before the load can return.
            const uint64_t most = nl > hl ? nl : hl;
            uint64_t cap = (uint64_t)tuning_get("TABLE_SLAB_LANES", 0);
            if (!cap) { cap = most >> 6; cap = cap < (1u << 16) ? (1u << 16) : cap > (1u << 18) ? (1u << 18) : cap; }
            tmp_bytes = msm_table_scratch_bytes(((most < cap ? most : cap) + 63) & ~(uint64_t)63);
            if (C->build_tmp.bytes < tmp_bytes) {
                // growing frees the old slab: no build may still be using it (rare: a larger key than any before, while builds run)
                WS_HIP_CHECK(hipStreamSynchronize(b));
                if (C->build_tmp.reserve(tmp_bytes) != hipSuccess) { (void)hipGetLastError(); tmp_bytes = 0; }
            }
            tmp = tmp_bytes ? C->build_tmp.p : nullptr;
        }
        K->tables_ready.store(0);
        WS_HIP_CHECK(hipEventRecord(K->ev_build0, b));
        rc = msm_build_table(0, K->pointsA.p, nl, K->table_cw, b, tmp, tmp_bytes);
        if (!rc) rc = msm_build_table(0, K->pointsB1.p, nl, K->table_cw, b, tmp, tmp_bytes);
        if (!rc) rc = msm_build_table(1, K->pointsB2.p, nl, K->table_cw, b, tmp, tmp_bytes);
        if (!rc) rc = msm_build_table(0, K->pointsC.p, nl, K->table_cw, b, tmp, tmp_bytes);
        if (!rc) rc = msm_build_table(0, K->pointsH.p, hl, K->table_ch, b, tmp, tmp_bytes);
        if (rc || hipEventRecord(K->ev_tables, b) != hipSuccess) {
            // whatever was queued still writes this key's buffers: wait for it before the handle (and its buffers) go away
            (void)hipStreamSynchronize(b);
            (void)hipGetLastError();
            K->tables_ready.store(-1);
            return rc ? rc : (int)WS_ERR_HIP;
        }
    }
    if ((K->table_cw || K->table_ch) && tuning_get("TABLE_ASYNC", 1) == 0) {
        if (K->tables_ready.load() == 0) {
            WS_HIP_CHECK(hipEventSynchronize(K->ev_tables));
            K->tables_ready.store(1);
        }
        K->load_ms[3] = lap(t_phase);           // the table build, waited for
    }
    K->load_ms[4] = std::chrono::duration<double, std::milli>(Clock::now() - t_begin).count();
    *out = K.release();
    return WS_OK;
}

// the sections of a proving_key.bin image (the reference's file format, tools/buildpkey.js:124-240), bounds checked
int pkey_parse(const uint8_t* buf, size_t len, KeySections* out) {
    if (!buf || !out) return WS_ERR_ARG;
    // 10 x u32 header (src/bn128.js:581-591), then alfa1, beta1, delta1 (3 x 64 B), beta2, delta2 (2 x 128 B)
    if (len < 40 + 448) { set_last_error("proving key shorter than its fixed header"); return WS_ERR_FORMAT; }
    uint32_t h[10];
    memcpy(h, buf, 40);
    const uint32_t nv = h[0], np = h[1], dom = h[2];
    const uint64_t pPolsA = h[3], pPolsB = h[4], pA = h[5], pB1 = h[6], pB2 = h[7], pC = h[8], pH = h[9];
    if (nv == 0 || (uint64_t)np + 1 > nv) { set_last_error("proving key: nPublic + 1 > nVars"); return WS_ERR_FORMAT; }
    const uint64_t nC = (uint64_t)nv - np - 1;
    if (!(pPolsA >= 488 && pPolsA <= pPolsB && pPolsB <= pA) || !range_ok(pA, (uint64_t)nv * 64, len) ||
        !range_ok(pB1, (uint64_t)nv * 64, len) || !range_ok(pB2, (uint64_t)nv * 128, len) ||
        !range_ok(pC, nC * 64, len) || !range_ok(pH, (uint64_t)dom * 64, len)) {
        set_last_error("proving key: section offsets out of range");
        return WS_ERR_FORMAT;
    }
    // true section bounds from the header (the reference slices over-long: src/bn128.js:592-593)
    *out = KeySections{nv, np, dom, buf + 40, buf + 104, buf + 168, buf + 232, buf + 360,
                       buf + pPolsA, pPolsB - pPolsA, buf + pPolsB, pA - pPolsB,
                       buf + pA, buf + pB1, buf + pB2, buf + pC, buf + pH,
                       len - pA, len - pB1, len - pB2, len - pC, len - pH};
    return WS_OK;
}
int pkey_load(const uint8_t* buf, size_t len, ProvingKey** out) {
    if (!ctx()) return WS_ERR_NOINIT;
    if (!out) return WS_ERR_ARG;
    KeySections S;
    int rc = pkey_parse(buf, len, &S);
    if (rc) return rc;
    return pkey_load_sections(S, out, KeyShard{});
}

void pkey_free(ProvingKey* K) { delete K; }
void pkey_info(const ProvingKey* K, uint32_t* nv, uint32_t* np, uint32_t* dom) {
    if (nv) *nv = K->n_vars;
    if (np) *np = K->n_public;
    if (dom) *dom = K->domain;
}

void pkey_table_info(const ProvingKey* K, uint32_t* cw, uint32_t* rw, uint32_t* ch, uint32_t* rh, uint64_t* bytes) {
    if (cw) *cw = K->table_cw;
    if (rw) *rw = msm_table_rows(K->table_cw);
    if (ch) *ch = K->table_ch;
    if (rh) *rh = msm_table_rows(K->table_ch);
    if (bytes) *bytes = (uint64_t)K->n_local * 320 * msm_table_rows(K->table_cw) + (uint64_t)K->h_local * 64 * msm_table_rows(K->table_ch);
}
void pkey_load_stats(const ProvingKey* K, double* out5) {
    memcpy(out5, K->load_ms, sizeof K->load_ms);
    // a background build reports its duration once it is over (0 until then)
    const int st = K->tables_ready.load(std::memory_order_acquire);      // (2: the event is not recorded yet; -1: never will be)
    if ((K->table_cw || K->table_ch) && out5[3] == 0 && K->ev_tables && (st == 0 || st == 1) && hipEventQuery(K->ev_tables) == hipSuccess) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, K->ev_build0, K->ev_tables) == hipSuccess) out5[3] = ms;
    } else {
        (void)hipGetLastError();
    }
}
void pkey_shard_info(const ProvingKey* K, uint32_t* rank, uint32_t* world, uint64_t* lo, uint64_t* n_local, uint64_t* h_local, uint32_t* h_log_m) {
    if (rank) *rank = K->shard_rank;
    if (world) *world = K->shard_world;
    if (lo) *lo = K->lo;
    if (n_local) *n_local = K->n_local;
    if (h_local) *h_local = K->h_local;
    if (h_log_m) *h_log_m = K->h_log_m;
}

// serial EC sum of Jacobian-Montgomery partials: the main-thread gather loop of the reference
// (src/bn128.js:374-382 g1m_add over the workers' results; :406-414 for G2).  Host arithmetic.
template <class C, class F>
static void sum_jac(const uint8_t* pts, uint64_t count, uint8_t* out) {
    typename C::Pt acc = C::infinity();
    for (uint64_t i = 0; i < count; i++) {
        Jac<F> j;
        memcpy(&j, pts + i * sizeof(Jac<F>), sizeof j);
        acc = C::add(acc, C::from_jac(j));
    }
    Jac<F> r = C::to_affine_jac(acc);
    memcpy(out, &r, sizeof r);
}
void g1_sum_host(const uint8_t* pts, uint64_t count, uint8_t* out96) { sum_jac<G1, Fq>(pts, count, out96); }
void g2_sum_host(const uint8_t* pts, uint64_t count, uint8_t* out192) { sum_jac<G2, Fq2>(pts, count, out192); }

// The blinding values' entropy: the reference draws crypto.randomBytes(32) twice (src/bn128.js:642-661).  getrandom(2) -- no file
// descriptor, works in a chroot without /dev, blocks only until the kernel's pool has been seeded once -- with an UNBUFFERED read
// of /dev/urandom for kernels without the system call (round 5 pulled 4 KiB through stdio for 64 bytes).
static int os_random(uint8_t* out, size_t n) {
    size_t got = 0;
#if defined(__linux__) && defined(SYS_getrandom)
    while (got < n) {
        const long r = syscall(SYS_getrandom, out + got, n - got, 0);
        if (r < 0) {
            if (errno == EINTR) continue;
            break;                                   // ENOSYS (kernel < 3.17) or a seccomp filter: the device file below
        }
        got += (size_t)r;
    }
    if (got == n) return 0;
#endif
    const int fd = open("/dev/urandom", O_RDONLY | O_CLOEXEC);
    if (fd < 0) return -1;
    while (got < n) {
        const ssize_t r = read(fd, out + got, n - got);
        if (r < 0 && errno == EINTR) continue;
        if (r <= 0) break;
        got += (size_t)r;
    }
    close(fd);
    return got == n ? 0 : -1;
}

static void store_plain(uint8_t* dst, const Fe& mont) {
    Fe p = Fq::from_mont(mont);
    memcpy(dst, &p, 32);
}

// WSNARK_TRACE=1: host-side wall-clock of the proof phases on stderr
struct Trace {
    bool on;
    std::chrono::steady_clock::time_point t0, last;
    Trace() : on(tuning_get("TRACE", 0) == 1) { t0 = last = std::chrono::steady_clock::now(); }
    void mark(const char* what) {
        if (!on) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[wsnark trace] %-28s +%8.3f ms  (t=%8.3f)\n", what,
                std::chrono::duration<double, std::milli>(now - last).count(),
                std::chrono::duration<double, std::milli>(now - t0).count());
        last = now;
    }
};

struct MsmSums {
    XYZZ<Fq> A, B1, C, H;
    XYZZ<Fq2> B2;
};

// CALC_H and the five MSMs (src/bn128.js:607-620) on lane L.  With a window shard other than {0, 1} the sums are
// this rank's partial sums.
//
// Two in-order queues.  Stream s: the witness plan (+ its variants), A, B1, C (one batched tail), B2.  The lane's
// second queue: CALC_H, the H plan and the H sum.  The reduction tails are chains of ~30 dependent
// point additions on a few hundred wavefronts; the other queue's full-width kernels take the SIMDs they leave idle.
// Measured on MI355X, prove 2^20: round 1 one queue (WSNARK_PROVE_OVERLAP=0) 14.5 ms -> two queues 13.2 ms; round 2 (dense
// key) one queue 12.8 ms -> two queues 11.2 ms, see the sweep notes at `overlap` below.  A tail that shares its SIMDs
// with a full-width kernel runs 2-3x slower (the kernel timeline of one proof: profiles/r02_session25_prove_timeline.txt).
// The host finishes each sum while the GPU works on the next ones; `after_ab1` (optional) runs on the host as soon
// as A and B1 are known.
// calc_h (optional): enqueues the computation of the h this handle's hExps share is summed against on the given queue
// (the distributed CALC_H of dist.hip) instead of the whole CALC_H.
typedef std::function<int(hipStream_t, Fe*)> CalcHFn;
// h_witness (optional): the witness is still in HOST memory.  It is uploaded into d_witness (which must then be the lane's
// own witness buffer) chunk by chunk on the lane's copy queue, and the first pass over it -- the digit histogram of the
// grouping pass -- runs on every chunk as it lands, instead of behind the whole transfer (src/bn128.js:580: the reference's
// callers hand over host memory; the 32 B x nVars of H2D are inside every drop-in call).
static int prove_msms(ProvingKey* K, Lane& L, const Fe* d_witness, WindowShard sh, MsmSums* out, hipStream_t s,
                      const std::function<void(const MsmSums&)>& after_ab1 = nullptr, bool skip_h = false, const CalcHFn& calc_h = nullptr,
                      const uint8_t* h_witness = nullptr) {
    Trace tr;
    // a points-sharded key sums its own pairs: the witness slice [lo, lo + n_local) against the resident slice of every
    // section (all windows), h[hlo ..] against its hExps slice
    const uint32_t nv = K->n_local, dom = K->domain;
    // tables or plain sections: decided once per proof (a window-sharded call waits for the tables: its partial sums must mean
    // the same on every rank)
    uint32_t table_cw, table_ch;
    int rc = pkey_table_state(K, sh.off != 0 || sh.stride != 1, &table_cw, &table_ch);
    if (rc) return rc;
    const Fe* d_witness_all = d_witness;
    d_witness += K->lo;
    if (K->shard_world > 1 && (sh.off != 0 || sh.stride != 1)) { set_last_error("prove: a points-sharded key cannot be window-sharded as well"); return WS_ERR_ARG; }
    if (K->h_log_m && !skip_h && !calc_h) { set_last_error("prove: this handle holds an interleaved hExps slice (distributed CALC_H only)"); return WS_ERR_ARG; }
    // WSNARK_PROVE_OVERLAP=0: everything on one queue (bench.py's pass that times every kernel alone); default: two queues.
    const bool overlap = tuning_get("PROVE_OVERLAP", 2) != 0;
    hipStream_t s2 = overlap ? L.stream2 : s;
    // launch slots: A, B1, C, B2, H.  On an error path the launches of THIS proof are forgotten (other lanes' are not touched)
    int slots[5] = {-1, -1, -1, -1, -1};
    int &hA = slots[0], &hB1 = slots[1], &hC = slots[2], &hB2 = slots[3], &hH = slots[4];
    struct Abort {
        Lane& L; int* slots; hipStream_t a, b; bool armed;
        ~Abort() { if (armed) { msm_select_plan(L, 0); if (L.stream3) (void)hipStreamSynchronize(L.stream3); msm_abort_slots(L, slots, 5, a, b); } }
    } guard{L, slots, s, s2, true};
    for (hipEvent_t* e : {&L.ev_start, &L.ev_h})
        if (!*e) WS_HIP_CHECK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    // the four sums whose scalars are the witness (:617-620)
    msm_select_plan(L, 0);
    if (h_witness) {
        hipStream_t sc = L.stream_copy;
        for (hipEvent_t* e : {&L.ev_chunk[0], &L.ev_chunk[1]})
            if (!*e) WS_HIP_CHECK(hipEventCreateWithFlags(e, hipEventDisableTiming));
        // the copy queue starts behind whatever the caller's queue still holds for this lane's witness buffer
        WS_HIP_CHECK(hipEventRecord(L.ev_start, s));
        WS_HIP_CHECK(hipStreamWaitEvent(sc, L.ev_start, 0));
        if ((rc = msm_plan_begin(L, nv, sh, s, table_cw))) return rc;
        unsigned k = 0;
        const uint64_t lo_sig = K->lo, hi_sig = (uint64_t)K->lo + nv;
        rc = upload_pipelined(const_cast<Fe*>(d_witness_all), h_witness, (size_t)K->n_vars * 32, sc, [&](size_t blo, size_t bhi, hipStream_t cq) -> int {
            hipEvent_t ev = L.ev_chunk[k++ & 1];             // (a wait captures the record that precedes it: two events suffice)
            WS_HIP_CHECK(hipEventRecord(ev, cq));
            WS_HIP_CHECK(hipStreamWaitEvent(s, ev, 0));
            uint64_t e0 = blo / 32, e1 = bhi / 32;            // signals of this chunk; the part of them this handle sums
            if (e0 < lo_sig) e0 = lo_sig;
            if (e1 > hi_sig) e1 = hi_sig;
            return e0 < e1 ? msm_plan_count(L, d_witness, e0 - lo_sig, e1 - lo_sig, s) : (int)WS_OK;
        });
        if (rc) return rc;
        tr.mark("witness staged, histogram enqueued per chunk");
        WS_HIP_CHECK(hipEventRecord(L.ev_start, s));      // the whole witness is resident (s has waited for every chunk)
        if ((rc = msm_plan_finish(L, d_witness, s))) return rc;
    } else {
        WS_HIP_CHECK(hipEventRecord(L.ev_start, s));      // the witness is ready on s
        if ((rc = msm_plan_dev(L, d_witness, nv, sh, s, table_cw))) return rc;
    }
    // CALC_H goes to the second queue HERE -- right behind the witness plan's few launches, before the host enqueues the dozens of
    // launches of the four sums (round 5).  The kernel trace of a 2^20 proof (profiles/r05_timeline_*) showed why: queued last, its
    // first kernel reached the GPU 0.3-0.4 ms into the proof, when the G2 accumulation had just filled every SIMD with workgroups
    // that live for ~1 ms, and did not START before 1.6 ms, while the grouping pass before that (0.4 ms, LDS-atomic bound) had the chip
    // to itself.  Queued here the sparse products and the first transforms run beside the grouping pass.  (Whole proofs measure the
    // same either way, profiles/r05_schedule_experiments.txt: time on this chip is the sum of the kernels' own times, section 5 of
    // DESIGN.md; the H sum's inputs are ready ~1 ms earlier.)
    // (the distributed CALC_H keeps its place behind the sums: its exchanges are host callbacks that may block this thread; queued
    //  first at a world of one it measured the same, +0.67 / +0.86 against +0.83 / +0.63 ms over the one-call prover in four runs)
    Fe* d_h = nullptr;
    bool calc_h_done = false;
    auto enqueue_calc_h = [&]() -> int {
        if (s2 != s) WS_HIP_CHECK(hipStreamWaitEvent(s2, L.ev_start, 0));
        WS_HIP_CHECK(L.h.reserve((size_t)dom * 32));
        d_h = L.h.as<Fe>();
        int r = calc_h ? calc_h(s2, d_h) : calc_h_dev(L, d_witness_all, K->n_vars, K->polsA, K->polsB, dom, d_h, s2);
        calc_h_done = true;
        return r;
    };
    if (!skip_h && !calc_h && s2 != s) {
        if ((rc = enqueue_calc_h())) return rc;
        tr.mark("calc_h enqueued");
    }
    // one grouping pass for all four; A and B1/B2 may run on variants of the plan that leave out the variables
    // absent from their matrix (plan 0 = full, 2 = without B's absentees, 3 = without A's)
    int planA = 0, planB = 0;
    if (K->sparseB && msm_plan_variant(L, 0, 2, K->maskB.as<uint8_t>(), s) == WS_OK) planB = 2;
    if (K->sparseA && msm_plan_variant(L, 0, 3, K->maskA.as<uint8_t>(), s) == WS_OK) planA = 3;
    // Two arrangements of the sums, by size.
    // FULL SIZE (>= 2^23 (row, pair) entries per G1 sum): everything on this queue -- B2 with its tail first, then the accumulations
    // of A and B1 under ONE batched tail, C last; CALC_H and the H sum on the second queue.  The G2 tail is the slowest chain of the
    // proof (a G2 addition on a lone wavefront: ~23 us, G1: ~7 us); early in the queue it runs beside CALC_H's kernels instead of
    // at the end of the proof, where nothing is left to fill the SIMDs.  A and B1 reach the host before C's accumulation ends, so
    // the host's share of pi_c (after_ab1) overlaps GPU work.  Every other arrangement tried -- tails on a third queue, all
    // accumulations back to back, the G2 sum in the middle or last, gates between the queues, stream and wavefront priorities --
    // measured the same or worse (rounds 1-3 and profiles/r05_schedule_experiments.txt).
    // SMALL (round 3; a rank's share of a points-sharded key, circuits up to 2^19): every sum of such a proof is a latency chain --
    // grouping, a sub-millisecond accumulation on a fraction of the SIMDs, a reduction tail of ~33 dependent additions -- so the
    // chains run BESIDE each other: B2 with its tail on a third queue, A, B1 and C under ONE batched tail on the first, CALC_H and
    // H on the second (profiles/r03_s12_prove_order3.txt: 2^16 proofs -31 %, 2^18 -35 %, 2^19 -8 %, 2^20 +3 %).
    const bool small = overlap && (uint64_t)nv * msm_table_rows(table_cw ? table_cw : 16) < ((uint64_t)1 << 23) && L.stream3 && s != L.stream3;
    const Affine<Fq>* g1sets[3] = {K->pointsA.as<Affine<Fq>>(), K->pointsB1.as<Affine<Fq>>(), K->pointsC.as<Affine<Fq>>()};
    const int plans[3] = {planA, planB, 0};
    int g1slots[3] = {-1, -1, -1};
    if (small) {
        hipStream_t s3 = L.stream3;
        for (hipEvent_t* e : {&L.ev_plan, &L.ev_g2})
            if (!*e) WS_HIP_CHECK(hipEventCreateWithFlags(e, hipEventDisableTiming));
        WS_HIP_CHECK(hipEventRecord(L.ev_plan, s));                  // the plan (and its variants) are complete on s
        WS_HIP_CHECK(hipStreamWaitEvent(s3, L.ev_plan, 0));
        msm_select_plan(L, planB);
        rc = msm_g2_launch(L, K->pointsB2.as<Affine<Fq2>>(), true, &hB2, s3);                              // :619, on its own queue
        msm_select_plan(L, 0);
        if (rc) return rc;
        WS_HIP_CHECK(hipEventRecord(L.ev_g2, s3));
        rc = msm_g1_launch_batch(L, g1sets, 3, true, g1slots, s, plans);                                   // :617, :618, :620: one tail
        hA = g1slots[0]; hB1 = g1slots[1]; hC = g1slots[2];
        if (rc) return rc;
        WS_HIP_CHECK(hipStreamWaitEvent(s, L.ev_g2, 0));              // s stays the caller's ordering point
        tr.mark("plan(w) + B2 on queue 3 + A, B1, C batched");
    } else {
        msm_select_plan(L, planB);
        rc = msm_g2_launch(L, K->pointsB2.as<Affine<Fq2>>(), true, &hB2, s);                               // :619
        msm_select_plan(L, 0);
        if (rc) return rc;
        tr.mark("plan(w) [+ variants] + launch B2");
        rc = msm_g1_launch_batch(L, g1sets, 2, true, g1slots, s, plans);                                   // :617, :618
        hA = g1slots[0]; hB1 = g1slots[1];
        if (rc) return rc;
        msm_select_plan(L, 0);
        if ((rc = msm_g1_launch(L, K->pointsC.as<Affine<Fq>>(), true, &hC, s))) return rc;                // :620 (padded)
        tr.mark("launch A, B1, C");
    }
    if (skip_h) {
        // distributed proving with the four-step CALC_H (wasmsnark_amd/dist.py): h and the H sum are the caller's
        if ((rc = msm_g1_finish(L, hA, &out->A))) return rc;
        if ((rc = msm_g1_finish(L, hB1, &out->B1))) return rc;
        if (after_ab1) after_ab1(*out);
        if ((rc = msm_g1_finish(L, hC, &out->C))) return rc;
        if ((rc = msm_g2_finish(L, hB2, &out->B2))) return rc;
        out->H = G1::infinity();
        guard.armed = false;
        return WS_OK;
    }
    // CALC_H (unless queued above), then the H MSM over domainSize pairs (src/bn128.js:607-615)
    if (!calc_h_done) {
        if ((rc = enqueue_calc_h())) return rc;
        tr.mark("calc_h enqueued");
    }
    msm_select_plan(L, s2 != s ? 1 : 0);
    rc = msm_plan_dev(L, d_h + K->hlo, K->h_local, sh, s2, table_ch);
    if (!rc) rc = msm_g1_launch(L, K->pointsH.as<Affine<Fq>>(), true, &hH, s2);                           // :614
    msm_select_plan(L, 0);
    if (rc) return rc;
    if (s2 != s) { WS_HIP_CHECK(hipEventRecord(L.ev_h, s2)); WS_HIP_CHECK(hipStreamWaitEvent(s, L.ev_h, 0)); }   // s stays the caller's ordering point
    tr.mark("plan(h) + launch H");
    const bool g2_fin = !small;                      // (small proofs: B2 ends on its own queue, A / B1 / C come first)
    if (g2_fin && (rc = msm_g2_finish(L, hB2, &out->B2))) return rc;
    if ((rc = msm_g1_finish(L, hA, &out->A))) return rc;
    if ((rc = msm_g1_finish(L, hB1, &out->B1))) return rc;
    tr.mark(g2_fin ? "finish B2, A, B1" : "finish A, B1");
    if (after_ab1) after_ab1(*out);
    if (!g2_fin && (rc = msm_g1_finish(L, hC, &out->C))) return rc;
    tr.mark(g2_fin ? "host work on A, B1" : "host work on A, B1; finish C");
    // the two queues end independently: finish whichever sum reaches the host first (its serial host tail then runs while
    // the GPU still works on the other one), so poll both instead of blocking on one
    {
        const int hX = g2_fin ? hC : hB2;                 // the first queue's last sum
        auto finish_x = [&]() -> int { return g2_fin ? msm_g1_finish(L, hC, &out->C) : msm_g2_finish(L, hB2, &out->B2); };
        bool doneX = false, doneH = false;
        for (unsigned spins = 0; !(doneX && doneH); spins++) {
            if (!doneX && (doneH || msm_ready(L, hX))) { if ((rc = finish_x())) return rc; doneX = true; continue; }
            if (!doneH && (doneX || msm_ready(L, hH))) { if ((rc = msm_g1_finish(L, hH, &out->H))) return rc; doneH = true; continue; }
            if (spins > 64) std::this_thread::yield();
            if (spins > (1u << 22)) {       // (never observed: an event that does not report ready -- block on the sums in turn)
                if (!doneH) { if ((rc = msm_g1_finish(L, hH, &out->H))) return rc; doneH = true; }
                if (!doneX) { if ((rc = finish_x())) return rc; doneX = true; }
            }
        }
    }
    tr.mark(g2_fin ? "finish H, C" : "finish H, B2");
    guard.armed = false;
    return WS_OK;
}

struct Blinding {
    Fe rr, ss, rs;                 // r mod r, s mod r, r*s mod r (plain)
    struct Pre { G1::Pt r_delta1, s_delta1, rs_delta1; G2::Pt s_delta2; };
    std::future<Pre> pre;
    const uint8_t* rb() const { return reinterpret_cast<const uint8_t*>(&rr); }
    const uint8_t* sb() const { return reinterpret_cast<const uint8_t*>(&ss); }
    ~Blinding() { if (pre.valid()) pre.wait(); }
};

// r, s are raw 256-bit values (not reduced, src/bn128.js:642-661); every point here has prime order r, so
// k*P == (k mod r)*P and (r*s)*P == ((r mod r)(s mod r) mod r)*P (:700-702).  The scalar multiplications
// that involve key points only start on a host thread at once (they overlap the GPU work).
static thread_local uint8_t t_last_rs[64];
static thread_local bool t_have_rs = false;
// the raw 32-byte r | s of the last proof assembled by the calling thread: the reference keeps them for its tests
// in the same way (`this._pr`, `this._ps`, src/bn128.js:662-664)
bool last_blinding(uint8_t* r32, uint8_t* s32) {
    if (!t_have_rs) return false;
    if (r32) memcpy(r32, t_last_rs, 32);
    if (s32) memcpy(s32, t_last_rs + 32, 32);
    return true;
}
static int start_blinding(ProvingKey* K, const uint8_t* r32, const uint8_t* s32, Blinding* B) {
    uint8_t rnd[64];
    if (!r32 || !s32) {
        if (os_random(rnd, 64)) { set_last_error("no entropy: getrandom(2) and /dev/urandom both failed"); return WS_ERR_ARG; }
        if (!r32) r32 = rnd;
        if (!s32) s32 = rnd + 32;
    }
    memcpy(t_last_rs, r32, 32);
    memcpy(t_last_rs + 32, s32, 32);
    t_have_rs = true;
    memcpy(&B->rr, r32, 32);
    memcpy(&B->ss, s32, 32);
    B->rr = Fr::reduce_full(B->rr);
    B->ss = Fr::reduce_full(B->ss);
    B->rs = Fr::from_mont(Fr::mul(Fr::to_mont(B->rr), Fr::to_mont(B->ss)));
    const Fe rr = B->rr, ss = B->ss, rs = B->rs;
    const G1::Pt delta1 = G1::from_affine(K->delta1);
    const G2::Pt delta2 = G2::from_affine(K->delta2);
    B->pre = std::async(std::launch::async, [rr, ss, rs, delta1, delta2]() {
        Blinding::Pre p;
        p.r_delta1 = G1::mul_bytes(delta1, reinterpret_cast<const uint8_t*>(&rr), 32);
        p.s_delta1 = G1::mul_bytes(delta1, reinterpret_cast<const uint8_t*>(&ss), 32);
        p.rs_delta1 = G1::mul_bytes(delta1, reinterpret_cast<const uint8_t*>(&rs), 32);
        p.s_delta2 = G2::mul_bytes(delta2, reinterpret_cast<const uint8_t*>(&ss), 32);
        return p;
    });
    return WS_OK;
}

// src/bn128.js:671-718, in two steps: everything that needs only A and B1 (incl. the two 256-bit scalar
// multiplications of pi_c) can run while the GPU is still busy with B2 and H
struct EarlyParts { G1::Pt pi_a, s_pi_a, r_pib1; bool done = false; };
static void prove_assemble_early(ProvingKey* K, const MsmSums& M, Blinding& B, Blinding::Pre& pp, EarlyParts* E) {
    const G1::Pt alfa1 = G1::from_affine(K->alfa1), beta1 = G1::from_affine(K->beta1);
    pp = B.pre.get();
    // pi_a = sum A + alfa1 + r*delta1                               (:671-673)
    E->pi_a = G1::add(G1::add(alfa1, M.A), pp.r_delta1);
    // pib1 = sum B1 + beta1 + s*delta1                              (:681-683)
    const G1::Pt pib1 = G1::add(G1::add(beta1, M.B1), pp.s_delta1);
    E->s_pi_a = G1::mul_bytes(E->pi_a, B.sb(), 32);                  // (:692)
    E->r_pib1 = G1::mul_bytes(pib1, B.rb(), 32);                     // (:696)
    E->done = true;
}
static void prove_assemble(ProvingKey* K, const MsmSums& M, Blinding& B, Blinding::Pre& pp, EarlyParts& E, uint8_t* out384) {
    if (!E.done) prove_assemble_early(K, M, B, pp, &E);
    const G2::Pt beta2 = G2::from_affine(K->beta2);
    // pi_b = sum B2 + beta2 + s*delta2                              (:676-678)
    G2::Pt pi_b = G2::add(G2::add(beta2, M.B2), pp.s_delta2);
    // pi_c = sum C + sum H + s*pi_a + r*pib1 - (r*s)*delta1         (:687-704)
    G1::Pt pi_c = G1::add(M.C, M.H);
    pi_c = G1::add(pi_c, E.s_pi_a);
    pi_c = G1::add(pi_c, E.r_pib1);
    pi_c = G1::add(pi_c, G1::neg(pp.rs_delta1));
    // affine + fromMontgomery (:706-712); infinity prints as (0, 1, 0)
    Jac<Fq> a = G1::to_affine_jac(E.pi_a), c = G1::to_affine_jac(pi_c);
    Jac<Fq2> b = G2::to_affine_jac(pi_b);
    store_plain(out384 + 0, a.x); store_plain(out384 + 32, a.y); store_plain(out384 + 64, a.z);
    store_plain(out384 + 96, b.x.c0); store_plain(out384 + 128, b.x.c1);
    store_plain(out384 + 160, b.y.c0); store_plain(out384 + 192, b.y.c1);
    store_plain(out384 + 224, b.z.c0); store_plain(out384 + 256, b.z.c1);
    store_plain(out384 + 288, c.x); store_plain(out384 + 320, c.y); store_plain(out384 + 352, c.z);
}

// witness on the device, lane L held by the caller
static int groth16_prove(ProvingKey* K, Lane& L, const Fe* d_witness, const uint8_t* r32, const uint8_t* s32, uint8_t* out384,
                         hipStream_t s, const uint8_t* h_witness = nullptr) {
    if (!s) s = L.stream;
    Blinding B;
    int rc = start_blinding(K, r32, s32, &B);
    if (rc) return rc;
    MsmSums M;
    Blinding::Pre pp;
    EarlyParts E;
    if ((rc = prove_msms(K, L, d_witness, WindowShard{}, &M, s, [&](const MsmSums& m) { prove_assemble_early(K, m, B, pp, &E); }, false, nullptr, h_witness)))
        return rc;
    Trace tr;
    prove_assemble(K, M, B, pp, E, out384);
    tr.mark("assemble (host)");
    return WS_OK;
}

static int check_witness_len(ProvingKey* K, size_t witness_len) {
    if ((uint64_t)witness_len < (uint64_t)K->n_vars * 32) { set_last_error("witness shorter than nVars*32 bytes"); return WS_ERR_SIZE; }
    return WS_OK;
}

// ---- multi-GPU proving: per-rank partial sums, then one 576-byte record per rank to combine ----
// record = A | B1 | C | H (4 x 96 B G1) | B2 (192 B G2), Jacobian-Montgomery, affine-normalised
static int prove_partial_on(ProvingKey* K, Lane& L, const Fe* d_witness, WindowShard sh, uint8_t* out576, hipStream_t s, bool skip_h,
                            const uint8_t* h_witness = nullptr) {
    if (K->shard_world > 1) {
        // the handle IS the shard (its slice of the points, every window): the per-call rank / world must name the same one
        if (sh.off != K->shard_rank || sh.stride != K->shard_world) {
            set_last_error("prove_partial: the handle holds points shard " + std::to_string(K->shard_rank) + " of " + std::to_string(K->shard_world) +
                           ", the call names " + std::to_string(sh.off) + " of " + std::to_string(sh.stride));
            return WS_ERR_ARG;
        }
        sh = WindowShard{};
    }
    MsmSums M;
    int rc = prove_msms(K, L, d_witness, sh, &M, s ? s : L.stream, nullptr, skip_h, nullptr, h_witness);
    if (rc) return rc;
    Jac<Fq> j;
    j = G1::to_affine_jac(M.A); memcpy(out576, &j, 96);
    j = G1::to_affine_jac(M.B1); memcpy(out576 + 96, &j, 96);
    j = G1::to_affine_jac(M.C); memcpy(out576 + 192, &j, 96);
    j = G1::to_affine_jac(M.H); memcpy(out576 + 288, &j, 96);
    Jac<Fq2> j2 = G2::to_affine_jac(M.B2); memcpy(out576 + 384, &j2, 192);
    return WS_OK;
}
int groth16_prove_partial(ProvingKey* K, const uint8_t* witness, size_t witness_len, WindowShard sh, uint8_t* out576, bool skip_h) {
    Context* C = ctx();
    if (!C) return WS_ERR_NOINIT;
    int rc = check_witness_len(K, witness_len);
    if (rc) return rc;
    LaneLock L = acquire_lane(C);
    WS_HIP_CHECK(L->witness.reserve((size_t)K->n_vars * 32));
    return prove_partial_on(K, *L, L->witness.as<Fe>(), sh, out576, L->stream, skip_h, witness);
}
int groth16_prove_partial_dev(ProvingKey* K, const Fe* d_witness, size_t witness_len, WindowShard sh, uint8_t* out576, hipStream_t s, bool skip_h) {
    Context* C = ctx();
    if (!C) return WS_ERR_NOINIT;
    int rc = check_witness_len(K, witness_len);
    if (rc) return rc;
    LaneLock L = acquire_lane(C);
    return prove_partial_on(K, *L, d_witness, sh, out576, s, skip_h);
}
int pkey_eval_ab_dev(ProvingKey* K, const Fe* d_witness, size_t witness_len, Fe* d_a, Fe* d_b, hipStream_t s) {
    Context* C = ctx();
    if (!C) return WS_ERR_NOINIT;
    int rc = check_witness_len(K, witness_len);
    if (rc) return rc;
    LaneLock L = acquire_lane(C);
    return eval_ab_dev(*L, d_witness, K->n_vars, K->polsA, K->polsB, K->domain, d_a, d_b, s);
}
// the gather loop over the ranks' records + the proof assembly, with a blinding whose key-only scalar multiplications may
// already be under way
static void finish_records(ProvingKey* K, const uint8_t* partials, uint64_t n_ranks, Blinding& B, uint8_t* out384) {
    MsmSums M;
    M.A = M.B1 = M.C = M.H = G1::infinity();
    M.B2 = G2::infinity();
    for (uint64_t i = 0; i < n_ranks; i++) {
        const uint8_t* rec = partials + i * 576;
        Jac<Fq> j;
        memcpy(&j, rec, 96); M.A = G1::add(M.A, G1::from_jac(j));
        memcpy(&j, rec + 96, 96); M.B1 = G1::add(M.B1, G1::from_jac(j));
        memcpy(&j, rec + 192, 96); M.C = G1::add(M.C, G1::from_jac(j));
        memcpy(&j, rec + 288, 96); M.H = G1::add(M.H, G1::from_jac(j));
        Jac<Fq2> j2;
        memcpy(&j2, rec + 384, 192); M.B2 = G2::add(M.B2, G2::from_jac(j2));
    }
    Blinding::Pre pp;
    EarlyParts E;
    prove_assemble(K, M, B, pp, E, out384);
}
int groth16_prove_finish(ProvingKey* K, const uint8_t* partials, uint64_t n_ranks, const uint8_t* r32,
                         const uint8_t* s32, uint8_t* out384) {
    Blinding B;
    int rc = start_blinding(K, r32, s32, &B);
    if (rc) return rc;
    finish_records(K, partials, n_ranks, B, out384);
    return WS_OK;
}

static int whole_key_only(ProvingKey* K) {
    if (K->shard_world > 1) { set_last_error("this handle holds a points shard of the key: use prove_partial + prove_finish"); return WS_ERR_ARG; }
    return WS_OK;
}

// The H sum of a points-sharded key on its own (distributed CALC_H: the rank's slice of h in the layout the handle's hExps
// slice was loaded in).  out96: Jacobian-Montgomery, affine-normalised.
int pkey_h_msm_dev(ProvingKey* K, const Fe* d_h_local, uint64_t n, uint8_t* out96, hipStream_t s) {
    Context* C = ctx();
    if (!C) return WS_ERR_NOINIT;
    if (n != K->h_local) { set_last_error("pkey_h_msm: the handle holds " + std::to_string(K->h_local) + " hExps points"); return WS_ERR_SIZE; }
    LaneLock L = acquire_lane(C);
    if (!s) s = L->stream;
    msm_select_plan(*L, 0);
    uint32_t table_cw, table_ch;
    int rc = pkey_table_state(K, false, &table_cw, &table_ch);
    if (!rc) rc = msm_plan_dev(*L, d_h_local, n, WindowShard{}, s, table_ch);
    if (rc) return rc;
    int slot = -1;
    if ((rc = msm_g1_launch(*L, K->pointsH.as<Affine<Fq>>(), true, &slot, s))) return rc;
    XYZZ<Fq> r;
    if ((rc = msm_g1_finish(*L, slot, &r))) return rc;
    const Jac<Fq> j = G1::to_affine_jac(r);
    memcpy(out96, &j, 96);
    return WS_OK;
}

// One proof over the ranks of a node: this rank's partial sums (points shard), CALC_H on the distributed transform, one
// all-gather of the records, the same host-side assembly on every rank.  r32 / s32 NULL: rank 0 draws the blinding values
// and they travel in its slot of the first all-gather, so that every rank returns the same proof.
//
// Error agreement (the collectives pair up by call order, so a rank must never skip one its peers will post):
//   * everything that can fail locally without a collective -- argument checks, the grow-only buffer reserves of the
//     distributed CALC_H -- happens BEFORE the first collective, and its outcome travels in that collective;
//   * the first all-gather is ALWAYS posted (80-byte records: status | which of r, s were injected | the 64 blinding bytes),
//     whether the caller injected r, s or not: ranks that disagree about the injection, or about the injected values, all
//     return WSNARK_ERR_ARG instead of pairing a 64-byte gather with a 576-byte one;
//   * a rank that fails after that keeps posting the exchanges its peers are waiting in (`drain` below: the data no longer
//     matters) and reports its status in the trailer of the record all-gather; every rank then returns an error.
// What is left to the transport: a callback that fails or hangs on one rank (a dead peer, a broken link) -- the host's
// collectives library has to time out there, nothing in this file can.
static const size_t kDistHello = 80, kDistRecord = 576 + 16;
int groth16_prove_dist(ProvingKey* K, const Fe* d_witness, size_t witness_len, const DistComm& cm, const uint8_t* r32, const uint8_t* s32,
                       uint8_t* out384, hipStream_t s) {
    Context* C = ctx();
    if (!C) return WS_ERR_NOINIT;
    // (argument errors every rank sees alike -- a communicator that does not match the handle -- return before any collective)
    if (cm.world == 0 || cm.rank >= cm.world || K->shard_world != cm.world || K->shard_rank != cm.rank) {
        set_last_error("prove_dist: the handle must hold this communicator's points shard (wsnark_pkey_load_shard(rank, world, floor(log2(domain) / 2)))");
        return WS_ERR_ARG;
    }
    if (!cm.d_send || (cm.world > 1 && (!cm.d_recv || !cm.all_to_all || !cm.all_gather))) return WS_ERR_ARG;
    const uint32_t P = cm.world;
    LaneLock L = acquire_lane(C);
    // ---- local preflight: may fail on this rank only, so its result goes into the first collective ----
    int st = check_witness_len(K, witness_len);
    std::string st_msg = st ? get_last_error() : std::string();
    if (!st && (st = calc_h_dist_reserve(*L, cm, K->n_vars, K->domain, K->h_log_m))) st_msg = get_last_error();
    if (!st && L->h.reserve((size_t)K->domain * 32) != hipSuccess) { st = WS_ERR_HIP; st_msg = "prove_dist: device allocation of h failed"; }
    uint8_t hello[kDistHello];
    memset(hello, 0, sizeof hello);
    const uint32_t inj = (r32 ? 1u : 0u) | (s32 ? 2u : 0u);
    if (!st && inj != 3u && cm.rank == 0 && os_random(hello + 16, 64)) { st = WS_ERR_ARG; st_msg = "no entropy: getrandom(2) and /dev/urandom both failed"; }
    if (r32) memcpy(hello + 16, r32, 32);
    if (s32) memcpy(hello + 48, s32, 32);
    const uint32_t st_u = (uint32_t)st;
    memcpy(hello, &st_u, 4);
    memcpy(hello + 4, &inj, 4);
    uint8_t rs[64];
    if (P > 1) {
        std::vector<uint8_t> all((size_t)P * kDistHello);
        if (cm.all_gather(cm.user, hello, all.data(), kDistHello) != 0) { set_last_error("prove_dist: the all-gather callback failed"); return WS_ERR_ARG; }
        for (uint32_t q = 0; q < P; q++) {
            uint32_t qs, qi;
            memcpy(&qs, &all[q * kDistHello], 4);
            memcpy(&qi, &all[q * kDistHello + 4], 4);
            if (qs) {
                set_last_error(q == cm.rank ? st_msg : "prove_dist: rank " + std::to_string(q) + " failed before the proof started (status " + std::to_string(qs) + ")");
                return (int)qs;
            }
            if (qi != inj) { set_last_error("prove_dist: the ranks disagree about which of r, s are injected (every rank must pass the same r32 / s32, or all NULL)"); return WS_ERR_ARG; }
        }
        memcpy(rs, &all[16], 64);                                        // rank 0's slot: its draw, overlaid with what it injected
        for (uint32_t q = 1; q < P; q++) {
            const uint8_t* o = &all[q * kDistHello + 16];
            if ((r32 && memcmp(o, rs, 32)) || (s32 && memcmp(o + 32, rs + 32, 32))) {
                set_last_error("prove_dist: the ranks injected different r / s");
                return WS_ERR_ARG;
            }
        }
    } else {
        if (st) { set_last_error(st_msg); return st; }
        memcpy(rs, hello + 16, 64);
    }
    // the blinding values are known: the scalar multiplications that involve key points only (r delta1, s delta1, rs delta1,
    // s delta2: ~0.5 ms on a host thread) run under the GPU work, as in the one-call prover
    Blinding B;
    int rc = start_blinding(K, rs, rs + 32, &B);
    uint8_t rec[kDistRecord];
    memset(rec, 0, sizeof rec);
    int exchanges = 0;
    uint32_t log_n = 0;
    while ((1u << log_n) < K->domain) log_n++;
    if (!rc) {
        MsmSums M;
        const CalcHFn calc_h = [&](hipStream_t s2, Fe* d_h) {
            return calc_h_dist(*L, cm, d_witness, K->n_vars, K->polsA, K->polsB, K->domain, K->h_log_m, d_h, s2, &exchanges);
        };
        rc = prove_msms(K, *L, d_witness, WindowShard{}, &M, s ? s : L->stream, nullptr, false, calc_h);
        if (!rc) {
            Jac<Fq> j;
            j = G1::to_affine_jac(M.A); memcpy(rec, &j, 96);
            j = G1::to_affine_jac(M.B1); memcpy(rec + 96, &j, 96);
            j = G1::to_affine_jac(M.C); memcpy(rec + 192, &j, 96);
            j = G1::to_affine_jac(M.H); memcpy(rec + 288, &j, 96);
            const Jac<Fq2> j2 = G2::to_affine_jac(M.B2); memcpy(rec + 384, &j2, 192);
        }
    }
    const std::string rc_msg = rc ? get_last_error() : std::string();
    if (rc && P > 1) {
        // drain: the peers are (or will be) waiting in the exchanges this rank has not posted yet: 3, 2, 1 vectors of n / P^2
        // elements per peer.  The send buffer's contents no longer matter -- every rank is about to learn of the failure.
        static const uint64_t vecs[3] = {3, 2, 1};
        const uint64_t per = ((uint64_t)K->domain / P / P) * sizeof(Fe);
        for (int e = exchanges; e < 3; e++) (void)cm.all_to_all(cm.user, vecs[e] * per, (void*)L->stream2);
        (void)hipStreamSynchronize(L->stream2);
    }
    const uint32_t rc_u = (uint32_t)rc;
    memcpy(rec + 576, &rc_u, 4);
    std::vector<uint8_t> parts((size_t)P * 576);
    if (P > 1) {
        std::vector<uint8_t> all((size_t)P * kDistRecord);
        if (cm.all_gather(cm.user, rec, all.data(), kDistRecord) != 0) { set_last_error("prove_dist: the all-gather callback failed"); return WS_ERR_ARG; }
        for (uint32_t q = 0; q < P; q++) {
            uint32_t qs;
            memcpy(&qs, &all[q * kDistRecord + 576], 4);
            if (qs) {
                set_last_error(q == cm.rank ? rc_msg : "prove_dist: rank " + std::to_string(q) + " failed (status " + std::to_string(qs) + ")");
                return (int)qs;
            }
            memcpy(&parts[(size_t)q * 576], &all[q * kDistRecord], 576);
        }
    } else {
        if (rc) { set_last_error(rc_msg); return rc; }
        memcpy(parts.data(), rec, 576);
    }
    finish_records(K, parts.data(), P, B, out384);
    return WS_OK;
}

int groth16_prove_host_witness(ProvingKey* K, const uint8_t* witness, size_t witness_len, const uint8_t* r32,
                               const uint8_t* s32, uint8_t* out384) {
    Context* C = ctx();
    if (!C) return WS_ERR_NOINIT;
    int rc = whole_key_only(K);
    if (rc) return rc;
    rc = check_witness_len(K, witness_len);
    if (rc) return rc;
    LaneLock L = acquire_lane(C);
    WS_HIP_CHECK(L->witness.reserve((size_t)K->n_vars * 32));
    return groth16_prove(K, *L, L->witness.as<Fe>(), r32, s32, out384, L->stream, witness);
}

int groth16_prove_dev_witness(ProvingKey* K, const Fe* d_witness, size_t witness_len, const uint8_t* r32,
                              const uint8_t* s32, uint8_t* out384, hipStream_t s) {
    Context* C = ctx();
    if (!C) return WS_ERR_NOINIT;
    int rc = whole_key_only(K);
    if (rc) return rc;
    rc = check_witness_len(K, witness_len);
    if (rc) return rc;
    LaneLock L = acquire_lane(C);
    return groth16_prove(K, *L, d_witness, r32, s32, out384, s);
}

}  // namespace wsnark
