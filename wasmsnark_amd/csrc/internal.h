// internal.h -- C++ interfaces between the translation units of libwsnark.
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "curve.h"
#include "curve_pair.h"
#include "rt.h"

namespace wsnark {

// ---- per-kernel timing (HIP events on the library's stream) ----
struct KernelTimer {
    bool enabled = false;
    bool dominant_only = false;   // mode 2: bracket only the kernel the roofline is quoted on (msm_accumulate_*)
    std::mutex mu;                // callers on several lanes bracket concurrently
    unsigned long generation = 0; // bumped by collect(): a bracket opened before it must not touch the new record list
    struct Rec { const char* name; hipEvent_t a, b; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;   // events are recycled: creating them costs more than recording them
    std::map<std::string, std::pair<double, uint64_t>> acc;   // name -> (total ms, launches)
    void begin(const char* name, hipStream_t s);
    void end(hipStream_t s);
    void collect();     // sync + fold recs into acc
    void reset();
};

// Users of a context-wide scratch area (NTT ping-pong buffer, CALC_H work arrays) come from any host thread on
// any stream: one of them enqueues at a time, and on the device each one starts after the previous one -- which may
// have run on another stream -- has finished.
struct ScratchChain {
    std::mutex mu;
    hipEvent_t done = nullptr;
    hipStream_t last = nullptr;
};
struct ScratchGuard {
    ScratchChain& c;
    hipStream_t s;
    std::unique_lock<std::mutex> lk;
    ScratchGuard(ScratchChain& c_, hipStream_t s_) : c(c_), s(s_), lk(c_.mu) {
        if (c.done && c.last != s) (void)hipStreamWaitEvent(s, c.done, 0);
    }
    ~ScratchGuard() {
        if (!c.done) (void)hipEventCreateWithFlags(&c.done, hipEventDisableTiming);
        (void)hipEventRecord(c.done, s);
        c.last = s;
    }
};

struct NttPlan;       // ntt.hip
struct MsmWorkspace;  // msm.hip

// Multi-GPU window sharding, given PER CALL: the call computes only the Pippenger windows w with
// w >= off && (w - off) % stride == 0 (rank, world of the C ABI); its result is then a partial sum already scaled
// by 2^(c*w).  {0, 1} = all windows.
struct WindowShard {
    uint32_t off = 0, stride = 1;
};

// A lane = everything one MSM / CALC_H / proof in flight needs besides the (read-only) key: two in-order queues,
// the MSM plans and launch slots, the transform scratch, the per-proof witness / h buffers.  A context has a few
// lanes (WSNARK_LANES, default 2), so that two callers (two proofs, or a proof and an MSM) overlap on one GPU: the
// reduction tails of one are latency chains on a few hundred wavefronts, the other's full-width kernels take the
// SIMDs they leave idle.  A caller holds the lane's mutex for the duration of its call.
struct Lane {
    int id = 0;
    std::mutex mu;
    hipStream_t stream = nullptr;               // first in-order queue (or the caller's stream for _dev entry points)
    hipStream_t stream2 = nullptr;              // second queue (prover: CALC_H and the H sum beside the tails)
    hipStream_t stream3 = nullptr;              // third queue (small proofs / points shards: the G2 sum beside the G1 sums)
    hipStream_t stream_copy = nullptr;          // host -> device copies of a call's inputs (the witness, chunk by chunk, beside the first kernels)
    hipEvent_t ev_chunk[2] = {nullptr, nullptr};   // ... a chunk has landed
    MsmWorkspace* msm = nullptr;                // plans, launch slots (owned; msm_workspace_free)
    DevBuf host_in[2];                          // host-pointer boundary: grow-only device copies of the caller's buffers
    DevBuf ntt_scratch;                         // ping-pong buffer of the multi-pass transforms
    DevBuf calch_buf[4];                        // sigM, A, B, E
    DevBuf dist_buf[2];                         // distributed CALC_H: the rank's slices of (a, b, E), ping-pong (dist.hip)
    ScratchChain ntt_chain, calch_chain;        // who may touch ntt_scratch / calch_buf next (calls return before the GPU is done)
    DevBuf witness, h;                          // per-proof device buffers (grow-only)
    hipEvent_t ev_start = nullptr, ev_tail = nullptr, ev_h = nullptr;   // cross-queue ordering of one proof
    hipEvent_t ev_plan = nullptr, ev_g2 = nullptr;                      // ... witness plan ready / G2 sum enqueued (third queue)
};
static const int kMaxLanes = 4;

struct StagePool;     // context.hip: the copy threads of the staging ring
// Everything that belongs to ONE device: queues, lanes, plan caches, the staging ring.  A process may hold several (round 5:
// wsnark_group_create makes one per device of a group; wsnark_init makes the default one); a host thread works on the one it
// has selected (CtxScope) -- HIP's own current device is per thread in the same way.
struct Context {
    int device = 0;
    hipStream_t stream = nullptr;               // utility queue: key ingestion, table builds, small host-pointer calls
    int num_cu = 256;
    std::string devinfo;                        // "<name> <arch> CUs=<n>" of THIS context's device (wsnark_device_info)
    int n_lanes = 2;
    Lane lanes[kMaxLanes];
    std::mutex lane_mu;                         // waiting for a lane: released lanes are announced on lane_cv
    std::condition_variable lane_cv;
    std::mutex mu;                              // NTT plan cache
    std::map<int, std::shared_ptr<NttPlan>> ntt_plans;   // by log2(n)
    KernelTimer timer;
    // host-pointer boundary: pinned staging ring for uploads
    void* pin_ring = nullptr;
    size_t pin_ring_bytes = 0;
    size_t pin_chunk_last = 0;                  // chunk size of the last upload through the ring (slot geometry of the events)
    hipEvent_t pin_ev[128] = {};                // one per ring slot: the slot's last DMA
    bool pin_ev_rec[128] = {};                  // ... and whether it has ever been recorded: a slot no DMA has left is free without a
                                                // call into the runtime (and an event is only ever looked at after a record on a queue
                                                // of THIS context: context.hip, queue_of_context)
    hipStream_t load_q[2] = {nullptr, nullptr}; // key loads: the two matrices are transposed side by side on these
    hipStream_t build_q = nullptr;              // the keys' background table builds, one after the other: the LOWEST stream priority (prove.hip)
    DevBuf build_tmp;                           // ... and their scratch slab (the builds are serial on build_q: one slab serves them all); grow-only
    std::mutex build_mu;                        // queueing a build (and growing the slab) is one key at a time
    std::thread *warm = nullptr, *warm_ring = nullptr;   // the context's start-up helpers: code objects; staging ring (joined by its destruction)
    std::mutex warm_mu;
    int owner_pid = 0;                          // the process that created the helpers: a fork()ed child has the objects but not the threads
    std::mutex ring_mu;                         // one upload at a time uses the ring
    StagePool* pool = nullptr;                  // ... and its copy threads (never destroyed: parked until the process ends)
    std::atomic<bool> ntt_attr_set{false};      // the transform kernel's dynamic-LDS attribute has been raised on this device
};
// the context the calling thread works on: the one selected by a CtxScope on this thread, else the default one (wsnark_init)
Context* ctx();   // nullptr before wsnark_init
Context* ctx_set_current(Context* c);          // returns the previous selection of this thread (nullptr = the default)
struct CtxScope {
    Context* prev;
    int prev_device = -1;          // the thread's device is the CALLER's too (torch reads hipGetDevice): put back what was there
    explicit CtxScope(Context* c) : prev(ctx_set_current(c)) {
        if (!c) return;
        int d = -1;
        if (hipGetDevice(&d) == hipSuccess && d != c->device) prev_device = d;
        (void)hipSetDevice(c->device);
    }
    ~CtxScope() {
        (void)ctx_set_current(prev);
        if (prev_device >= 0) (void)hipSetDevice(prev_device);
    }
    CtxScope(const CtxScope&) = delete;
    CtxScope& operator=(const CtxScope&) = delete;
};
// the calling thread's device as it was (group create / free walk over devices on the CALLER's thread: torch reads hipGetDevice)
struct DeviceRestore {
    int prev = -1;
    DeviceRestore() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~DeviceRestore() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceRestore(const DeviceRestore&) = delete;
    DeviceRestore& operator=(const DeviceRestore&) = delete;
};
int context_create(int device, Context** out, bool wrap = false);  // a context of its own on that device (wsnark_group_create)
void context_destroy(Context* C);
// One no-op kernel per translation unit: the runtime loads a TU's code object when the first of its kernels is launched (7-12 ms for
// msm.hip under ROCm 7.2) -- wsnark_init's helper thread launches these so that the first key load and proof do not pay for it
void warm_msm(hipStream_t s);
void warm_ntt(hipStream_t s);
void warm_calch(hipStream_t s);
void warm_dist(hipStream_t s);
void warm_fixedbase(hipStream_t s);
void context_join_warm(Context* C);
#define WS_DEFINE_WARM(tu) \
    __global__ void warm_kernel_##tu(int) {} \
    void warm_##tu(hipStream_t s) { hipLaunchKernelGGL(warm_kernel_##tu, dim3(1), dim3(64), 0, s, 0); (void)hipGetLastError(); }

// RAII: a free lane; with every lane busy the caller waits until ANY of them is released (not for one picked in advance:
// a caller must not queue behind a long proof while another lane has already come free)
struct LaneLock {
    Lane* L = nullptr;
    Context* C = nullptr;
    std::unique_lock<std::mutex> lk;
    LaneLock() {}
    LaneLock(LaneLock&& o) : L(o.L), C(o.C), lk(std::move(o.lk)) { o.L = nullptr; o.C = nullptr; }
    LaneLock(const LaneLock&) = delete;
    LaneLock& operator=(const LaneLock&) = delete;
    ~LaneLock();
    Lane* operator->() const { return L; }
    Lane& operator*() const { return *L; }
};
LaneLock acquire_lane(Context* C);

// Host -> device copy of caller memory that may be pageable and never touched by the runtime before: worker
// threads memcpy chunks into a pinned ring and queue one DMA per chunk on `s` (the runtime's own pageable
// path pins fresh pages at ~10 GB/s; this runs at memcpy speed, ~40 GB/s).  Returns once the source has been
// read completely (the caller may free or rewrite it); the DMAs out of the ring may still be in flight on `s`.  A source that
// is already pinned is DMA'd in place, and then the call WAITS for `s`: the contract is the same for both kinds of memory.
int upload_staged(void* d_dst, const void* h_src, size_t bytes, hipStream_t s);
// The same, announcing every chunk: on_chunk(lo, hi) runs on the calling thread right after the DMA of bytes [lo, hi) has
// been queued on `s` (chunks arrive in order; a non-zero return aborts the upload and is returned).  Typical use: record an
// event on `s` and make another queue start its first pass over that part while the rest is still being staged.
// (the third argument of on_chunk is the queue that chunk's DMA went to: `s`)
// A pinned source is DMA'd in place and the call returns with those DMAs in flight: the source must stay valid until `s` has
// passed them (*direct_out, optional, says whether that path was taken).
typedef std::function<int(size_t, size_t, hipStream_t)> ChunkFn;
int upload_pipelined(void* d_dst, const void* h_src, size_t bytes, hipStream_t s, const ChunkFn& on_chunk, bool* direct_out = nullptr);

// Measurement switches (A/B runs): the value set through wsnark_tuning_set for `name`, else the environment variable
// WSNARK_<name>, else dflt.  Read per call, so that ONE process can time several settings on the same resident key
// (a gpurun box is charged by the minute; a fresh process per setting pays the key setup every time).
long tuning_get(const char* name, long dflt);
double tuning_get_real(const char* name, double dflt);      // fraction-capable (WSNARK_TABLE_MAX_GB=0.5); negatives read as 0
void tuning_set(const char* name, long value);     // value == LONG_MIN: forget the override

// ---- NTT (ntt.hip) ----
// In-place transform of n Montgomery Fr elements resident on the device.
// Semantics of the reference's fft_fft / fft_ifft (src/build_fft.js:159-221):
//   forward: y[k] = sum_i x[i] * w_{2n}^{(2k+odd) i}       (natural order in and out)
//   inverse: rawfft, then y[i] = raw[(n-i) mod n] / n
// The multi-pass scratch comes from lane L (the caller holds it).
// count > 1: that many independent transforms stored back to back (the row / column steps of the four-step transform)
int ntt_dev(Lane& L, Fe* d_data, uint64_t n, int odd, int inverse, hipStream_t s, uint64_t count = 1);
// the general form: reads d_src (times d_in2 element-wise, if given), writes d_dst; combine_e (inverse only): the last
// pass stores CALC_H's h[t] = fromMontgomery((e[t] - w_2n^-t v[t]) / 2) instead of the transform v (calch.hip)
// rc_pre (batched column step of the distributed transform, dist.hip): transform b of the batch is row (b & row_mask) of a rank's
// block of a longer vector; its element g is multiplied on load by that vector's coset factor w_2N^(row0 + row + (g << shift))
// (tables of the LONGER length, ntt_coset_tables_kernel_format) -- the coset pre-scale without a pass of its own
struct NttRowCoset { const Fe* lo; const Fe* hi; uint32_t hc, shift, row0, row_mask; };
// rc_post (the same column step's LAST pass): element c of transform b = (vector << lr1 | row r) leaves multiplied by the four-step
// twiddle w_N^(+-(row0 + r) c) (lo / hi: ntt_twiddle_tables of the longer length N, internal form) at out[q][vector][r][c2],
// c = q 2^lr2 + c2 -- the block order of the exchange -- instead of being stored in place and re-read by a packing pass
struct NttRowPost { Fe* out; const Fe* lo; const Fe* hi; uint32_t h, lr1, lr2, k; uint64_t row0; };
// rc_gather (the row step's FIRST pass): transform b = (vector << lr2 | c2) reads its element i1 = q 2^lr1 + r from in[q][vector][r][c2],
// the receive buffer of the exchange, instead of from a transposed copy
struct NttRowGather { const Fe* in; uint32_t lr1, lr2, k; };
int ntt_run(Lane& L, const Fe* d_src, const Fe* d_in2, Fe* d_dst, const Fe* combine_e, uint64_t n, int odd, int inverse,
            hipStream_t s, uint64_t count = 1, const NttRowCoset* rc_pre = nullptr, const NttRowPost* rc_post = nullptr,
            const struct NttRowGather* rc_gather = nullptr);
int ntt_coset_tables_kernel_format(int bits, const Fe** lo, const Fe** hi, uint32_t* hc, hipStream_t s);
// internal = true: the same tables in the internal form of the radix-2^29 field (entries x 2^5), as the transform kernels read them
int ntt_twiddle_tables(int bits, int inverse, const Fe** lo, const Fe** hi, int* h, hipStream_t s, bool internal = false);

int ntt_coset_tables(int bits, const Fe** lo, const Fe** hi, int* hc, Fe* n_inv, hipStream_t s);

// ---- MSM (msm.hip) ----
// sum_i scalars[i] * points[i]; scalars raw 256-bit LE (not reduced), points affine
// Montgomery (x == 0 => infinity).  Result written to host memory as the reference's
// Jacobian-Montgomery triple, affine-normalised: (x, y, 1) or (0, 1, 0).
// Every function works on lane L, which the caller holds.
int msm_g1_dev(Lane& L, const Fe* d_scalars, const Affine<Fq>* d_points, uint64_t n, WindowShard sh, Jac<Fq>* out_host, hipStream_t s);
int msm_g2_dev(Lane& L, const Fe* d_scalars, const Affine<Fq2>* d_points, uint64_t n, WindowShard sh, Jac<Fq2>* out_host, hipStream_t s);
// the same from host buffers (staged upload; the plan is built while the points are still on their way)
int msm_g1_host(Lane& L, const void* h_scalars, const void* h_points, uint64_t n, WindowShard sh, Jac<Fq>* out_host);
int msm_g2_host(Lane& L, const void* h_scalars, const void* h_points, uint64_t n, WindowShard sh, Jac<Fq2>* out_host);
// two-phase form: one digit/sort/task plan per scalar vector, then any number of point sets of the
// same length against it (the prover's A, B1, B2 and C sums all use the witness as scalars).
// table_c != 0: plan for fixed-base window tables of that window width (msm_build_table; the launches then take the table)
int msm_plan_dev(Lane& L, const Fe* d_scalars, uint64_t n, WindowShard sh, hipStream_t s, uint32_t table_c = 0);
// the same plan in three steps: the digit histogram -- the first pass over the scalars -- may be taken over parts [i0, i1) of
// the vector as they become available (a witness uploaded chunk by chunk), on `s` or on queues ordered before the finish
int msm_plan_begin(Lane& L, uint64_t n, WindowShard sh, hipStream_t s, uint32_t table_c = 0);
int msm_plan_count(Lane& L, const Fe* d_scalars, uint64_t i0, uint64_t i1, hipStream_t s);
int msm_plan_finish(Lane& L, const Fe* d_scalars, hipStream_t s);
uint32_t msm_table_rows(uint32_t table_c);
uint32_t msm_table_window(uint64_t n);
int msm_build_table(int which, void* d_table, uint64_t n, uint32_t table_c, hipStream_t s, void* d_tmp = nullptr, size_t tmp_bytes = 0);
size_t msm_table_scratch_bytes(uint64_t lanes);      // scratch of the build in short launches, for `lanes` points at a time   // rows 1.. from row 0 (device domain: after msm_prepare_points)
// mask[i] = 0 where the point is infinity (x == 0, reference format) in every given set; *skipped_host = how many
int msm_points_mask(const Affine<Fq>* d_g1, const Affine<Fq2>* d_g2, uint64_t n, uint8_t* d_mask, uint32_t* skipped_host, hipStream_t s);
// asynchronous form: launch enqueues the kernels and the copy of the window sums, finish waits
// for that copy and runs the serial host tail.  `prepared` = the point array was converted in place by
// msm_prepare_points (resident keys).
int msm_g1_launch(Lane& L, const Affine<Fq>* d_points, bool prepared, int* slot, hipStream_t s);
int msm_g2_launch(Lane& L, const Affine<Fq2>* d_points, bool prepared, int* slot, hipStream_t s);
// plan_ids (optional): the plan each set is accumulated against (variants of one plan: same geometry)
int msm_g1_launch_batch(Lane& L, const Affine<Fq>* const* d_points, int nsets, bool prepared, int* slots, hipStream_t s, const int* plan_ids = nullptr);
// several independent plans (digit/sort/task buffers) can be alive at once; plan and launches use the selected one
void msm_select_plan(Lane& L, int id);      // id in [0, 4)
// A second plan over the SAME scalars that leaves out the pairs with mask[i] == 0, derived from plan `src_id`
// (which must just have been built by msm_plan_dev): only the per-bin counting sort and the task list are redone,
// the digit extraction and the coarse scatter are shared.  Becomes plan `dst_id`.
int msm_plan_variant(Lane& L, int src_id, int dst_id, const uint8_t* d_mask, hipStream_t s);
bool msm_ready(Lane& L, int slot);     // the launch's window sums have reached the host (finish will not block)
int msm_g1_finish(Lane& L, int slot, XYZZ<Fq>* out_host);
int msm_g2_finish(Lane& L, int slot, XYZZ<Fq2>* out_host);
// error paths: forget the given launches (their results will never be collected); waits for the streams first
void msm_abort_slots(Lane& L, const int* slots, int nslots, hipStream_t a, hipStream_t b);
void msm_workspace_free(Lane& L);
int msm_prepare_points(int which, void* d_points, uint64_t n, hipStream_t s);

// ---- CALC_H pieces (calch.hip) ----
struct CsrMatrix {            // row-major transpose of the reference's column-major pols blob
    uint32_t n_rows = 0;      // domain size
    uint32_t n_cols = 0;      // nSignals
    uint64_t nnz = 0;
    DevBuf row_ptr;           // (n_rows + 1) x u32
    DevBuf col;               // nnz x u32 (signal index)
    DevBuf coef;              // nnz x Fe (Fr Montgomery)
};
// parse `ncoefs, (idx, coef)*` records (src/build_pol.js:62-144) for n_signals columns;
// returns bytes consumed via *consumed.
// release: see KeySections::release (nullptr: the records are walked, then uploaded in one piece)
int pols_to_csr(const uint8_t* pols, size_t len, uint32_t n_signals, uint32_t domain, CsrMatrix* out,
                size_t* consumed, hipStream_t s, void (*release)(const void* p, size_t n) = nullptr);
// d_h_out[domain] (plain form) from a device-resident plain witness; work arrays of lane L
int calc_h_dev(Lane& L, const Fe* d_signals_plain, uint32_t n_signals, const CsrMatrix& A, const CsrMatrix& B,
               uint32_t domain, Fe* d_h_out, hipStream_t s);
int fr_map_dev(const Fe* d_in, Fe* d_out, uint64_t n, int to_mont, hipStream_t s);
int fr_mul_dev(const Fe* d_a, const Fe* d_b, Fe* d_out, uint64_t n, hipStream_t s);
int eval_ab_dev(Lane& L, const Fe* d_signals_plain, uint32_t n_signals, const CsrMatrix& A, const CsrMatrix& B, uint32_t domain,
                Fe* d_a, Fe* d_b, hipStream_t s);
int dist_combine_dev(const Fe* d_e, const Fe* d_o, Fe* d_h, uint64_t rows, uint64_t cols, uint64_t row0, uint32_t log_n1, uint32_t log_n, hipStream_t s);
int dist_scale_dev(Fe* d_data, uint64_t stack, uint64_t rows, uint64_t cols, uint64_t row0, uint32_t log_n1, uint32_t log_n, int mode, int inverse, hipStream_t s);

// ---- proving keys (prove.hip) ----
struct ProvingKey;
struct KeySections {      // everything wsnark_pkey_load reads from proving_key.bin, as separate host buffers
    uint32_t n_vars, n_public, domain;
    const uint8_t *alfa1, *beta1, *delta1, *beta2, *delta2;
    const uint8_t* polsA; uint64_t lenA;
    const uint8_t* polsB; uint64_t lenB;
    const uint8_t *A, *B1, *B2, *Cpts, *H;     // nVars, nVars, nVars, nVars-nPublic-1, domain points
    uint64_t lenPA, lenPB1, lenPB2, lenPC, lenPH;   // bytes the caller vouches for behind each of those
    // Optional (the file loader, keyfile.hip): called with every source range the load has finished reading -- a mapped file drops
    // those pages again (madvise), so a load never holds more of the file than the ranges in flight.
    void (*release)(const void* p, size_t n) = nullptr;
};
struct KeyShard { uint32_t rank = 0, world = 1, h_log_m = 0; };
int pkey_parse(const uint8_t* buf, size_t len, KeySections* out);
// a key FILE (keyfile.hip): proving_key.bin, or the u64-offset container for keys beyond its 4 GiB (include/wsnark.h: WSNARK64)
struct KeyFile {
    int fd = -1;
    const uint8_t* base = nullptr;
    size_t len = 0;
    int format = 0;                     // 1 = proving_key.bin (u32 offsets), 2 = WSNARK64 container
    KeyFile() = default;
    KeyFile(const KeyFile&) = delete;
    KeyFile& operator=(const KeyFile&) = delete;
    ~KeyFile();
};
int keyfile_open(const char* path, KeyFile* F, KeySections* S);      // maps the file read-only, bounds-checks the header; S points into the map
int pkey_load_sections(const KeySections& S, ProvingKey** out, KeyShard shard);      // on the calling thread's context
void pkey_free(ProvingKey* K);
int pkey_wait_tables(ProvingKey* K);
Context* pkey_context(const ProvingKey* K);
void pkey_info(const ProvingKey* K, uint32_t* nv, uint32_t* np, uint32_t* dom);

// ---- one proof over the ranks of a node (dist.hip) ----
// The host's transport (include/wsnark.h: wsnark_comm_t): exchange buffers it owns and two callbacks.
struct DistComm {
    uint32_t rank = 0, world = 1;
    Fe* d_send = nullptr;
    Fe* d_recv = nullptr;
    uint64_t buf_bytes = 0;
    int (*all_to_all)(void* user, uint64_t bytes_per_rank, void* stream) = nullptr;
    int (*all_gather)(void* user, const void* send, void* recv, uint64_t bytes) = nullptr;
    void* user = nullptr;
};
// the rank's slice of h (plain form, 2^l2-interleaved rows of the rank), three exchanges
// *exchanges (optional): how many of the three all-to-alls this rank has posted when the call returns
int calc_h_dist(Lane& L, const DistComm& cm, const Fe* d_signals_plain, uint32_t n_signals, const CsrMatrix& A, const CsrMatrix& B,
                uint32_t domain, uint32_t l2_expected, Fe* d_h_local, hipStream_t s, int* exchanges = nullptr);
// its checks and buffer reserves alone (what can fail on one rank before any collective)
int calc_h_dist_reserve(Lane& L, const DistComm& cm, uint32_t n_signals, uint32_t domain, uint32_t l2_expected);

// One row of pol_constructLC (src/build_pol.js:62-144): sum_k coef[k] * sig[col[k]] over the row's CSR range, on the radix-2^29 field
// (round 6, session 3: the kernel was priced by the counters at 0.78 of its ISSUE floor -- not gather latency -- with the saturated
// field's 584-instruction product; a wavefront runs as many terms as its longest row).  Terms go in PAIRS through the fused double
// product (one Montgomery reduction for two terms), the rest through a single one; the sum is kept in [0, 2r).  coef is c R^2 in the
// reference's Montgomery form (R = 2^256: calch.hip pols_to_csr), sig the plain signal -- any 256-bit value --, so a product here is
// c w R 2^-5 on this field's radix (2^261); one closing product by 2^266 mod r (the field's to_internal constant) makes the row's sum
// the Montgomery term sum c w R, canonical: bit for bit what the saturated field gave.
__device__ __forceinline__ Fe lc_row_dot(const Fe* __restrict__ coef, const uint32_t* __restrict__ col, const Fe* __restrict__ sig,
                                         uint32_t k, uint32_t e) {
    typedef Fr29 F;
    F29 acc = F::zero();
    for (; k + 1 < e; k += 2)
        acc = F::add(acc, F::mul2add_inl(F::unpack(coef[k]), F::unpack(sig[col[k]]), F::unpack(coef[k + 1]), F::unpack(sig[col[k + 1]])));
    if (k < e) acc = F::add(acc, F::mul_inl(F::unpack(coef[k]), F::unpack(sig[col[k]])));
    return F::pack(F::canonical(F::mul_inl(acc, F::from_words(Fr29Params::CIN0, Fr29Params::CIN1, Fr29Params::CIN2, Fr29Params::CIN3))));
}

}  // namespace wsnark
