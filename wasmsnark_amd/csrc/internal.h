// internal.h -- C++ interfaces between the translation units of libwsnark.
#pragma once
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "curve.h"
#include "rt.h"

namespace wsnark {

// ---- per-kernel timing (HIP events on the library's stream) ----
struct KernelTimer {
    bool enabled = false;
    bool dominant_only = false;   // mode 2: bracket only the kernel the roofline is quoted on (msm_accumulate_*)
    bool skipped = false;         // the begin() of the current bracket was filtered out
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

struct NttPlan;    // ntt.hip
struct MsmScratch; // msm.hip

struct Context {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;              // second in-order queue (prover: CALC_H and the H sum beside the tails)
    int num_cu = 256;
    uint32_t shard_off = 0, shard_stride = 1;   // MSM window sharding (wsnark_set_window_shard)
    std::mutex mu;
    std::map<int, std::shared_ptr<NttPlan>> ntt_plans;   // by log2(n)
    DevBuf ntt_scratch;
    std::shared_ptr<MsmScratch> msm_scratch[4];          // buffers of the selectable plans (msm_select_plan)
    DevBuf calch_buf[4];                                 // sigM, A, B, E
    ScratchChain ntt_chain, calch_chain;                 // who may touch ntt_scratch / calch_buf next
    KernelTimer timer;
    // host-pointer boundary: pinned staging ring for uploads, grow-only device copies of the caller's buffers
    void* pin_ring = nullptr;
    hipEvent_t pin_ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    DevBuf host_in[2];
};

// Host -> device copy of caller memory that may be pageable and never touched by the runtime before: worker
// threads memcpy chunks into a pinned ring and queue one DMA per chunk on `s` (the runtime's own pageable
// path pins fresh pages at ~10 GB/s; this runs at memcpy speed, ~40 GB/s).  Returns once the source has been
// read completely; the DMAs may still be in flight on `s`.
int upload_staged(void* d_dst, const void* h_src, size_t bytes, hipStream_t s);

Context* ctx();   // nullptr before wsnark_init

// ---- NTT (ntt.hip) ----
// In-place transform of n Montgomery Fr elements resident on the device.
// Semantics of the reference's fft_fft / fft_ifft (src/build_fft.js:159-221):
//   forward: y[k] = sum_i x[i] * w_{2n}^{(2k+odd) i}       (natural order in and out)
//   inverse: rawfft, then y[i] = raw[(n-i) mod n] / n
int ntt_dev(Fe* d_data, uint64_t n, int odd, int inverse, hipStream_t s);

int ntt_coset_tables(int bits, const Fe** lo, const Fe** hi, int* hc, Fe* n_inv, hipStream_t s);

// ---- MSM (msm.hip) ----
// sum_i scalars[i] * points[i]; scalars raw 256-bit LE (not reduced), points affine
// Montgomery (x == 0 => infinity).  Result written to host memory as the reference's
// Jacobian-Montgomery triple, affine-normalised: (x, y, 1) or (0, 1, 0).
int msm_g1_dev(const Fe* d_scalars, const Affine<Fq>* d_points, uint64_t n, Jac<Fq>* out_host, hipStream_t s);
// the same from host buffers (staged upload; the plan is built while the points are still on their way)
int msm_g1_host(const void* h_scalars, const void* h_points, uint64_t n, Jac<Fq>* out_host);
int msm_g2_host(const void* h_scalars, const void* h_points, uint64_t n, Jac<Fq2>* out_host);
int msm_g2_dev(const Fe* d_scalars, const Affine<Fq2>* d_points, uint64_t n, Jac<Fq2>* out_host, hipStream_t s);
// XYZZ result left to the caller (host memory), no affine normalisation
// `prepared` = the point array was converted in place by msm_prepare_points (resident keys)
int msm_g1_dev_xyzz(const Fe* d_scalars, const Affine<Fq>* d_points, uint64_t n, XYZZ<Fq>* out_host, hipStream_t s,
                    bool prepared = false);
int msm_g2_dev_xyzz(const Fe* d_scalars, const Affine<Fq2>* d_points, uint64_t n, XYZZ<Fq2>* out_host, hipStream_t s,
                    bool prepared = false);
// two-phase form: one digit/sort/task plan per scalar vector, then any number of point sets of the
// same length against it (the prover's A, B1, B2 and C sums all use the witness as scalars).
// Callers serialise on Context::mu.
// d_mask (optional, n bytes): pairs with mask 0 are left out of the plan -- an optimisation only, for point sets
// that are infinity there (a plan built without it gives the same sums)
int msm_plan_dev(const Fe* d_scalars, uint64_t n, hipStream_t s, const uint8_t* d_mask = nullptr);
// mask[i] = 0 where the point is infinity (x == 0, reference format) in every given set; *skipped_host = how many
int msm_points_mask(const Affine<Fq>* d_g1, const Affine<Fq2>* d_g2, uint64_t n, uint8_t* d_mask, uint32_t* skipped_host, hipStream_t s);
int msm_g1_exec_xyzz(const Affine<Fq>* d_points, XYZZ<Fq>* out_host, hipStream_t s, bool prepared);
int msm_g2_exec_xyzz(const Affine<Fq2>* d_points, XYZZ<Fq2>* out_host, hipStream_t s, bool prepared);
// asynchronous form of exec: launch enqueues the kernels and the copy of the window sums, finish waits
// for that copy and runs the serial host tail
int msm_g1_launch(const Affine<Fq>* d_points, bool prepared, int* slot, hipStream_t s);
int msm_g2_launch(const Affine<Fq2>* d_points, bool prepared, int* slot, hipStream_t s, hipEvent_t before_tail = nullptr);
// `before_tail` (optional) is recorded on s after the accumulations, before the batched reduction tail
// plan_ids (optional): the plan each set is accumulated against (variants of one plan: same geometry)
int msm_g1_launch_batch(const Affine<Fq>* const* d_points, int nsets, bool prepared, int* slots, hipStream_t s,
                        hipEvent_t before_tail = nullptr, const int* plan_ids = nullptr);
// two independent plans (digit/sort/task buffers) can be alive at once; plan and launches use the selected one
void msm_select_plan(int id);      // id in [0, 4)
// A second plan over the SAME scalars that leaves out the pairs with mask[i] == 0, derived from plan `src_id`
// (which must just have been built by msm_plan_dev): only the per-bin counting sort and the task list are redone,
// the digit extraction and the coarse scatter are shared.  Becomes plan `dst_id`.
int msm_plan_variant(int src_id, int dst_id, const uint8_t* d_mask, hipStream_t s);
bool msm_ready(int slot);     // the launch's window sums have reached the host (finish will not block)
int msm_g1_finish(int slot, XYZZ<Fq>* out_host);
int msm_g2_finish(int slot, XYZZ<Fq2>* out_host);
void msm_abort_pending(hipStream_t s);
int msm_prepare_points(int which, void* d_points, uint64_t n, hipStream_t s);
bool msm_uses_field29();

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
int pols_to_csr(const uint8_t* pols, size_t len, uint32_t n_signals, uint32_t domain, CsrMatrix* out,
                size_t* consumed, hipStream_t s);
// d_h_out[domain] (plain form) from a device-resident plain witness
int calc_h_dev(const Fe* d_signals_plain, uint32_t n_signals, const CsrMatrix& A, const CsrMatrix& B,
               uint32_t domain, Fe* d_h_out, hipStream_t s);
int fr_map_dev(const Fe* d_in, Fe* d_out, uint64_t n, int to_mont, hipStream_t s);

}  // namespace wsnark
