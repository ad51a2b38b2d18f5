// group.hip -- several GPUs in ONE process: a group of per-device contexts, host threads that drive them, and the transport
// between them inside the library (round 5).
//
// Replaces, for a host that is a single process (the Node.js drop-in the north star names), what the reference does with its
// worker pool: `build()` starts W workers (src/bn128.js:173-265), g1_multiexp / g2_multiexp cut the pairs into W contiguous
// ranges, post one to each worker and add the W partial results (:353-415), groth16GenProof runs the five sums that way
// (:607-622).  Here a worker is a GPU: wsnark_group_create makes one context per device (csrc/context.hip) and one host thread
// per context; a group key holds one POINTS SHARD of the key per device (wsnark_pkey_load_shard: the same contiguous split,
// 1 / N of the memory each); wsnark_group_prove runs csrc/dist.hip's one-call distributed prover on every device at once --
// partial sums over the device's shard, CALC_H on the four-step transform, one gather of the 576-byte records -- with the two
// collectives it needs implemented HERE instead of by the host:
//   all_to_all   device to device: every rank pulls its blocks out of its peers' send buffers with hipMemcpyPeerAsync on its own
//                queue, ordered by events (send buffer complete / pulls done) that the ranks record and wait for across devices;
//                two host barriers per exchange make sure an event is recorded before a peer waits for it
//   all_gather   the records are host memory: a shared array, two barriers
// Groups whose size is not a power of two, or too large for the transform's geometry (world > 2^floor(log2(domain) / 2)), fall back
// to the reference's own arrangement: CALC_H complete on every device, only the sums sharded (groth16_prove_partial + finish).
// With the multi-process hosts (one process per GPU, torch.distributed / RCCL: wasmsnark_amd/dist.py) nothing changes: they keep
// passing their own transport to wsnark_groth16_prove_dist.
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/wsnark.h"
#include "internal.h"

namespace wsnark {

const std::string& get_last_error();

// all ranks arrive, all leave: spins briefly (the exchanges of a proof are microseconds apart), then yields
struct SpinBarrier {
    std::atomic<uint32_t> count{0}, generation{0};
    uint32_t n = 1;
    void wait() {
        const uint32_t gen = generation.load(std::memory_order_acquire);
        if (count.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
            count.store(0, std::memory_order_relaxed);
            generation.fetch_add(1, std::memory_order_release);
            return;
        }
        for (unsigned spins = 0; generation.load(std::memory_order_acquire) == gen; spins++)
            if (spins > 4096) std::this_thread::yield();
    }
};

struct Group;
struct GroupRank {
    Group* g = nullptr;
    uint32_t rank = 0;
    Context* C = nullptr;
    DevBuf send, recv, witness;                   // exchange buffers of the distributed CALC_H; the rank's copy of the witness
    hipEvent_t ev_ready = nullptr, ev_done = nullptr;   // my send buffer is complete / my pulls out of the peers' buffers are done
    // the worker thread
    std::thread th;
    std::function<int(GroupRank&)> job;
    bool has_job = false, quit = false;
    int rc = 0;
    std::string err;
};
struct Group {
    uint32_t n = 0;
    std::vector<std::unique_ptr<GroupRank>> ranks;
    std::mutex mu;                                // one collective call (load, prove, sum) at a time
    std::mutex job_mu;
    std::condition_variable cv_go, cv_done;
    uint32_t pending = 0;
    SpinBarrier bar;
    std::vector<uint8_t> gather;                  // n x kGatherMax bytes
    static const size_t kGatherMax = 1024;
    bool peer_access = false;
    std::vector<struct GroupKey*> live_keys;     // keys not yet freed: wsnark_group_free frees what is left
    uint8_t last_rs[64] = {0};                    // the raw r | s of the last proof (wsnark_group_last_blinding)
    bool have_rs = false;

    // run fn on every rank's thread (its context selected), wait for all; returns the first non-zero status (its message becomes
    // this thread's last error)
    int run(const std::function<int(GroupRank&)>& fn) {
        {
            std::lock_guard<std::mutex> lk(job_mu);
            for (auto& r : ranks) { r->job = fn; r->has_job = true; r->rc = 0; r->err.clear(); }
            pending = n;
        }
        cv_go.notify_all();
        std::unique_lock<std::mutex> lk(job_mu);
        cv_done.wait(lk, [&] { return pending == 0; });
        for (auto& r : ranks)
            if (r->rc) { set_last_error("device " + std::to_string(r->C->device) + " (rank " + std::to_string(r->rank) + "): " + r->err); return r->rc; }
        return WS_OK;
    }
};

static void rank_loop(GroupRank* R) {
    Group* G = R->g;
    CtxScope scope(R->C);
    for (;;) {
        std::function<int(GroupRank&)> fn;
        {
            std::unique_lock<std::mutex> lk(G->job_mu);
            G->cv_go.wait(lk, [&] { return R->has_job || R->quit; });
            if (R->quit) return;
            fn = R->job;
            R->has_job = false;
        }
        const int rc = fn(*R);
        {
            std::lock_guard<std::mutex> lk(G->job_mu);
            R->rc = rc;
            if (rc) R->err = get_last_error();
            if (--G->pending == 0) G->cv_done.notify_all();
        }
    }
}

// ---- the transport (wsnark_comm_t's two callbacks), inside the library ----
static int group_all_to_all(void* user, uint64_t bytes_per_rank, void* stream) {
    GroupRank& R = *(GroupRank*)user;
    Group& G = *R.g;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t P = G.n, p = R.rank;
    int rc = 0;
    if (bytes_per_rank * P > R.send.bytes || bytes_per_rank * P > R.recv.bytes) rc = 1;
    if (!rc && hipEventRecord(R.ev_ready, s) != hipSuccess) rc = 1;          // my blocks are complete when this fires
    G.bar.wait();                                                              // ... and every peer's record precedes my wait for it
    for (uint32_t k = 0; k < P && !rc; k++) {
        const uint32_t q = (p + k) % P;                                        // (start with my own block: the ranks spread over the links)
        GroupRank& Q = *G.ranks[q];
        if (q != p && hipStreamWaitEvent(s, Q.ev_ready, 0) != hipSuccess) { rc = 1; break; }
        const uint8_t* src = Q.send.as<uint8_t>() + (uint64_t)p * bytes_per_rank;
        uint8_t* dst = R.recv.as<uint8_t>() + (uint64_t)q * bytes_per_rank;
        const hipError_t e = (q == p || Q.C->device == R.C->device)
                                 ? hipMemcpyAsync(dst, src, bytes_per_rank, hipMemcpyDeviceToDevice, s)
                                 : hipMemcpyPeerAsync(dst, R.C->device, src, Q.C->device, bytes_per_rank, s);
        if (e != hipSuccess) rc = 1;
    }
    if (!rc && hipEventRecord(R.ev_done, s) != hipSuccess) rc = 1;
    G.bar.wait();
    // nobody's send buffer is rewritten before its peers have pulled their blocks out of it
    for (uint32_t q = 0; q < P && !rc; q++)
        if (q != p && hipStreamWaitEvent(s, G.ranks[q]->ev_done, 0) != hipSuccess) rc = 1;
    if (rc) (void)hipGetLastError();
    return rc;
}
static int group_all_gather(void* user, const void* send, void* recv, uint64_t bytes) {
    GroupRank& R = *(GroupRank*)user;
    Group& G = *R.g;
    const bool ok = bytes <= Group::kGatherMax;
    if (ok) memcpy(G.gather.data() + (size_t)R.rank * bytes, send, bytes);
    G.bar.wait();
    if (ok) memcpy(recv, G.gather.data(), (size_t)G.n * bytes);
    G.bar.wait();                                                              // (the array is free for the next gather)
    return ok ? 0 : 1;
}

struct GroupKey {
    Group* g = nullptr;
    std::vector<ProvingKey*> keys;                // one points shard per rank
    uint32_t n_vars = 0, n_public = 0, domain = 0;
    bool dist = false;                            // CALC_H on the distributed transform (else: complete on every device)
    uint32_t h_log_m = 0;
};

int groth16_prove_dist(ProvingKey* K, const Fe* d_witness, size_t witness_len, const DistComm& cm, const uint8_t* r32, const uint8_t* s32,
                       uint8_t* out384, hipStream_t s);
int groth16_prove_partial(ProvingKey* K, const uint8_t* witness, size_t witness_len, WindowShard sh, uint8_t* out576, bool skip_h);
int groth16_prove_finish(ProvingKey* K, const uint8_t* partials, uint64_t n_ranks, const uint8_t* r32, const uint8_t* s32, uint8_t* out384);
bool last_blinding(uint8_t* r32, uint8_t* s32);
void g1_sum_host(const uint8_t* pts, uint64_t count, uint8_t* out96);
void g2_sum_host(const uint8_t* pts, uint64_t count, uint8_t* out192);

}  // namespace wsnark

using namespace wsnark;

extern "C" {

int wsnark_group_create(const int* devices, uint32_t n, wsnark_group_t** out) {
    if (!out || !devices || n == 0 || n > 64) return WSNARK_ERR_ARG;
    DeviceRestore caller_device;                  // (context_create selects every member's device in turn on this thread)
    std::unique_ptr<Group> G(new Group());
    G->n = n;
    G->bar.n = n;
    G->gather.resize((size_t)n * Group::kGatherMax);
    int rc = WS_OK;
    for (uint32_t i = 0; i < n && !rc; i++) {
        std::unique_ptr<GroupRank> R(new GroupRank());
        R->g = G.get();
        R->rank = i;
        rc = context_create(devices[i], &R->C);
        if (!rc) {
            CtxScope scope(R->C);
            if (hipEventCreateWithFlags(&R->ev_ready, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&R->ev_done, hipEventDisableTiming) != hipSuccess) { set_last_error("group: event creation failed"); rc = WS_ERR_HIP; }
        }
        G->ranks.push_back(std::move(R));
    }
#ifndef WSNARK_EMUL
    // direct device-to-device copies where the devices can reach each other (over xGMI on an MI355X node); hipMemcpyPeerAsync works
    // without it, staged by the runtime
    for (uint32_t i = 0; i < n && !rc; i++) {
        CtxScope scope(G->ranks[i]->C);
        for (uint32_t j = 0; j < n; j++) {
            const int di = G->ranks[i]->C->device, dj = G->ranks[j]->C->device;
            if (di == dj) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, di, dj) == hipSuccess && can) {
                const hipError_t e = hipDeviceEnablePeerAccess(dj, 0);
                if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) G->peer_access = true;
            }
            (void)hipGetLastError();
        }
    }
#endif
    if (rc) {
        for (auto& R : G->ranks) {
            if (!R->C) continue;
            { CtxScope scope(R->C); if (R->ev_ready) (void)hipEventDestroy(R->ev_ready); if (R->ev_done) (void)hipEventDestroy(R->ev_done); }
            context_destroy(R->C);
        }
        return rc;
    }
    for (auto& R : G->ranks) R->th = std::thread(rank_loop, R.get());
    *out = reinterpret_cast<wsnark_group_t*>(G.release());
    return WSNARK_OK;
}

static void group_key_release(GroupKey* K) {      // caller holds the group's mutex
    Group* G = K->g;
    (void)G->run([&](GroupRank& R) -> int { if (K->keys[R.rank]) pkey_free(K->keys[R.rank]); return WS_OK; });
    for (size_t i = 0; i < G->live_keys.size(); i++)
        if (G->live_keys[i] == K) { G->live_keys.erase(G->live_keys.begin() + (long)i); break; }
    delete K;
}
void wsnark_group_free(wsnark_group_t* h) {
    if (!h) return;
    DeviceRestore caller_device;
    Group* G = reinterpret_cast<Group*>(h);
    {
        std::lock_guard<std::mutex> lk(G->mu);
        while (!G->live_keys.empty()) group_key_release(G->live_keys.back());
    }
    {
        std::lock_guard<std::mutex> lk(G->job_mu);
        for (auto& R : G->ranks) R->quit = true;
    }
    G->cv_go.notify_all();
    for (auto& R : G->ranks) if (R->th.joinable()) R->th.join();
    for (auto& R : G->ranks) {
        {
            CtxScope scope(R->C);
            R->send.release(); R->recv.release(); R->witness.release();
            if (R->ev_ready) (void)hipEventDestroy(R->ev_ready);
            if (R->ev_done) (void)hipEventDestroy(R->ev_done);
        }
        context_destroy(R->C);
    }
    delete G;
}

uint32_t wsnark_group_size(const wsnark_group_t* h) { return h ? reinterpret_cast<const Group*>(h)->n : 0; }

static int group_load(Group* G, const KeySections& S, wsnark_group_pkey_t** out) {
    std::lock_guard<std::mutex> lk(G->mu);
    std::unique_ptr<GroupKey> K(new GroupKey());
    K->g = G;
    K->keys.assign(G->n, nullptr);
    K->n_vars = S.n_vars; K->n_public = S.n_public; K->domain = S.domain;
    uint32_t log_n = 0;
    while (log_n < 31 && ((uint64_t)1 << (log_n + 1)) <= S.domain) log_n++;
    const uint32_t l2 = log_n / 2;
    // the distributed transform needs a power-of-two world of at most 2^l2 ranks (a world of one takes the same path: the
    // exchanges are the identity)
    K->dist = (G->n & (G->n - 1)) == 0 && ((uint64_t)1 << l2) >= G->n && S.domain >= 4 && (S.domain & (S.domain - 1)) == 0;
    K->h_log_m = K->dist ? l2 : 0;
    GroupKey* Kp = K.get();
    int rc = G->run([&](GroupRank& R) -> int {
        int r = pkey_load_sections(S, &Kp->keys[R.rank], KeyShard{R.rank, G->n, Kp->h_log_m});
        if (r || !Kp->dist) return r;
        // exchange buffers of the distributed CALC_H: three vectors of domain / world elements (include/wsnark.h: wsnark_comm_t)
        const size_t nbytes = (size_t)3 * (S.domain / G->n) * 32;
        if (R.send.reserve(nbytes) != hipSuccess || R.recv.reserve(nbytes) != hipSuccess) { set_last_error("group key: exchange buffers"); return WS_ERR_HIP; }
        return WS_OK;
    });
    if (rc) {
        const std::string msg = get_last_error();
        (void)G->run([&](GroupRank& R) -> int { if (Kp->keys[R.rank]) { pkey_free(Kp->keys[R.rank]); Kp->keys[R.rank] = nullptr; } return WS_OK; });
        set_last_error(msg);
        return rc;
    }
    G->live_keys.push_back(K.get());
    *out = reinterpret_cast<wsnark_group_pkey_t*>(K.release());
    return WSNARK_OK;
}

int wsnark_group_pkey_load_sections(wsnark_group_t* g, const wsnark_key_sections_t* ks, wsnark_group_pkey_t** out) {
    if (!g || !out || !ks || !ks->alfa1 || !ks->beta1 || !ks->delta1 || !ks->beta2 || !ks->delta2 || !ks->polsA ||
        !ks->polsB || !ks->pointsA || !ks->pointsB1 || !ks->pointsB2 || !ks->pointsH ||
        (!ks->pointsC && (uint64_t)ks->n_vars > (uint64_t)ks->n_public + 1))
        return WSNARK_ERR_ARG;
    KeySections S{ks->n_vars, ks->n_public, ks->domain, (const uint8_t*)ks->alfa1, (const uint8_t*)ks->beta1,
                  (const uint8_t*)ks->delta1, (const uint8_t*)ks->beta2, (const uint8_t*)ks->delta2,
                  (const uint8_t*)ks->polsA, ks->polsA_len, (const uint8_t*)ks->polsB, ks->polsB_len,
                  (const uint8_t*)ks->pointsA, (const uint8_t*)ks->pointsB1, (const uint8_t*)ks->pointsB2,
                  (const uint8_t*)ks->pointsC, (const uint8_t*)ks->pointsH,
                  ks->pointsA_len, ks->pointsB1_len, ks->pointsB2_len, ks->pointsC_len, ks->pointsH_len};
    return group_load(reinterpret_cast<Group*>(g), S, out);
}
int wsnark_group_pkey_load(wsnark_group_t* g, const void* pkey, size_t len, wsnark_group_pkey_t** out) {
    if (!g || !out) return WSNARK_ERR_ARG;
    KeySections S;
    int rc = pkey_parse((const uint8_t*)pkey, len, &S);
    if (rc) return rc;
    return group_load(reinterpret_cast<Group*>(g), S, out);
}
int wsnark_group_pkey_load_file(wsnark_group_t* g, const char* path, wsnark_group_pkey_t** out) {
    if (!g || !out || !path) return WSNARK_ERR_ARG;
    KeyFile F;                // ONE mapping for all members: each reads its own slices of it (and the two matrices)
    KeySections S;
    int rc = keyfile_open(path, &F, &S);
    if (rc) return rc;
    return group_load(reinterpret_cast<Group*>(g), S, out);
}
void wsnark_group_pkey_free(wsnark_group_pkey_t* h) {
    if (!h) return;
    GroupKey* K = reinterpret_cast<GroupKey*>(h);
    std::lock_guard<std::mutex> lk(K->g->mu);
    group_key_release(K);
}
int wsnark_group_pkey_info(const wsnark_group_pkey_t* h, uint32_t* n_vars, uint32_t* n_public, uint32_t* domain, uint32_t* world, int* distributed_calc_h) {
    if (!h) return WSNARK_ERR_ARG;
    const GroupKey* K = reinterpret_cast<const GroupKey*>(h);
    if (n_vars) *n_vars = K->n_vars;
    if (n_public) *n_public = K->n_public;
    if (domain) *domain = K->domain;
    if (world) *world = K->g->n;
    if (distributed_calc_h) *distributed_calc_h = K->dist ? 1 : 0;
    return WSNARK_OK;
}
int wsnark_group_pkey_wait_tables(wsnark_group_pkey_t* h) {
    if (!h) return WSNARK_ERR_ARG;
    GroupKey* K = reinterpret_cast<GroupKey*>(h);
    std::lock_guard<std::mutex> lk(K->g->mu);
    return K->g->run([&](GroupRank& R) -> int { return pkey_wait_tables(K->keys[R.rank]); });
}

// One proof on all devices of the group: the reference's groth16GenProof (src/bn128.js:580-720) with GPUs as its workers.
int wsnark_group_prove(wsnark_group_pkey_t* h, const void* witness, size_t witness_len, const void* r32, const void* s32, void* out384) {
    if (!h || !witness || !out384) return WSNARK_ERR_ARG;
    GroupKey* K = reinterpret_cast<GroupKey*>(h);
    Group* G = K->g;
    if ((uint64_t)witness_len < (uint64_t)K->n_vars * 32) { set_last_error("witness shorter than nVars*32 bytes"); return WSNARK_ERR_SIZE; }
    std::lock_guard<std::mutex> lk(G->mu);
    const uint32_t P = G->n;
    if (K->dist) {
        std::vector<uint8_t> proofs((size_t)P * 384);
        int rc = G->run([&](GroupRank& R) -> int {
            Context* C = R.C;
            int st = WS_OK;
            // every device needs the whole witness (its sparse products read any signal; its sums their slice).  A failure here must
            // not keep this rank out of the collectives its peers are about to post: the distributed prover is called regardless and
            // agrees on the error through its first gather (a too-short length makes its preflight fail on this rank).
            if (R.witness.reserve((size_t)K->n_vars * 32) != hipSuccess) { (void)hipGetLastError(); st = WS_ERR_HIP; }
            if (!st) st = upload_staged(R.witness.p, witness, (size_t)K->n_vars * 32, C->stream);
            DistComm cm;
            cm.rank = R.rank; cm.world = P;
            cm.d_send = R.send.as<Fe>(); cm.d_recv = R.recv.as<Fe>(); cm.buf_bytes = R.send.bytes < R.recv.bytes ? R.send.bytes : R.recv.bytes;
            cm.all_to_all = group_all_to_all; cm.all_gather = group_all_gather; cm.user = &R;
            const int rc2 = groth16_prove_dist(K->keys[R.rank], R.witness.as<Fe>(), st ? 0 : witness_len, cm, (const uint8_t*)r32, (const uint8_t*)s32,
                                               &proofs[(size_t)R.rank * 384], C->stream);
            if (!st && !rc2 && R.rank == 0) G->have_rs = last_blinding(G->last_rs, G->last_rs + 32);
            return st ? st : rc2;
        });
        if (rc) return rc;
        memcpy(out384, proofs.data(), 384);       // (every rank assembles the same proof from the gathered records)
        return WSNARK_OK;
    }
    // CALC_H complete on every device, the five sums over the device's points shard; records combined here
    std::vector<uint8_t> recs((size_t)P * 576);
    int rc = G->run([&](GroupRank& R) -> int {
        return groth16_prove_partial(K->keys[R.rank], (const uint8_t*)witness, witness_len, WindowShard{R.rank, P}, &recs[(size_t)R.rank * 576], false);
    });
    if (rc) return rc;
    CtxScope scope(G->ranks[0]->C);
    rc = groth16_prove_finish(K->keys[0], recs.data(), P, (const uint8_t*)r32, (const uint8_t*)s32, (uint8_t*)out384);
    if (!rc) G->have_rs = last_blinding(G->last_rs, G->last_rs + 32);
    return rc;
}
/* the raw 32-byte r | s of the group's last proof (drawn by rank 0 when none were given): the reference keeps them for its tests in
 * `_pr`, `_ps` (src/bn128.js:662-664) */
int wsnark_group_last_blinding(wsnark_group_t* g, void* r32, void* s32) {
    if (!g) return WSNARK_ERR_ARG;
    Group* G = reinterpret_cast<Group*>(g);
    std::lock_guard<std::mutex> lk(G->mu);
    if (!G->have_rs) return WSNARK_ERR_ARG;
    if (r32) memcpy(r32, G->last_rs, 32);
    if (s32) memcpy(s32, G->last_rs + 32, 32);
    return WSNARK_OK;
}

// G1_MULTIEXP / G2_MULTIEXP over the group: the reference's split (src/bn128.js:353-415) -- contiguous ranges of the pairs, the
// remainder to the last worker, the partial results added on the host
static int group_msm(Group* G, int which, const void* scalars, const void* points, uint64_t n, void* out) {
    if (!out || (n && (!scalars || !points))) return WSNARK_ERR_ARG;
    if (n > ((uint64_t)1 << 28)) return WSNARK_ERR_SIZE;
    std::lock_guard<std::mutex> lk(G->mu);
    const uint32_t P = G->n;
    const size_t psz = which ? 128 : 64, osz = which ? 192 : 96;
    std::vector<uint8_t> parts((size_t)P * osz);
    const uint64_t per = n / P;
    int rc = G->run([&](GroupRank& R) -> int {
        const uint64_t lo = (uint64_t)R.rank * per, cnt = R.rank == P - 1 ? n - lo : per;
        LaneLock L = acquire_lane(R.C);
        const uint8_t* sc = (const uint8_t*)scalars + lo * 32;
        const uint8_t* pt = (const uint8_t*)points + lo * psz;
        if (which) { Jac<Fq2> r; int e = msm_g2_host(*L, sc, pt, cnt, WindowShard{}, &r); if (e) return e; memcpy(&parts[(size_t)R.rank * osz], &r, sizeof r); }
        else { Jac<Fq> r; int e = msm_g1_host(*L, sc, pt, cnt, WindowShard{}, &r); if (e) return e; memcpy(&parts[(size_t)R.rank * osz], &r, sizeof r); }
        return WS_OK;
    });
    if (rc) return rc;
    if (which) g2_sum_host(parts.data(), P, (uint8_t*)out);
    else g1_sum_host(parts.data(), P, (uint8_t*)out);
    return WSNARK_OK;
}
int wsnark_group_g1_msm(wsnark_group_t* g, const void* scalars, const void* points, uint64_t n, void* out96) {
    if (!g) return WSNARK_ERR_ARG;
    return group_msm(reinterpret_cast<Group*>(g), 0, scalars, points, n, out96);
}
int wsnark_group_g2_msm(wsnark_group_t* g, const void* scalars, const void* points, uint64_t n, void* out192) {
    if (!g) return WSNARK_ERR_ARG;
    return group_msm(reinterpret_cast<Group*>(g), 1, scalars, points, n, out192);
}

}  // extern "C"
