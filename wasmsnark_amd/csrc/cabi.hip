// cabi.hip -- the extern "C" surface declared in include/wsnark.h.
#include <limits.h>
#include <string.h>

#include "../../include/wsnark.h"
#include "internal.h"

namespace wsnark {
int context_init(int device);
void context_shutdown();
const std::string& get_last_error();
const std::string& device_info();
struct ProvingKey;
int pkey_load(const uint8_t* buf, size_t len, ProvingKey** out);
void pkey_free(ProvingKey* K);
void pkey_info(const ProvingKey* K, uint32_t* nv, uint32_t* np, uint32_t* dom);
void pkey_table_info(const ProvingKey* K, uint32_t* cw, uint32_t* rw, uint32_t* ch, uint32_t* rh, uint64_t* bytes);
int groth16_prove_host_witness(ProvingKey* K, const uint8_t* witness, size_t witness_len, const uint8_t* r32,
                               const uint8_t* s32, uint8_t* out384);
int pkey_load_sections(const KeySections& S, ProvingKey** out, KeyShard shard);
void pkey_shard_info(const ProvingKey* K, uint32_t* rank, uint32_t* world, uint64_t* lo, uint64_t* n_local, uint64_t* h_local, uint32_t* h_log_m);
void pkey_load_stats(const ProvingKey* K, double* out5);
int pkey_wait_tables(ProvingKey* K);
Context* pkey_context(const ProvingKey* K);
int groth16_prove_dist(ProvingKey* K, const Fe* d_witness, size_t witness_len, const DistComm& cm, const uint8_t* r32, const uint8_t* s32,
                       uint8_t* out384, hipStream_t s);
int pkey_h_msm_dev(ProvingKey* K, const Fe* d_h_local, uint64_t n, uint8_t* out96, hipStream_t s);
int groth16_prove_partial(ProvingKey* K, const uint8_t* witness, size_t witness_len, WindowShard sh, uint8_t* out576, bool skip_h);
int groth16_prove_partial_dev(ProvingKey* K, const Fe* d_witness, size_t witness_len, WindowShard sh, uint8_t* out576, hipStream_t s, bool skip_h);
int pkey_eval_ab_dev(ProvingKey* K, const Fe* d_witness, size_t witness_len, Fe* d_a, Fe* d_b, hipStream_t s);
int groth16_prove_finish(ProvingKey* K, const uint8_t* partials, uint64_t n_ranks, const uint8_t* r32,
                         const uint8_t* s32, uint8_t* out384);
int groth16_prove_dev_witness(ProvingKey* K, const Fe* d_witness, size_t witness_len, const uint8_t* r32,
                              const uint8_t* s32, uint8_t* out384, hipStream_t s);
bool last_blinding(uint8_t* r32, uint8_t* s32);
int groth16_verify(const uint8_t* vk, size_t vk_len, const uint8_t* inputs, uint64_t n_inputs, const uint8_t* proof384, int* valid);
int dist_scale_dev(Fe* d_data, uint64_t stack, uint64_t rows, uint64_t cols, uint64_t row0, uint32_t log_n1, uint32_t log_n, int mode, int inverse, hipStream_t s);
void g1_sum_host(const uint8_t* pts, uint64_t count, uint8_t* out96);
void g2_sum_host(const uint8_t* pts, uint64_t count, uint8_t* out192);
int g1_mul_base_batch(const void* base, const void* scalars, uint64_t n, void* out);
int g2_mul_base_batch(const void* base, const void* scalars, uint64_t n, void* out);
int peak_probe(int probe, double* gops);
struct ResidentPoints;
Context* points_context(const ResidentPoints* H);
int points_load(int which, const void* h_points, uint64_t n, ResidentPoints** out);
void points_free(ResidentPoints* H);
void points_info(const ResidentPoints* H, int* which, uint64_t* n, uint32_t* table_c, uint32_t* rows, uint64_t* bytes);
int points_msm(ResidentPoints* H, const void* scalars, bool on_device, uint64_t n, void* out, hipStream_t s);
int selftest_field(int which, int impl, int op, const uint8_t* a, const uint8_t* b, uint8_t* out, uint64_t n);
int selftest_curve(int g, int impl, int op, const uint8_t* p, const uint8_t* q, uint8_t* out, uint64_t n);
}  // namespace wsnark

using namespace wsnark;

static_assert(sizeof(Fe) == 32, "field element must be 32 bytes");
static_assert(sizeof(Affine<Fq>) == 64 && sizeof(Affine<Fq2>) == 128, "affine layouts");
static_assert(sizeof(Jac<Fq>) == 96 && sizeof(Jac<Fq2>) == 192, "jacobian layouts");
static_assert(sizeof(XYZZ<Fq>) == 128 && sizeof(XYZZ<Fq2>) == 256, "xyzz layouts");

// HIP's current device is per host thread (default 0) and every entry point may be called from any thread (the Node
// addon runs on the libuv pool): select the context's device first.
#define REQUIRE_CTX()                                                        \
    Context* C = ctx();                                                      \
    if (!C) { set_last_error("wsnark_init() has not been called"); return WSNARK_ERR_NOINIT; } \
    WS_HIP_CHECK(hipSetDevice(C->device))
// entry points that take a key handle run on the context (device) the key is resident on, whichever thread calls
#define REQUIRE_KEY_CTX(h)                                                   \
    if (!(h)) return WSNARK_ERR_ARG;                                         \
    Context* C = pkey_context(reinterpret_cast<const ProvingKey*>(h));       \
    if (!C) { set_last_error("wsnark_init() has not been called"); return WSNARK_ERR_NOINIT; } \
    CtxScope _key_scope(C)
static bool shard_ok(uint32_t rank, uint32_t world) { return world != 0 && rank < world; }

template <class JacT, class Fn>
static int msm_entry(Context* C, bool host, const void* scalars, const void* points, uint64_t n, uint32_t rank, uint32_t world,
                     void* out, Fn run) {
    if (!out || (n && (!scalars || !points))) return WSNARK_ERR_ARG;
    if (!shard_ok(rank, world)) return WSNARK_ERR_ARG;
    if (n > ((uint64_t)1 << 28)) return WSNARK_ERR_SIZE;
    (void)host;
    JacT r;
    LaneLock L = acquire_lane(C);
    int rc = run(*L, WindowShard{rank, world}, &r);
    if (rc) return rc;
    memcpy(out, &r, sizeof r);
    return WSNARK_OK;
}

extern "C" {

int wsnark_init(int device) { return context_init(device); }
void wsnark_shutdown(void) { context_shutdown(); }
const char* wsnark_last_error(void) { return get_last_error().c_str(); }
const char* wsnark_device_info(void) { return device_info().c_str(); }

// ---- MSM ----
int wsnark_g1_msm_windows(const void* scalars, const void* points, uint64_t n, uint32_t rank, uint32_t world, void* out96) {
    REQUIRE_CTX();
    return msm_entry<Jac<Fq>>(C, true, scalars, points, n, rank, world, out96,
                              [&](Lane& L, WindowShard sh, Jac<Fq>* r) { return msm_g1_host(L, scalars, points, n, sh, r); });
}
int wsnark_g2_msm_windows(const void* scalars, const void* points, uint64_t n, uint32_t rank, uint32_t world, void* out192) {
    REQUIRE_CTX();
    return msm_entry<Jac<Fq2>>(C, true, scalars, points, n, rank, world, out192,
                               [&](Lane& L, WindowShard sh, Jac<Fq2>* r) { return msm_g2_host(L, scalars, points, n, sh, r); });
}
int wsnark_g1_msm_windows_dev(const void* d_scalars, const void* d_points, uint64_t n, uint32_t rank, uint32_t world,
                              void* out96_host, void* stream) {
    REQUIRE_CTX();
    return msm_entry<Jac<Fq>>(C, false, d_scalars, d_points, n, rank, world, out96_host, [&](Lane& L, WindowShard sh, Jac<Fq>* r) {
        return msm_g1_dev(L, (const Fe*)d_scalars, (const Affine<Fq>*)d_points, n, sh, r, (hipStream_t)stream);
    });
}
int wsnark_g2_msm_windows_dev(const void* d_scalars, const void* d_points, uint64_t n, uint32_t rank, uint32_t world,
                              void* out192_host, void* stream) {
    REQUIRE_CTX();
    return msm_entry<Jac<Fq2>>(C, false, d_scalars, d_points, n, rank, world, out192_host, [&](Lane& L, WindowShard sh, Jac<Fq2>* r) {
        return msm_g2_dev(L, (const Fe*)d_scalars, (const Affine<Fq2>*)d_points, n, sh, r, (hipStream_t)stream);
    });
}
int wsnark_g1_msm(const void* scalars, const void* points, uint64_t n, void* out96) {
    return wsnark_g1_msm_windows(scalars, points, n, 0, 1, out96);
}
int wsnark_g2_msm(const void* scalars, const void* points, uint64_t n, void* out192) {
    return wsnark_g2_msm_windows(scalars, points, n, 0, 1, out192);
}
int wsnark_g1_msm_dev(const void* d_scalars, const void* d_points, uint64_t n, void* out96_host, void* stream) {
    return wsnark_g1_msm_windows_dev(d_scalars, d_points, n, 0, 1, out96_host, stream);
}
int wsnark_g2_msm_dev(const void* d_scalars, const void* d_points, uint64_t n, void* out192_host, void* stream) {
    return wsnark_g2_msm_windows_dev(d_scalars, d_points, n, 0, 1, out192_host, stream);
}

int wsnark_g1_sum(const void* jac_points, uint64_t count, void* out96) {
    if (!out96 || (count && !jac_points)) return WSNARK_ERR_ARG;
    g1_sum_host((const uint8_t*)jac_points, count, (uint8_t*)out96);
    return WSNARK_OK;
}
int wsnark_g2_sum(const void* jac_points, uint64_t count, void* out192) {
    if (!out192 || (count && !jac_points)) return WSNARK_ERR_ARG;
    g2_sum_host((const uint8_t*)jac_points, count, (uint8_t*)out192);
    return WSNARK_OK;
}

// ---- resident bases: a point set kept on the device as fixed-base window tables (fixedbase.hip) ----
int wsnark_points_load(int group, const void* points, uint64_t n, wsnark_points_t** out_handle) {
    REQUIRE_CTX();
    if (group != 1 && group != 2) return WSNARK_ERR_ARG;
    ResidentPoints* H = nullptr;
    int rc = points_load(group - 1, points, n, &H);
    if (rc) return rc;
    *out_handle = reinterpret_cast<wsnark_points_t*>(H);
    return WSNARK_OK;
}
void wsnark_points_free(wsnark_points_t* h) {
    if (!h) return;
    CtxScope scope(points_context(reinterpret_cast<const ResidentPoints*>(h)));
    points_free(reinterpret_cast<ResidentPoints*>(h));
}
int wsnark_points_info(const wsnark_points_t* h, int* group, uint64_t* n, uint32_t* window_bits, uint32_t* rows, uint64_t* table_bytes) {
    if (!h) return WSNARK_ERR_ARG;
    int which = 0;
    points_info(reinterpret_cast<const ResidentPoints*>(h), &which, n, window_bits, rows, table_bytes);
    if (group) *group = which + 1;
    return WSNARK_OK;
}
int wsnark_points_msm(wsnark_points_t* h, const void* scalars, uint64_t n, void* out) {
    if (!h) return WSNARK_ERR_ARG;
    Context* C = points_context(reinterpret_cast<const ResidentPoints*>(h));
    if (!C) return WSNARK_ERR_NOINIT;
    CtxScope scope(C);
    return points_msm(reinterpret_cast<ResidentPoints*>(h), scalars, false, n, out, nullptr);
}
int wsnark_points_msm_dev(wsnark_points_t* h, const void* d_scalars, uint64_t n, void* out_host, void* stream) {
    if (!h) return WSNARK_ERR_ARG;
    Context* C = points_context(reinterpret_cast<const ResidentPoints*>(h));
    if (!C) return WSNARK_ERR_NOINIT;
    CtxScope scope(C);
    return points_msm(reinterpret_cast<ResidentPoints*>(h), d_scalars, true, n, out_host, (hipStream_t)stream);
}

// ---- NTT ----
int wsnark_fr_ntt_dev(void* d_buf, uint64_t n, int odd, int inverse, void* stream) {
    REQUIRE_CTX();
    LaneLock L = acquire_lane(C);
    return ntt_dev(*L, (Fe*)d_buf, n, odd, inverse, (hipStream_t)stream);
}
int wsnark_fr_ntt_batch_dev(void* d_buf, uint64_t n, uint64_t count, int inverse, void* stream) {
    REQUIRE_CTX();
    LaneLock L = acquire_lane(C);
    return ntt_dev(*L, (Fe*)d_buf, n, 0, inverse, (hipStream_t)stream, count);
}
int wsnark_fr_dist_scale_dev(void* d_buf, uint64_t stack, uint64_t rows, uint64_t cols, uint64_t row0, uint32_t log_n1, uint32_t log_n, int mode,
                             int inverse, void* stream) {
    REQUIRE_CTX();
    return dist_scale_dev((Fe*)d_buf, stack, rows, cols, row0, log_n1, log_n, mode, inverse, (hipStream_t)stream);
}
int wsnark_fr_ntt(void* buf, uint64_t n, int odd, int inverse) {
    REQUIRE_CTX();
    if (!buf) return WSNARK_ERR_ARG;
    if (n == 0 || (n & (n - 1)) || n > ((uint64_t)1 << 28)) return WSNARK_ERR_SIZE;
    DevBuf d;
    WS_HIP_CHECK(d.alloc(n * 32));
    LaneLock L = acquire_lane(C);
    WS_HIP_CHECK(hipMemcpyAsync(d.p, buf, n * 32, hipMemcpyHostToDevice, L->stream));
    int rc = ntt_dev(*L, d.as<Fe>(), n, odd, inverse, L->stream);
    if (rc) return rc;
    WS_HIP_CHECK(hipMemcpyAsync(buf, d.p, n * 32, hipMemcpyDeviceToHost, L->stream));
    WS_HIP_CHECK(hipStreamSynchronize(L->stream));
    return WSNARK_OK;
}
static int fr_map_host(const void* in, void* out, uint64_t n, int to_mont) {
    REQUIRE_CTX();
    if (n == 0) return WSNARK_OK;
    if (!in || !out) return WSNARK_ERR_ARG;
    DevBuf d;
    WS_HIP_CHECK(d.alloc(n * 32));
    int rc = upload_staged(d.p, in, n * 32, C->stream);
    if (!rc) rc = fr_map_dev(d.as<Fe>(), d.as<Fe>(), n, to_mont, C->stream);
    if (rc) return rc;
    WS_HIP_CHECK(hipMemcpyAsync(out, d.p, n * 32, hipMemcpyDeviceToHost, C->stream));
    WS_HIP_CHECK(hipStreamSynchronize(C->stream));
    return WSNARK_OK;
}
int wsnark_fr_to_montgomery(const void* in, void* out, uint64_t n) { return fr_map_host(in, out, n, 1); }
int wsnark_fr_from_montgomery(const void* in, void* out, uint64_t n) { return fr_map_host(in, out, n, 0); }

// ---- CALC_H ----
int wsnark_calc_h(const void* signals, const void* polsA, size_t lenA, const void* polsB, size_t lenB,
                  uint32_t n_signals, uint32_t domain, void* out_h) {
    REQUIRE_CTX();
    if (!signals || !polsA || !polsB || !out_h) return WSNARK_ERR_ARG;
    if (domain < 2 || (domain & (domain - 1)) || domain > (1u << 27)) return WSNARK_ERR_SIZE;
    CsrMatrix A, B;
    size_t used = 0;
    int rc = pols_to_csr((const uint8_t*)polsA, lenA, n_signals, domain, &A, &used, C->stream);
    if (rc) return rc;
    rc = pols_to_csr((const uint8_t*)polsB, lenB, n_signals, domain, &B, &used, C->stream);
    if (rc) return rc;
    DevBuf dsig, dh;
    WS_HIP_CHECK(dsig.alloc((size_t)n_signals * 32));
    WS_HIP_CHECK(dh.alloc((size_t)domain * 32));
    LaneLock L = acquire_lane(C);
    hipStream_t s = L->stream;
    WS_HIP_CHECK(hipMemcpyAsync(dsig.p, signals, (size_t)n_signals * 32, hipMemcpyHostToDevice, s));
    rc = calc_h_dev(*L, dsig.as<Fe>(), n_signals, A, B, domain, dh.as<Fe>(), s);
    if (rc) return rc;
    WS_HIP_CHECK(hipMemcpyAsync(out_h, dh.p, (size_t)domain * 32, hipMemcpyDeviceToHost, s));
    WS_HIP_CHECK(hipStreamSynchronize(s));
    return WSNARK_OK;
}

// ---- proving key / prove ----
int wsnark_pkey_load(const void* pkey, size_t len, wsnark_pkey_t** out_handle) {
    REQUIRE_CTX();
    if (!out_handle) return WSNARK_ERR_ARG;
    ProvingKey* K = nullptr;
    int rc = pkey_load((const uint8_t*)pkey, len, &K);
    if (rc) return rc;
    *out_handle = reinterpret_cast<wsnark_pkey_t*>(K);
    return WSNARK_OK;
}
void wsnark_pkey_free(wsnark_pkey_t* h) {
    if (!h) return;
    CtxScope scope(pkey_context(reinterpret_cast<const ProvingKey*>(h)));
    pkey_free(reinterpret_cast<ProvingKey*>(h));
}
int wsnark_pkey_info(const wsnark_pkey_t* h, uint32_t* nv, uint32_t* np, uint32_t* dom) {
    if (!h) return WSNARK_ERR_ARG;
    pkey_info(reinterpret_cast<const ProvingKey*>(h), nv, np, dom);
    return WSNARK_OK;
}
int wsnark_pkey_table_info(const wsnark_pkey_t* h, uint32_t* cw, uint32_t* rw, uint32_t* ch, uint32_t* rh, uint64_t* bytes) {
    if (!h) return WSNARK_ERR_ARG;
    pkey_table_info(reinterpret_cast<const ProvingKey*>(h), cw, rw, ch, rh, bytes);
    return WSNARK_OK;
}
int wsnark_groth16_prove(wsnark_pkey_t* h, const void* witness, size_t witness_len, const void* r32, const void* s32,
                         void* out384) {
    REQUIRE_KEY_CTX(h);
    if (!h || !witness || !out384) return WSNARK_ERR_ARG;
    return groth16_prove_host_witness(reinterpret_cast<ProvingKey*>(h), (const uint8_t*)witness, witness_len,
                                      (const uint8_t*)r32, (const uint8_t*)s32, (uint8_t*)out384);
}
int wsnark_groth16_prove_dev(wsnark_pkey_t* h, const void* d_witness, size_t witness_len, const void* r32,
                             const void* s32, void* out384_host, void* stream) {
    REQUIRE_KEY_CTX(h);
    if (!h || !d_witness || !out384_host) return WSNARK_ERR_ARG;
    return groth16_prove_dev_witness(reinterpret_cast<ProvingKey*>(h), (const Fe*)d_witness, witness_len,
                                     (const uint8_t*)r32, (const uint8_t*)s32, (uint8_t*)out384_host, (hipStream_t)stream);
}

static int load_sections_entry(const wsnark_key_sections_t* ks, KeyShard shard, wsnark_pkey_t** out_handle) {
    if (!ks || !out_handle || !ks->alfa1 || !ks->beta1 || !ks->delta1 || !ks->beta2 || !ks->delta2 || !ks->polsA ||
        !ks->polsB || !ks->pointsA || !ks->pointsB1 || !ks->pointsB2 || !ks->pointsH ||
        (!ks->pointsC && (uint64_t)ks->n_vars > (uint64_t)ks->n_public + 1))
        return WSNARK_ERR_ARG;
    KeySections S{ks->n_vars, ks->n_public, ks->domain, (const uint8_t*)ks->alfa1, (const uint8_t*)ks->beta1,
                  (const uint8_t*)ks->delta1, (const uint8_t*)ks->beta2, (const uint8_t*)ks->delta2,
                  (const uint8_t*)ks->polsA, ks->polsA_len, (const uint8_t*)ks->polsB, ks->polsB_len,
                  (const uint8_t*)ks->pointsA, (const uint8_t*)ks->pointsB1, (const uint8_t*)ks->pointsB2,
                  (const uint8_t*)ks->pointsC, (const uint8_t*)ks->pointsH,
                  ks->pointsA_len, ks->pointsB1_len, ks->pointsB2_len, ks->pointsC_len, ks->pointsH_len};
    ProvingKey* K = nullptr;
    int rc = pkey_load_sections(S, &K, shard);
    if (rc) return rc;
    *out_handle = reinterpret_cast<wsnark_pkey_t*>(K);
    return WSNARK_OK;
}
int wsnark_pkey_load_sections(const wsnark_key_sections_t* ks, wsnark_pkey_t** out_handle) {
    REQUIRE_CTX();
    return load_sections_entry(ks, KeyShard{}, out_handle);
}
int wsnark_pkey_load_shard(const wsnark_key_sections_t* ks, uint32_t rank, uint32_t world, uint32_t h_interleave_log,
                           wsnark_pkey_t** out_handle) {
    REQUIRE_CTX();
    if (!shard_ok(rank, world)) return WSNARK_ERR_ARG;
    return load_sections_entry(ks, KeyShard{rank, world, h_interleave_log}, out_handle);
}
// a key FILE: proving_key.bin or the WSNARK64 container (keyfile.hip); rank / world / h_interleave_log as wsnark_pkey_load_shard
int wsnark_pkey_load_file(const char* path, uint32_t rank, uint32_t world, uint32_t h_interleave_log, wsnark_pkey_t** out_handle) {
    REQUIRE_CTX();
    if (!path || !out_handle || !shard_ok(rank, world)) return WSNARK_ERR_ARG;
    KeyFile F;
    KeySections S;
    int rc = keyfile_open(path, &F, &S);
    if (rc) return rc;
    ProvingKey* K = nullptr;
    rc = pkey_load_sections(S, &K, KeyShard{rank, world, h_interleave_log});      // (returns with every byte it needs copied: F may go)
    if (rc) return rc;
    *out_handle = reinterpret_cast<wsnark_pkey_t*>(K);
    return WSNARK_OK;
}
int wsnark_pkey_file_info(const char* path, uint32_t* n_vars, uint32_t* n_public, uint32_t* domain, uint64_t* file_bytes, int* format) {
    if (!path) return WSNARK_ERR_ARG;
    KeyFile F;
    KeySections S;
    int rc = keyfile_open(path, &F, &S);
    if (rc) return rc;
    if (n_vars) *n_vars = S.n_vars;
    if (n_public) *n_public = S.n_public;
    if (domain) *domain = S.domain;
    if (file_bytes) *file_bytes = F.len;
    if (format) *format = F.format;
    return WSNARK_OK;
}
int wsnark_pkey_shard_info(const wsnark_pkey_t* h, uint32_t* rank, uint32_t* world, uint64_t* first_signal, uint64_t* n_signals,
                           uint64_t* n_hexps, uint32_t* h_interleave_log) {
    if (!h) return WSNARK_ERR_ARG;
    pkey_shard_info(reinterpret_cast<const ProvingKey*>(h), rank, world, first_signal, n_signals, n_hexps, h_interleave_log);
    return WSNARK_OK;
}
int wsnark_pkey_load_stats(const wsnark_pkey_t* h, double* ms5) {
    if (!h || !ms5) return WSNARK_ERR_ARG;
    pkey_load_stats(reinterpret_cast<const ProvingKey*>(h), ms5);
    return WSNARK_OK;
}
int wsnark_pkey_wait_tables(wsnark_pkey_t* h) {
    REQUIRE_KEY_CTX(h);
    if (!h) return WSNARK_ERR_ARG;
    return pkey_wait_tables(reinterpret_cast<ProvingKey*>(h));
}
int wsnark_pkey_h_msm_dev(wsnark_pkey_t* h, const void* d_h_slice, uint64_t n, void* out96_host, void* stream) {
    REQUIRE_KEY_CTX(h);
    if (!h || !out96_host || (n && !d_h_slice)) return WSNARK_ERR_ARG;
    return pkey_h_msm_dev(reinterpret_cast<ProvingKey*>(h), (const Fe*)d_h_slice, n, (uint8_t*)out96_host, (hipStream_t)stream);
}
int wsnark_groth16_prove_partial(wsnark_pkey_t* h, const void* witness, size_t witness_len, uint32_t rank, uint32_t world,
                                 uint32_t flags, void* out576) {
    REQUIRE_KEY_CTX(h);
    if (!h || !witness || !out576 || !shard_ok(rank, world) || (flags & ~(uint32_t)WSNARK_PARTIAL_SKIP_H)) return WSNARK_ERR_ARG;
    return groth16_prove_partial(reinterpret_cast<ProvingKey*>(h), (const uint8_t*)witness, witness_len, WindowShard{rank, world},
                                 (uint8_t*)out576, (flags & WSNARK_PARTIAL_SKIP_H) != 0);
}
int wsnark_groth16_prove_partial_dev(wsnark_pkey_t* h, const void* d_witness, size_t witness_len, uint32_t rank, uint32_t world,
                                     uint32_t flags, void* out576_host, void* stream) {
    REQUIRE_KEY_CTX(h);
    if (!h || !d_witness || !out576_host || !shard_ok(rank, world) || (flags & ~(uint32_t)WSNARK_PARTIAL_SKIP_H)) return WSNARK_ERR_ARG;
    return groth16_prove_partial_dev(reinterpret_cast<ProvingKey*>(h), (const Fe*)d_witness, witness_len, WindowShard{rank, world},
                                     (uint8_t*)out576_host, (hipStream_t)stream, (flags & WSNARK_PARTIAL_SKIP_H) != 0);
}
int wsnark_groth16_prove_dist(wsnark_pkey_t* h, const void* d_witness, size_t witness_len, const wsnark_comm_t* comm, const void* r32,
                              const void* s32, void* out384_host, void* stream) {
    REQUIRE_KEY_CTX(h);
    if (!h || !d_witness || !comm || !out384_host) return WSNARK_ERR_ARG;
    DistComm cm;
    cm.rank = comm->rank; cm.world = comm->world;
    cm.d_send = (Fe*)comm->d_send; cm.d_recv = (Fe*)comm->d_recv; cm.buf_bytes = comm->buf_bytes;
    cm.all_to_all = comm->all_to_all; cm.all_gather = comm->all_gather; cm.user = comm->user;
    return groth16_prove_dist(reinterpret_cast<ProvingKey*>(h), (const Fe*)d_witness, witness_len, cm, (const uint8_t*)r32, (const uint8_t*)s32,
                              (uint8_t*)out384_host, (hipStream_t)stream);
}
int wsnark_pkey_eval_ab_dev(wsnark_pkey_t* h, const void* d_witness, size_t witness_len, void* d_a_out, void* d_b_out, void* stream) {
    REQUIRE_KEY_CTX(h);
    if (!h || !d_witness || !d_a_out || !d_b_out) return WSNARK_ERR_ARG;
    return pkey_eval_ab_dev(reinterpret_cast<ProvingKey*>(h), (const Fe*)d_witness, witness_len, (Fe*)d_a_out, (Fe*)d_b_out, (hipStream_t)stream);
}
int wsnark_fr_mul_dev(const void* d_a, const void* d_b, void* d_out, uint64_t n, void* stream) {
    REQUIRE_CTX();
    return fr_mul_dev((const Fe*)d_a, (const Fe*)d_b, (Fe*)d_out, n, (hipStream_t)stream);
}
int wsnark_fr_dist_combine_dev(const void* d_e, const void* d_o, void* d_h_out, uint64_t rows, uint64_t cols, uint64_t row0, uint32_t log_n1,
                               uint32_t log_n, void* stream) {
    REQUIRE_CTX();
    return dist_combine_dev((const Fe*)d_e, (const Fe*)d_o, (Fe*)d_h_out, rows, cols, row0, log_n1, log_n, (hipStream_t)stream);
}
int wsnark_groth16_verify(const void* vk, size_t vk_len, const void* inputs, uint64_t n_inputs, const void* proof384, int* valid) {
    if (!vk || !proof384 || !valid || (n_inputs && !inputs)) return WSNARK_ERR_ARG;
    return groth16_verify((const uint8_t*)vk, vk_len, (const uint8_t*)inputs, n_inputs, (const uint8_t*)proof384, valid);
}
int wsnark_last_blinding(void* r32, void* s32) {
    return last_blinding((uint8_t*)r32, (uint8_t*)s32) ? WSNARK_OK : WSNARK_ERR_ARG;
}
int wsnark_groth16_prove_finish(wsnark_pkey_t* h, const void* partials, uint64_t n_ranks, const void* r32,
                                const void* s32, void* out384) {
    if (!h || !out384 || (n_ranks && !partials)) return WSNARK_ERR_ARG;
    return groth16_prove_finish(reinterpret_cast<ProvingKey*>(h), (const uint8_t*)partials, n_ranks,
                                (const uint8_t*)r32, (const uint8_t*)s32, (uint8_t*)out384);
}

// ---- synthetic-input helpers (no reference counterpart) ----
int wsnark_g1_mul_base_batch(const void* base64, const void* scalars, uint64_t n, void* out_affine) {
    REQUIRE_CTX();
    return g1_mul_base_batch(base64, scalars, n, out_affine);
}
int wsnark_g2_mul_base_batch(const void* base128, const void* scalars, uint64_t n, void* out_affine) {
    REQUIRE_CTX();
    return g2_mul_base_batch(base128, scalars, n, out_affine);
}

// ---- device self-test hooks (tests only; selftest.hip) ----
int wsnark_selftest_field(int which, int impl, int op, const void* a, const void* b, void* out, uint64_t n) {
    REQUIRE_CTX();
    if (n && (!a || !b || !out)) return WSNARK_ERR_ARG;
    return selftest_field(which, impl, op, (const uint8_t*)a, (const uint8_t*)b, (uint8_t*)out, n);
}
int wsnark_selftest_curve(int g, int impl, int op, const void* p, const void* q, void* out, uint64_t n) {
    REQUIRE_CTX();
    if (n && (!p || !q || !out)) return WSNARK_ERR_ARG;
    return selftest_curve(g, impl, op, (const uint8_t*)p, (const uint8_t*)q, (uint8_t*)out, n);
}

int wsnark_peak_probe(int probe, double* gops_per_s) {
    REQUIRE_CTX();
    return peak_probe(probe, gops_per_s);
}

// ---- pinned host buffers ----
int wsnark_host_alloc(size_t bytes, void** out) {
    REQUIRE_CTX();
    if (!out) return WSNARK_ERR_ARG;
    *out = nullptr;
    WS_HIP_CHECK(hipHostMalloc(out, bytes ? bytes : 16, 0));
    return WSNARK_OK;
}
void wsnark_host_free(void* p) {
    if (!p) return;
    if (Context* C = ctx()) (void)hipSetDevice(C->device);
    (void)hipHostFree(p);
}

// ---- measurement switches ----
int wsnark_tuning_set(const char* name, int64_t value) {
    if (!name || !*name) return WSNARK_ERR_ARG;
    tuning_set(name, value == INT64_MIN ? LONG_MIN : (long)value);
    return WSNARK_OK;
}

// ---- timing ----
void wsnark_timing_enable(int on) {
    Context* C = ctx();
    if (C) { C->timer.enabled = on != 0; C->timer.dominant_only = on == 2; }
}
void wsnark_timing_reset(void) {
    Context* C = ctx();
    if (C) C->timer.reset();
}
size_t wsnark_timing_report(char* buf, size_t cap) {
    Context* C = ctx();
    if (!C) return 0;
    (void)hipSetDevice(C->device);
    (void)hipStreamSynchronize(C->stream);
    for (int i = 0; i < C->n_lanes; i++) {
        (void)hipStreamSynchronize(C->lanes[i].stream);
        (void)hipStreamSynchronize(C->lanes[i].stream2);
    }
    C->timer.collect();
    std::string s;
    for (auto& kv : C->timer.acc) {
        char line[256];
        snprintf(line, sizeof line, "%s %.6f %llu\n", kv.first.c_str(), kv.second.first, (unsigned long long)kv.second.second);
        s += line;
    }
    if (buf && cap) {
        size_t k = s.size() < cap - 1 ? s.size() : cap - 1;
        memcpy(buf, s.data(), k);
        buf[k] = 0;
    }
    return s.size();
}

}  // extern "C"
