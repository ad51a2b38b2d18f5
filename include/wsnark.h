/*
 * wsnark.h -- C ABI of libwsnark.so: the MI355X-native replacement for the BN128
 * Groth16 prove hot path of iden3/wasmsnark.
 *
 * Every entry point names the reference interface it replaces (paths relative to the
 * reference root, /root/reference in the build container).  The seam is the one the
 * reference itself has: the three worker commands G1_MULTIEXP / G2_MULTIEXP / CALC_H
 * (src/bn128.js:102-166), their host wrappers Bn128.g1_multiexp / g2_multiexp / calcH
 * (src/bn128.js:353-415, 569-578) and Bn128.groth16GenProof (src/bn128.js:580-720).
 * INTEGRATION.md shows the N-API / ctypes bindings over this header.
 *
 * Byte layouts are the reference's (tools/buildpkey.js:57-77, tools/buildwitness.js:36-41):
 *   field element  32 B little-endian; Montgomery form (R = 2^256) unless stated "plain"
 *   G1 affine      64 B  (x, y)                x == 0  => point at infinity
 *   G2 affine      128 B (x.c0, x.c1, y.c0, y.c1)
 *   G1 Jacobian    96 B  (x, y, z)             z == 0  => infinity; canonical (0, 1, 0)
 *   G2 Jacobian    192 B
 *   scalars        32 B raw 256-bit little-endian, NOT required to be < r
 *
 * All functions return 0 on success or a WSNARK_ERR_* code; none throws or aborts.
 * wsnark_last_error() gives a thread-local human-readable message.
 * Input pointers are borrowed for the duration of the call only.
 * "_dev" variants take DEVICE pointers (hipMalloc / torch tensor data_ptr) and a
 * hipStream_t (as void*; NULL = the library's own stream); results that are a single
 * group element are still written to HOST memory.
 *
 * Threading: every entry point may be called from any host thread at any time (the Node addon calls from
 * the libuv pool); each call selects the context's GPU for its thread first.  A context has a few LANES
 * (WSNARK_LANES, default 2): a lane holds everything one call in flight needs besides the read-only key (queues,
 * MSM plans, transform scratch, per-proof buffers), so two proofs -- on one key handle or on two -- or a proof and
 * an MSM overlap on one GPU; further concurrent callers wait for a lane.  There is no process-global mode: sharding
 * parameters are per call.  Unlike the reference's WASM instance (static scratch,
 * non-re-entrant) no caller-side queue is needed.
 * Host buffers are read through a pinned staging ring and are free for reuse when the call returns.
 */
#ifndef WSNARK_H
#define WSNARK_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WSNARK_OK 0
#define WSNARK_ERR_SIZE 1    /* n not a power of two, > 2^28, or inconsistent lengths */
#define WSNARK_ERR_FORMAT 2  /* malformed proving key / pols records / offsets out of range */
#define WSNARK_ERR_HIP 3     /* HIP runtime error, see wsnark_last_error() */
#define WSNARK_ERR_ARG 4     /* NULL pointer or bad handle */
#define WSNARK_ERR_NOINIT 5  /* wsnark_init() not called */

typedef struct wsnark_pkey wsnark_pkey_t;

/* Replaces build() / worker INIT (src/bn128.js:173-265, 55-66): selects the GPU, creates
 * the library stream.  device < 0 => use $LOCAL_RANK or 0.  Idempotent. */
int wsnark_init(int device);
/* Replaces Bn128.terminate() / worker TERMINATE (src/bn128.js:562-566, 167-169). */
void wsnark_shutdown(void);
const char* wsnark_last_error(void);
/* "<device name> <gcn arch> CUs=<n> device=<ordinal>" of the device in use */
const char* wsnark_device_info(void);

/* Worker command G1_MULTIEXP + Bn128.g1_multiexp (src/bn128.js:102-113, 353-383;
 * kernel g1m_multiexp2, src/build_multiexp.js:651-744).  Host pointers.
 * out96 = sum_i scalars[i]*points[i] as Jacobian-Montgomery, affine-normalised
 * ((x, y, 1) or (0, 1, 0)); n == 0 => infinity. */
int wsnark_g1_msm(const void* scalars, const void* points_affine, uint64_t n, void* out96);
/* Worker command G2_MULTIEXP + Bn128.g2_multiexp (src/bn128.js:114-125, 385-415;
 * kernel g2m_multiexp, src/build_multiexp.js:498-580). */
int wsnark_g2_msm(const void* scalars, const void* points_affine, uint64_t n, void* out192);
int wsnark_g1_msm_dev(const void* d_scalars, const void* d_points_affine, uint64_t n, void* out96_host, void* stream);
int wsnark_g2_msm_dev(const void* d_scalars, const void* d_points_affine, uint64_t n, void* out192_host, void* stream);

/* The gather step of Bn128.g1_multiexp / g2_multiexp (src/bn128.js:374-382, 406-414): serial EC
 * sum of `count` Jacobian-Montgomery partial results (any z), host arithmetic, no GPU needed.
 * Used to combine per-GPU partial MSMs after the one all-gather of a sharded run.
 * out = affine-normalised Jacobian-Montgomery. */
int wsnark_g1_sum(const void* jac_points, uint64_t count, void* out96);
int wsnark_g2_sum(const void* jac_points, uint64_t count, void* out192);

/* Multi-GPU alternative to splitting the pairs (the shard is a PER-CALL argument): every rank is given ALL pairs
 * and computes only the Pippenger windows w with w % world == rank; its result is then a partial sum already scaled
 * by 2^(c*w), so the partials of all ranks combine by plain EC addition (all_gather + wsnark_g*_sum) -- the worker
 * split and gather of src/bn128.js:353-415 with GPUs as workers.  (rank 0, world 1) is the whole sum. */
int wsnark_g1_msm_windows(const void* scalars, const void* points_affine, uint64_t n, uint32_t rank, uint32_t world, void* out96);
int wsnark_g2_msm_windows(const void* scalars, const void* points_affine, uint64_t n, uint32_t rank, uint32_t world, void* out192);
int wsnark_g1_msm_windows_dev(const void* d_scalars, const void* d_points_affine, uint64_t n, uint32_t rank, uint32_t world,
                              void* out96_host, void* stream);
int wsnark_g2_msm_windows_dev(const void* d_scalars, const void* d_points_affine, uint64_t n, uint32_t rank, uint32_t world,
                              void* out192_host, void* stream);

/* fft_fft / fft_ifft (src/build_fft.js:159-221), in place on n Montgomery Fr elements.
 * n must be a power of two <= 2^28 (the reference traps otherwise, :137-154);
 * inverse with n == 1 is rejected (the reference never returns, :575-583). */
int wsnark_fr_ntt(void* buf, uint64_t n, int odd, int inverse);
int wsnark_fr_ntt_dev(void* d_buf, uint64_t n, int odd, int inverse, void* stream);
/* Building blocks of the DISTRIBUTED transform (four-step, n = n1 * n2 over the ranks of one node; orchestrated by
 * wasmsnark_amd/dist.py: dist_ntt -- the reference never parallelises a transform, src/bn128.js:126-166 runs CALC_H
 * on one worker).  batch: `count` independent length-n transforms stored back to back (the column step and the row
 * step), same semantics per transform as wsnark_fr_ntt_dev with odd = 0.  dist_scale: element (r, c) of a rank's
 * rows x cols row-major block stands at global position t = (row0 + r) + 2^log_n1 * c of the length-2^log_n vector
 * (cols must be 2^(log_n - log_n1)); `stack` such blocks -- the slices of several vectors that go through the same
 * transform together, one exchange for all of them -- lie one after the other; every element is multiplied by
 *   mode 0: w_n^((row0 + r) * c)  -- the twiddle between the two steps (inverse != 0: the inverse root)
 *   mode 1: w_2n^t                -- the coset pre-scale of odd = 1 (src/build_fft.js:159-187) */
int wsnark_fr_ntt_batch_dev(void* d_buf, uint64_t n, uint64_t count, int inverse, void* stream);
/* The other pieces of a distributed CALC_H (src/bn128.js:126-166), all on device arrays:
 *   pkey_eval_ab : a = A w, b = B w -- the two pol_constructLC calls (src/build_pol.js:62-144, bn128.js:139-145) from a
 *                  device-resident plain witness; `domain` Montgomery elements each
 *   fr_mul       : fft_mulN (src/build_fft.js:461-505), out[i] = a[i] * b[i]
 *   dist_combine : h[t] = fromMontgomery((e[t] - w_2n^-t o[t]) / 2) on a rank's block, t as in dist_scale (the upper
 *                  half of the reference's size-2n inverse transform, bn128.js:160-164; derivation in csrc/calch.hip) */
int wsnark_pkey_eval_ab_dev(wsnark_pkey_t* handle, const void* d_witness, size_t witness_len, void* d_a_out, void* d_b_out, void* stream);
int wsnark_fr_mul_dev(const void* d_a, const void* d_b, void* d_out, uint64_t n, void* stream);
int wsnark_fr_dist_combine_dev(const void* d_e, const void* d_o, void* d_h_out, uint64_t rows, uint64_t cols, uint64_t row0,
                               uint32_t log_n1, uint32_t log_n, void* stream);
int wsnark_fr_dist_scale_dev(void* d_buf, uint64_t stack, uint64_t rows, uint64_t cols, uint64_t row0, uint32_t log_n1, uint32_t log_n,
                             int mode, int inverse, void* stream);
/* fft_toMontgomeryN / fft_fromMontgomeryN (src/build_fft.js:418-458, 507-547) */
int wsnark_fr_to_montgomery(const void* in, void* out, uint64_t n);
int wsnark_fr_from_montgomery(const void* in, void* out, uint64_t n);

/* Worker command CALC_H + Bn128.calcH (src/bn128.js:126-166, 569-578).
 * signals: n_signals x 32 B plain; polsA/polsB: the key's record streams
 * (u32 ncoefs, then ncoefs x (u32 idx, 32 B coef)) per signal (src/build_pol.js:62-144);
 * out_h: domain x 32 B plain. */
int wsnark_calc_h(const void* signals, const void* polsA, size_t lenA, const void* polsB, size_t lenB,
                  uint32_t n_signals, uint32_t domain, void* out_h);

/* Parses proving_key.bin (tools/buildpkey.js:124-240; header read at src/bn128.js:581-604)
 * and makes it device-resident (points as-is, pols transposed to row-major CSR). */
int wsnark_pkey_load(const void* pkey, size_t len, wsnark_pkey_t** out_handle);
void wsnark_pkey_free(wsnark_pkey_t* handle);
int wsnark_pkey_info(const wsnark_pkey_t* handle, uint32_t* n_vars, uint32_t* n_public, uint32_t* domain);
/* How the key's five point sections are resident (no counterpart in the reference, which reads the sections in place:
 * src/bn128.js:592-604).  By default each section is kept as a fixed-base window table -- rows x n points, row w =
 * 2^(c w) * section -- so that a sum is `rows` passes into one bucket set (13 rows, c = 20, at 2^20 pairs; about 6 GB
 * for the five sections).  c_w / rows_w: the A, B1, B2, C tables (nVars pairs); c_h / rows_h: the hExps table
 * (domainSize pairs); bytes: device memory of all five.  Plain sections (WSNARK_KEY_TABLE=0, or tables that do not
 * fit: WSNARK_TABLE_MAX_GB, half of the free memory): c = 0, rows = 1.  Any out pointer may be NULL. */
int wsnark_pkey_table_info(const wsnark_pkey_t* handle, uint32_t* c_w, uint32_t* rows_w, uint32_t* c_h, uint32_t* rows_h,
                           uint64_t* bytes);

/* The table rows beyond the plain sections are built in the BACKGROUND (the context's build queue, lowest stream priority, one key
 * after the other): wsnark_pkey_load* return as soon as the sections are resident (about 30 ms for a 0.59 GB key instead of 214),
 * proofs that start before the build is over (~0.17 s at 2^20) run on the plain sections -- same results, about twice the
 * steady-state time while the build shares the GPU with them (10 % over it on an idle GPU) -- and later ones on the tables.  This call blocks until the handle's tables are
 * built (benchmarks; callers that want the steady state before their first proof).  WSNARK_TABLE_ASYNC=0 makes the load itself wait,
 * as in round 3.  Calls that shard the WINDOWS of a whole key over ranks (wsnark_groth16_prove_partial with world > 1 on a whole
 * key) wait by themselves: their partial sums must mean the same on every rank. */
int wsnark_pkey_wait_tables(wsnark_pkey_t* handle);

/* Wall-clock of the handle's load, in ms: [1] point sections host -> device, [2] infinity masks + conversion to the device
 * field's domain, [0] polsA / polsB -> CSR (header walk on the host, upload, transposition on the GPU), [3] fixed-base table
 * build, [4] the whole call -- what a cold caller waits before its first proof (the reference re-parses the key inside every
 * groth16GenProof call, src/bn128.js:581-604).  With the background build (the default) [4] = [1] + [2] + [0] and [3] is the
 * build's own duration on the GPU (0 until it is over); with WSNARK_TABLE_ASYNC=0 the call waits for it and [4] includes it. */
int wsnark_pkey_load_stats(const wsnark_pkey_t* handle, double* ms5);

/* The same key given as separate host buffers with 64-bit lengths: proving_key.bin addresses its
 * sections with u32 byte offsets (tools/buildpkey.js:133-139), which caps a key at 4 GiB (~2^23
 * constraints); this entry point is the container for larger keys (BASELINE config 5).
 * pointsC holds nVars-nPublic-1 points (signals nPublic+1..), as in the file. */
typedef struct {
    uint32_t n_vars, n_public, domain;
    const void *alfa1, *beta1, *delta1;      /* 64 B each  */
    const void *beta2, *delta2;              /* 128 B each */
    const void* polsA; uint64_t polsA_len;   /* record streams, src/build_pol.js:62-144 */
    const void* polsB; uint64_t polsB_len;
    /* point arrays with the number of BYTES readable behind each pointer; a section shorter than the header
     * implies (nVars x 64, nVars x 64, nVars x 128, (nVars-nPublic-1) x 64, domain x 64) is WSNARK_ERR_FORMAT */
    const void* pointsA;  uint64_t pointsA_len;
    const void* pointsB1; uint64_t pointsB1_len;
    const void* pointsB2; uint64_t pointsB2_len;
    const void* pointsC;  uint64_t pointsC_len;
    const void* pointsH;  uint64_t pointsH_len;
} wsnark_key_sections_t;
int wsnark_pkey_load_sections(const wsnark_key_sections_t* sections, wsnark_pkey_t** out_handle);

/* Multi-GPU: ONE RANK'S SHARE of a key, split by points -- the reference's own worker split (src/bn128.js:353-361:
 * contiguous ranges of the pairs, the remainder to the last worker) with GPUs as workers.  Every rank passes the same
 * (complete) sections and keeps only the points of signals [rank * floor(nVars / world), ...) of A, B1, B2 and C and its
 * share of hExps: 1 / world of the device memory and of the additions of every sum, uniform whatever the window count
 * (wsnark_pkey_table_info reports the shard's own tables).  The two sparse matrices stay complete (CALC_H needs all rows).
 * h_interleave_log = 0: the hExps share is the contiguous range [rank * floor(domain / world), ...): every rank computes
 * the whole h (wsnark_groth16_prove_partial without WSNARK_PARTIAL_SKIP_H).  h_interleave_log = k > 0 (world a power of
 * two, world <= 2^k <= domain): the share is the rank's rows of the 2^k-interleaved layout in which the distributed CALC_H
 * leaves its slice of h -- local element (r, j) = hExps[(rank * 2^k / world + r) + 2^k * j], row-major -- to be summed with
 * wsnark_pkey_h_msm_dev.  On such a handle wsnark_groth16_prove_partial[_dev] must be called with the handle's own (rank,
 * world) and returns the partial sums over the handle's pairs (all windows); wsnark_groth16_prove[_dev] is WSNARK_ERR_ARG;
 * wsnark_groth16_prove_finish works on any handle of the key (it only needs the five fixed points). */
int wsnark_pkey_load_shard(const wsnark_key_sections_t* sections, uint32_t rank, uint32_t world, uint32_t h_interleave_log,
                           wsnark_pkey_t** out_handle);
/* A key FILE (round 6): `path` names the reference's proving_key.bin (tools/buildpkey.js:124-186: u32
 * section offsets, at most 4 GiB) or the WSNARK64 container -- the same sections in the same order behind 64-bit offsets, for keys the
 * reference's format cannot hold (BASELINE config 5, 7.8 GB at 2^24); layout: wasmsnark_amd/csrc/keyfile.hip, written by
 * wasmsnark_amd/formats.py / js/formats.js.  The file is mapped read-only and ONLY the pages of the share asked for are read:
 * (rank, world, h_interleave_log) as in wsnark_pkey_load_shard -- (0, 1, 0) loads the whole key -- so eight ranks read the five point
 * sections once between them (each the two matrices); every range goes back to the kernel as soon as it has been staged, so the
 * load's resident set does not grow with the key.  Replaces the caller-side `fs.readFileSync` + `new Uint32Array(pkey)` of
 * src/bn128.js:580-604 for keys that do not fit one ArrayBuffer.  Errors: WSNARK_ERR_ARG (cannot open), WSNARK_ERR_FORMAT. */
int wsnark_pkey_load_file(const char* path, uint32_t rank, uint32_t world, uint32_t h_interleave_log, wsnark_pkey_t** out_handle);
/* header of a key file without loading it (no GPU needed).  format: 1 = proving_key.bin, 2 = WSNARK64.  Out pointers may be NULL. */
int wsnark_pkey_file_info(const char* path, uint32_t* n_vars, uint32_t* n_public, uint32_t* domain, uint64_t* file_bytes, int* format);
/* which share a handle holds: (0, 1, 0, nVars, domain, 0) for a whole key.  Any out pointer may be NULL. */
int wsnark_pkey_shard_info(const wsnark_pkey_t* handle, uint32_t* rank, uint32_t* world, uint64_t* first_signal,
                           uint64_t* n_signals, uint64_t* n_hexps, uint32_t* h_interleave_log);
/* sum_i h[i] * hExps_local[i] over the handle's resident hExps share (n must equal its n_hexps; d_h_slice: device, plain
 * form, in the share's layout) -- the H call of src/bn128.js:614 for one rank.  out96: Jacobian-Montgomery, normalised. */
int wsnark_pkey_h_msm_dev(wsnark_pkey_t* handle, const void* d_h_slice, uint64_t n, void* out96_host, void* stream);

/* Bn128.groth16GenProof (src/bn128.js:580-720).  witness: nVars x 32 B plain
 * (tools/buildwitness.js:36-41); witness_len < nVars * 32 is WSNARK_ERR_SIZE, a LONGER buffer is accepted and only its first
 * nVars signals are read -- exactly what the reference does with an over-long signals buffer (it walks nSignals records,
 * src/bn128.js:607-620).  r32 / s32: the two 32-byte blinding values the reference
 * draws from crypto.randomBytes (src/bn128.js:642-661); NULL => drawn from the OS CSPRNG.
 * out384 = pi_a (x, y, z) 96 B | pi_b (x.c0, x.c1, y.c0, y.c1, z.c0, z.c1) 192 B |
 * pi_c 96 B: affine, PLAIN (non-Montgomery) little-endian, i.e. exactly the integers
 * bin2g1/bin2g2 print (src/bn128.js:329-351, 706-718); infinity = (0, 1, 0). */
int wsnark_groth16_prove(wsnark_pkey_t* handle, const void* witness, size_t witness_len, const void* r32,
                         const void* s32, void* out384);
/* same, witness already on the device */
int wsnark_groth16_prove_dev(wsnark_pkey_t* handle, const void* d_witness, size_t witness_len, const void* r32,
                             const void* s32, void* out384_host, void* stream);

/* Host buffers the GPU can read in place (no counterpart in the reference, whose inputs live in the WASM heap).  A witness --
 * or the scalars / points of an MSM -- that is WRITTEN into such a buffer is DMA'd straight from it, chunk by chunk; any other
 * host pointer is first copied into the library's pinned staging ring by worker threads (about as fast as the link, but it
 * costs host cores and a second pass over the bytes).  The Node addon hands these out as external ArrayBuffers
 * (Bn128.allocInput).  Needs wsnark_init; free with wsnark_host_free (NULL is ignored). */
int wsnark_host_alloc(size_t bytes, void** out);
void wsnark_host_free(void* p);

/* Bn128.groth16Verify (src/bn128.js:722-791; pairing bn128_pairingEq4, src/bn128/build_bn128.js:265-1374): native host
 * arithmetic, no GPU and no wsnark_init needed.  Checks e(A,B) e(-IC(inputs),gamma2) e(-C,delta2) e(-alfa1,beta2) == 1.
 *   vk     : alfa1 (64 B) | beta2 (128 B) | gamma2 (128 B) | delta2 (128 B) | IC[0 .. n_inputs] (64 B each) -- the points
 *            of verification_key.json as affine PLAIN (non-Montgomery) little-endian integers, G2 as (x.c0, x.c1, y.c0, y.c1)
 *   inputs : n_inputs x 32 B plain little-endian public signals; one >= r gives *valid = 0 like the reference (:772)
 *   proof384: what wsnark_groth16_prove writes (pi_a | pi_b | pi_c with their z coordinates).  The z coordinates are IGNORED,
 *            as the reference's setG1Affine / setG2Affine do (src/bn128.js:741-760): (x, y) is the point.  Consequence: a proof
 *            with a component at infinity -- printed (0, 1, 0); reachable only with r = s = 0 on degenerate witnesses -- is read
 *            as the point (0, 1), which is not on the curve, and is therefore always INVALID here
 * Returns WSNARK_OK with *valid = 1 / 0; WSNARK_ERR_FORMAT if a coordinate is not a reduced field element,
 * WSNARK_ERR_SIZE if vk holds fewer than n_inputs + 1 IC points.  Unlike the reference, which makes no such test (its
 * verdict on malformed points is an accident of its Miller loop), a point that is not on its curve, or a G2 point outside
 * the order-r subgroup, makes the proof invalid (*valid = 0).  Like the reference (src/bn128.js:741-760 force z = 1) the z
 * coordinates of the proof are ignored: (x, y) is the point. */
int wsnark_groth16_verify(const void* vk, size_t vk_len, const void* inputs, uint64_t n_inputs, const void* proof384, int* valid);

/* The two 32-byte blinding values of the last proof assembled by the CALLING THREAD (wsnark_groth16_prove[_dev] or
 * _prove_finish), whether injected or drawn from the OS CSPRNG: the reference keeps them the same way, "for tests",
 * as this._pr / this._ps (src/bn128.js:662-664).  WSNARK_ERR_ARG if this thread has not proved yet. */
int wsnark_last_blinding(void* r32_out, void* s32_out);

/* Multi-GPU proving (one process per GPU; windows sharded as in wsnark_g1_msm_windows, rank / world per call).
 * prove_partial runs CALC_H and the five MSMs on this rank's windows and writes ONE 576-byte record:
 * A | B1 | C | H (4 x 96 B G1) | B2 (192 B G2), Jacobian-Montgomery -- the reference's per-worker
 * partial results (src/bn128.js:374-382, 406-414) for all five sums at once.  After a single
 * all_gather of these records, prove_finish (host arithmetic only) sums them and assembles the proof
 * exactly as wsnark_groth16_prove does (src/bn128.js:671-718).
 * flags: WSNARK_PARTIAL_SKIP_H leaves CALC_H and the H sum to the caller (the record's H slot is infinity): the ranks then
 * compute h with the distributed four-step transform and each sums its own slice of h against its slice of the H points
 * (wasmsnark_amd/dist.py: DistProver) instead of every rank repeating the whole CALC_H. */
#define WSNARK_PARTIAL_SKIP_H 1u
int wsnark_groth16_prove_partial(wsnark_pkey_t* handle, const void* witness, size_t witness_len, uint32_t rank,
                                 uint32_t world, uint32_t flags, void* out576);
int wsnark_groth16_prove_partial_dev(wsnark_pkey_t* handle, const void* d_witness, size_t witness_len, uint32_t rank,
                                     uint32_t world, uint32_t flags, void* out576_host, void* stream);
int wsnark_groth16_prove_finish(wsnark_pkey_t* handle, const void* partials, uint64_t n_ranks, const void* r32,
                                const void* s32, void* out384);

/* ONE CALL per proof and rank for the whole multi-GPU prover (csrc/dist.hip): the rank's partial sums over its points shard,
 * CALC_H on the distributed four-step transform (three all-to-alls per proof), the H sum over the rank's hExps share, one
 * all-gather of the 576-byte records and the host-side assembly -- no host language between the kernels.  The library links
 * no collectives library: the host passes its transport as two callbacks (torch.distributed on RCCL in
 * wasmsnark_amd/dist.py; MPI, a thread pool, ... elsewhere):
 *   d_send / d_recv  two device buffers of buf_bytes >= 3 * 32 * domain / world each, owned by the host side
 *   all_to_all(user, bytes_per_rank, stream)   block q (bytes_per_rank bytes) of d_send goes to rank q, block q of d_recv comes
 *                    from rank q; must be ORDERED ON `stream` (the library's kernels that fill d_send were enqueued on it and
 *                    the ones that read d_recv follow on it) -- enqueue it there, or synchronise it; return 0 on success
 *   all_gather(user, send, recv, bytes)        host memory: recv = the `bytes`-byte records of all ranks in rank order
 * handle: the rank's points shard with h_interleave_log = floor(log2(domain) / 2) (wsnark_pkey_load_shard); world must be a
 * power of two <= 2^floor(log2(domain) / 2).  world == 1: callbacks may be NULL (the exchange is the identity).
 * Collectives per proof, posted by EVERY rank in the same order whatever happens on it: all_gather (80-byte records: the
 * rank's preflight status, which of r / s it was given, the blinding bytes -- rank 0's draw when r32 / s32 are NULL -- before
 * the GPU work, so that the host's key-only scalar multiplications run under it), three all_to_all, all_gather (592-byte
 * records: the 576 bytes of partial sums + the rank's status).  EVERY RANK MUST PASS THE SAME r32 / s32 (all NULL, or the
 * same bytes): ranks that disagree all return WSNARK_ERR_ARG.  A rank that fails locally still posts the exchanges its peers
 * are waiting in and reports through the last gather: every rank then returns an error, nobody is left in a collective
 * (what remains the transport's job: a callback that itself fails or hangs on one rank).
 * stream: the queue d_witness is ready on (NULL: the library's own). */
typedef struct {
    uint32_t rank, world;
    void* d_send;
    void* d_recv;
    uint64_t buf_bytes;
    int (*all_to_all)(void* user, uint64_t bytes_per_rank, void* stream);
    int (*all_gather)(void* user, const void* send, void* recv, uint64_t bytes);
    void* user;
} wsnark_comm_t;
int wsnark_groth16_prove_dist(wsnark_pkey_t* handle, const void* d_witness, size_t witness_len, const wsnark_comm_t* comm,
                              const void* r32, const void* s32, void* out384_host, void* stream);

/* ---- resident bases (round 5; csrc/fixedbase.hip): NO reference counterpart ----
 * The reference's g1_multiexp / g2_multiexp (src/bn128.js:353-415) take the points with every call.  A caller that sums over the
 * SAME bases repeatedly can make them resident once -- as fixed-base window tables, the layout a resident proving key's sections
 * have: rows x n points, row w = 2^(c w) * P, c = log2 n (13 rows and 13 x the bytes at 2^20) -- and then pays neither the points'
 * H2D copy nor the per-window plans and the host's doubling chain: every sum is one bucket set and one reduction tail.
 *   group: 1 = G1 (64-byte affine Montgomery points), 2 = G2 (128-byte); x == 0 is infinity, as everywhere.
 *   wsnark_points_msm[_dev]: n must be the set's size (one raw 256-bit scalar per point); out = the Jacobian-Montgomery triple
 *   (96 / 192 bytes, affine-normalised) that wsnark_g{1,2}_msm returns for the same pairs. */
typedef struct wsnark_points wsnark_points_t;
int wsnark_points_load(int group, const void* points, uint64_t n, wsnark_points_t** out_handle);
void wsnark_points_free(wsnark_points_t* handle);
int wsnark_points_info(const wsnark_points_t* handle, int* group, uint64_t* n, uint32_t* window_bits, uint32_t* rows, uint64_t* table_bytes);
int wsnark_points_msm(wsnark_points_t* handle, const void* scalars, uint64_t n, void* out);
int wsnark_points_msm_dev(wsnark_points_t* handle, const void* d_scalars, uint64_t n, void* out_host, void* stream);

/* ---- several GPUs in ONE process (csrc/group.hip; round 5) ----
 * The reference's host is one process that starts W workers (src/bn128.js:173-265, `build()`), cuts every multi-exponentiation
 * into W contiguous ranges of the pairs, posts one to each worker and adds the partial results (:353-415), and runs the five
 * sums of a proof that way (:607-622).  A group is that arrangement with GPUs as the workers, for hosts that are ONE process
 * (the Node.js drop-in): one context and one host thread per device, the transport between the devices inside the library
 * (device-to-device copies ordered by events for the distributed CALC_H's three exchanges; the 576-byte records gathered in
 * host memory).  Hosts that run one process per GPU keep using wsnark_groth16_prove_dist with their own transport.
 *   wsnark_group_create      devices[n]: HIP device ordinals (the same ordinal may appear more than once: two contexts on one
 *                            GPU -- how the path is tested on a single-GPU box).  Independent of wsnark_init.
 *   wsnark_group_pkey_load*  one POINTS SHARD of the key per device (wsnark_pkey_load_shard: pairs [g floor(n/N), ...), all
 *                            fixed-base table rows of that range, 1 / N of the key's memory each; the two matrices complete)
 *   wsnark_group_prove       = wsnark_groth16_prove over the group: same inputs, same 384 bytes.  CALC_H runs on the
 *                            distributed four-step transform when N is a power of two <= 2^floor(log2(domain) / 2)
 *                            (wsnark_group_pkey_info reports which), otherwise complete on every device.
 *   wsnark_group_g{1,2}_msm  = wsnark_g{1,2}_msm with the reference's split of the pairs over the devices
 * Calls on one group are serialised (one collective at a time); different groups are independent.
 * OWNERSHIP: a group owns its keys.  wsnark_group_free waits for the group's call in flight, then frees the group AND every key
 * handle still loaded on it: those handles are invalid afterwards and must not be passed to wsnark_group_pkey_free (the Python and
 * Node hosts forget them when the group dies).  The caller must not free a group while another thread may still START a call on
 * it (the Node addon counts its queued jobs and defers the free to the last one). */
typedef struct wsnark_group wsnark_group_t;
typedef struct wsnark_group_pkey wsnark_group_pkey_t;
int wsnark_group_create(const int* devices, uint32_t n, wsnark_group_t** out_group);
void wsnark_group_free(wsnark_group_t* group);
uint32_t wsnark_group_size(const wsnark_group_t* group);
int wsnark_group_pkey_load(wsnark_group_t* group, const void* pkey, size_t len, wsnark_group_pkey_t** out_handle);
int wsnark_group_pkey_load_sections(wsnark_group_t* group, const wsnark_key_sections_t* ks, wsnark_group_pkey_t** out_handle);
/* the same from a key file (wsnark_pkey_load_file): ONE read-only mapping, every member reads its own shard's pages */
int wsnark_group_pkey_load_file(wsnark_group_t* group, const char* path, wsnark_group_pkey_t** out_handle);
void wsnark_group_pkey_free(wsnark_group_pkey_t* handle);
int wsnark_group_pkey_info(const wsnark_group_pkey_t* handle, uint32_t* n_vars, uint32_t* n_public, uint32_t* domain, uint32_t* world,
                           int* distributed_calc_h);
int wsnark_group_pkey_wait_tables(wsnark_group_pkey_t* handle);
int wsnark_group_prove(wsnark_group_pkey_t* handle, const void* witness, size_t witness_len, const void* r32, const void* s32,
                       void* out384);
int wsnark_group_last_blinding(wsnark_group_t* group, void* r32, void* s32);     /* = wsnark_last_blinding for the group's last proof */
int wsnark_group_g1_msm(wsnark_group_t* group, const void* scalars, const void* points, uint64_t n, void* out96);
int wsnark_group_g2_msm(wsnark_group_t* group, const void* scalars, const void* points, uint64_t n, void* out192);

/* ---- synthetic-input helpers: NO reference counterpart ----
 * out[i] = scalars[i] * base (affine Montgomery in and out; infinity written as all-zero bytes).
 * The reference ships no proving key (its test/data/proving_key.bin is absent), so benches and
 * tests build valid synthetic keys from known toxic waste with these (wasmsnark_amd/synth.py). */
int wsnark_g1_mul_base_batch(const void* base64, const void* scalars, uint64_t n, void* out_affine);
int wsnark_g2_mul_base_batch(const void* base128, const void* scalars, uint64_t n, void* out_affine);

/* A whole synthetic circuit + trusted setup from KNOWN toxic waste, on the host (csrc/synth.hip): the multiplication-chain
 * R1CS SURVEY.md section 8d C4 specifies (style 0 = 1-3 non-zeros per COLUMN of A and B, every variable present; style 1 =
 * 1-2 terms per ROW, ~40 % of the variables absent from A resp. B; style 2 = bit decompositions: groups of 14 free bits with their
 * booleanity rows b (b - 1) = 0, a recomposition row and a product row -- 87.5 % of the witness is 0 / 1), its witness, its polsA / polsB record streams
 * (tools/buildpkey.js:79-89), the discrete logarithm of every key point (feed them to wsnark_g{1,2}_mul_base_batch) and
 * the discrete logarithms of the proof for given r, s -- the closed form the full-size parity tests compare against.
 * key_scalars group 1: alfa1, beta1, delta1, A[nVars], B1[nVars], C[nVars-nPublic-1], hExps[domain], IC[nPublic+1];
 * group 2: beta2, delta2, gamma2, B2[nVars]; 32-byte plain little-endian each.  expected: a | b | c, 32 B plain each. */
typedef struct wsnark_synth wsnark_synth_t;
typedef struct {
    uint32_t n_vars, n_public, domain;
    uint64_t nnz_a, nnz_b, absent_a, absent_b;   /* non-zeros; variables that never occur in A resp. B */
    uint64_t pols_a_len, pols_b_len;             /* bytes of the record streams */
    uint64_t n_g1_scalars, n_g2_scalars;
} wsnark_synth_info_t;
int wsnark_synth_new(uint32_t log_domain, uint32_t n_public, uint64_t circuit_seed, uint64_t setup_seed, int style,
                     wsnark_synth_t** out);
void wsnark_synth_free(wsnark_synth_t* h);
int wsnark_synth_info(const wsnark_synth_t* h, wsnark_synth_info_t* out);
int wsnark_synth_witness(const wsnark_synth_t* h, void* out_plain);
int wsnark_synth_pols(const wsnark_synth_t* h, int which, void* out, uint64_t cap);
int wsnark_synth_key_scalars(const wsnark_synth_t* h, int group, void* out);
int wsnark_synth_expected(const wsnark_synth_t* h, const void* r32, const void* s32, void* out96);

/* ---- device self-test hooks (tests/test_gpu_primitives.py): NO reference counterpart ----
 * The reference tests its field and group primitives directly (test/f1.js:296-400, test/bn128.js:84-185); here they
 * are __device__ code reached only through whole kernels, so these two entry points run ONE LANE PER VECTOR through
 * the device arithmetic itself.  Inputs and outputs are in the reference's formats; the conversion to the kernels'
 * internal representation and back is part of what is tested.
 *   which: 0 = Fq, 1 = Fr (32-byte elements), 2 = Fq2 (64-byte)
 *   impl : 0 = radix-2^29 lazy field (MSM / NTT kernels), 1 = saturated 4 x 64 field on the device (light kernels),
 *          2 = the host field (what proof assembly uses; runs on the CPU)
 *   op   : WSNARK_ST_* below; out is n x 32 (64 for Fq2) bytes. */
enum {
    WSNARK_ST_MUL = 0, WSNARK_ST_SQR = 1, WSNARK_ST_ADD = 2, WSNARK_ST_SUB = 3, WSNARK_ST_NEG = 4,
    WSNARK_ST_TOMONT = 5, WSNARK_ST_FROMMONT = 6,
    WSNARK_ST_SUB_WEAK = 7,      /* (a - b) through the uncorrected difference feeding a product     */
    WSNARK_ST_ADD_LAZY_MUL = 8,  /* (a + b) * b with the carry-free sum as a direct product operand  */
    WSNARK_ST_NEG_WEAK_MUL = 9,  /* (-a) * b through the 2p - a form                                 */
    WSNARK_ST_MUL2ADD = 10,      /* a*a + b*b with one Montgomery reduction                          */
    WSNARK_ST_MULSUB2 = 11,      /* (a - b)*a - b*a: fused, first operand an uncorrected difference   */
    WSNARK_ST_EQ = 12,           /* out = 1 if a == b else 0 (zero test of the strict difference)    */
    WSNARK_ST_EQ_WEAK = 13,      /* the same through the uncorrected difference                      */
    WSNARK_ST_INVERSE = 14,      /* 1/a (a != 0): impl 2 = the host's inversion (proof assembly), impl 0 / 1 (Fq, Fr) = the DEVICE's Fermat
                                    inversion a^(p-2) on that field -- what the table build's normalisation runs (msm_table_norm_kernel) */
    WSNARK_ST_SQR_WEAK = 15,     /* (a - b)^2 with the uncorrected difference as the operand of the squaring        */
    WSNARK_ST_MUL_WEAK_A = 16,   /* (a - b) * b with the uncorrected difference as the FIRST operand of the product */
    WSNARK_ST_MULSUB2_WEAK_B = 17 /* a*(a - b) - b*a: fused, second operand an uncorrected difference              */
};
int wsnark_selftest_field(int which, int impl, int op, const void* a, const void* b, void* out, uint64_t n);
/*   g: 1 or 2; impl: 0 = radix-2^29 curve of the accumulation kernels, 1 = saturated-field curve on the device,
 *   2 = host curve, 3 = (G1 only) the radix-2^29 variant of the reduction-tail kernels (inlined products), 4 = (G2 only) the
 *   reduction-tail variant with the quadratic extension's two components on two adjacent lanes (two lanes per vector).
 *   p, q: n Jacobian-Montgomery points (96 / 192 B, any z; z == 0 = infinity); out: n affine-normalised
 *   Jacobian-Montgomery points ((x, y, 1) or (0, 1, 0)) like every group element this ABI returns.
 *   op: 0 = p + q (full addition), 1 = 2p, 2 = -p, 3 = p (normalisation only), 4 = p + q as a MIXED addition
 *   (q must be affine: z == 1, or infinity), 5 = p - q as a mixed addition with the negate flag,
 *   6 = p + q + q and 7 = p + q - q as two mixed additions of the ACCUMULATION LOOP's lazy form (x kept "wide" between
 *   them, field29.h) followed by its narrowing, 8 = timesScalar (src/build_timesscalar.js:20-80): q's bytes are NOT a point
 *   but a little-endian scalar in bytes [0, 64) and its length -- 32 or 64 -- in byte 64 (impl 0-3). */
int wsnark_selftest_curve(int g, int impl, int op, const void* p, const void* q, void* out, uint64_t n);

/* ---- measurement hooks (bench.py) ---- */
/* A/B switches of the library (queue arrangement of a proof, reduction-tail geometry, ...; the names are the WSNARK_<name>
 * environment variables DESIGN.md lists, without the prefix): an override set here wins over the environment and is read by
 * every later call, so that one process can time several settings on one resident key.  value == INT64_MIN forgets the
 * override.  Results never depend on these switches -- only the schedule does. */
int wsnark_tuning_set(const char* name, int64_t value);
/* per-kernel HIP-event timing on the stream the kernels are launched on: 0 = off, 1 = every kernel,
 * 2 = only the dominant kernel (msm_accumulate_*), for timed regions where the brackets themselves must
 * stay out of the way */
void wsnark_timing_enable(int on);
void wsnark_timing_reset(void);
/* writes "name total_ms launches\n" lines; returns bytes needed (excluding NUL) */
size_t wsnark_timing_report(char* buf, size_t cap);
/* The integer roofline's peak, measured on the device in use (about 10 ms each): probe 0 = a dependent chain of the
 * library's own radix-2^29 Montgomery product on every lane, 8 x 256 lanes per CU (Gmodmul/s: what a kernel of nothing
 * but products reaches); 1 = the same with the inlined product body; 2 = eight independent v_mad_u64_u32 chains per lane
 * (Gmad/s: the raw 32x32+64 multiply-add issue rate).  3, 4 = traffic calibration kernels for the
 * FETCH_SIZE counter (run under rocprofv3 --pmc by tools/gpu_session.sh): 3 = 2^25 pseudo-random 64-byte point gathers out
 * of a 1 GiB table (the accumulation kernel's access pattern; 2 GiB of known bytes per launch, kernel
 * `probe_gather64_kernel`), 4 = a 16-B-per-lane streaming read of the same 1 GiB (`probe_stream16_kernel`); both return
 * GB/s of those known bytes.  5 = one field inversion per lane (the library's Fermat inversion on the radix-2^29 product,
 * every lane busy): G inversions/s -- what a lane-parallel batch inversion costs per batch. */
int wsnark_peak_probe(int probe, double* gops_per_s);

#ifdef __cplusplus
}
#endif
#endif
