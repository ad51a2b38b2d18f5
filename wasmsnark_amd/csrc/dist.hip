// dist.hip -- one proof over the ranks of one node, orchestrated behind the C ABI (wsnark_groth16_prove_dist).
//
// Reference shape: the reference's only parallel strategy is the worker split + gather of Bn128.g1_multiexp /
// g2_multiexp (/root/reference src/bn128.js:353-415) driven from groth16GenProof (:607-622), with CALC_H on ONE worker
// (:126-166).  Here the workers are GPUs, every rank holds a POINTS shard of the key (wsnark_pkey_load_shard: the
// reference's own contiguous split), and CALC_H -- which the reference never parallelises -- runs as a distributed
// four-step transform so that nothing but the two sparse-matrix structures is replicated:
//
//   queue 1   plan(witness slice) -> B2, A, B1, C partial sums over the rank's pairs           (prove.hip: prove_msms)
//   queue 2   a, b = rows of A w, B w that this rank's slice needs (row-sharded sparse products, written directly in the
//             layout the transform reads) -> E = a.b -> [iNTT of a, b, E: ONE exchange] -> [coset NTT of a, b: ONE
//             exchange] -> O = a.b -> [iNTT of O: ONE exchange] -> h slice -> H partial sum over the rank's hExps slice
//   host      ONE all-gather of the 576-byte records (+ rank 0's blinding bytes) -> prove_finish on every rank
//
// Four-step transform, n = n1 n2 (wasmsnark_amd/dist.py documents the layouts; this file is the same algorithm without
// Python between the kernels):  X[k2 + n2 k1] = sum_i1 w_n1^(i1 k1) [ w_n^(i1 k2) sum_i2 x[i1 + n1 i2] w_n2^(i2 k2) ].
// A rank holds r1 = n1 / P complete sub-sequences (rows i1, all i2): column step and twiddle are local; the pack kernel
// applies the twiddle while it writes the send buffer in exactly the block order the exchange needs; after the exchange
// the unpack kernel writes the (n2 / P) x n1 blocks the row step reads.  No permute().contiguous() round trips.
//
// Transport: the library does not link a collectives library.  The host passes two callbacks (all-to-all on device
// buffers it owns, ordered on the stream the library names; all-gather of small host records) -- torch.distributed on
// RCCL in wasmsnark_amd/dist.py, anything else a host has.
#include <string.h>

#include "../../include/wsnark.h"
#include "internal.h"
#include "field29.h"

namespace wsnark {

// res[(r, j)] = sum_k coef[k] * sig[col[k]] over row t = (row0 + r) + (j << log_n1) of the CSR matrix: pol_constructLC
// (src/build_pol.js:62-144) for the rows of one rank's slice only, stored in the slice's own (rows x cols) layout
__global__ __launch_bounds__(256) void lc_spmv_rows_kernel(const uint32_t* __restrict__ row_ptr, const uint32_t* __restrict__ col,
                                                             const Fe* __restrict__ coef, const Fe* __restrict__ sig,
                                                             uint32_t log_rows, uint64_t cols, uint64_t row0, uint32_t log_n1, Fe* __restrict__ res) {
    // consecutive lanes take CONSECUTIVE matrix rows (r fastest): the CSR row pointers, column indices and coefficients of a
    // wavefront are then contiguous; only the 32-byte results are stored with a stride (first version: j fastest, rows 2^log_n1
    // apart per lane -- 0.78 ms per launch at 2^20 against 0.16 ms for the single-GPU product, rocprofv3 r03_s17)
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (cols << log_rows)) return;
    const uint64_t r = idx & (((uint64_t)1 << log_rows) - 1), j = idx >> log_rows;
    const uint64_t t = (row0 + r) + (j << log_n1);
    res[r * cols + j] = lc_row_dot(coef, col, sig, row_ptr[t], row_ptr[t + 1]);
}

// send[q][v][r][c2] = x[v][r][q r2 + c2] * w_n^(+-(row0 + r)(q r2 + c2)): twiddle + block order of the exchange in one pass.
// r1, r2 and the world size are powers of two (lr1, lr2, lw): the index arithmetic is shifts and masks, blockIdx.y = the
// vector v of the stack (64-bit divisions by run-time values cost more than the two field products here).
// The twiddle is formed and applied on the radix-2^29 field: lo29 / hi29 are the two-level tables in its internal form
// (entry x 2^5), product of the two = the factor in internal form, times the unpacked element = the result in the reference
// Montgomery form (R = 2^256) after one canonicalisation -- 2 x 217 instructions instead of 2 x 584 on the saturated field.
__global__ __launch_bounds__(256) void dist_pack_kernel(const Fe* __restrict__ x, Fe* __restrict__ send, uint32_t k, uint32_t lr1, uint32_t lr2,
                                                          uint32_t lw, uint64_t row0, const Fe* __restrict__ lo29, const Fe* __restrict__ hi29, uint32_t h) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >> (lr1 + lr2 + lw)) return;
    const uint32_t v = blockIdx.y;
    const uint64_t c2 = t & (((uint64_t)1 << lr2) - 1), r = (t >> lr2) & (((uint64_t)1 << lr1) - 1), q = t >> (lr1 + lr2);
    const uint64_t c = (q << lr2) + c2;
    const uint64_t e = (row0 + r) * c;
    const F29 f = Fr29::mul(Fr29::unpack(hi29[e >> h]), Fr29::unpack(lo29[e & (((uint64_t)1 << h) - 1)]));
    const F29 y = Fr29::mul(Fr29::unpack(x[((((uint64_t)v << lr1) + r) << (lr2 + lw)) + c]), f);
    send[((((q * k + v) << lr1) + r) << lr2) + c2] = Fr29::pack(Fr29::canonical(y));
}
// y[v][c2][q r1 + r] = recv[q][v][r][c2]: the (n2 / P) x n1 blocks of the row step
__global__ __launch_bounds__(256) void dist_unpack_kernel(const Fe* __restrict__ recv, Fe* __restrict__ y, uint32_t k, uint32_t lr1, uint32_t lr2, uint32_t lw) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;       // index inside y[v]: c2 x n1
    if (t >> (lr1 + lr2 + lw)) return;
    const uint32_t v = blockIdx.y;
    const uint64_t i1 = t & (((uint64_t)1 << (lr1 + lw)) - 1), c2 = t >> (lr1 + lw);
    const uint64_t q = i1 >> lr1, r = i1 & (((uint64_t)1 << lr1) - 1);
    y[((uint64_t)v << (lr1 + lr2 + lw)) + t] = recv[((((q * k + v) << lr1) + r) << lr2) + c2];
}

// fft_fft / fft_ifft (src/build_fft.js:159-221) of `k` stacked vectors spread over the ranks: x = the rank's k blocks of
// r1 x n2 (n1-interleaved layout), overwritten; the result -- k blocks of r2 x n1, n2-interleaved -- is written to y.
// *exchanges (optional) counts the all-to-all callbacks this rank has posted: a caller that gives up half-way still owes its
// peers the remaining ones (prove.hip: groth16_prove_dist).
static int dist_ntt_native(Lane& L, const DistComm& cm, Fe* x, Fe* y, uint32_t log_n, uint32_t log_n1, int odd, int inverse, uint64_t k, hipStream_t s,
                           int* exchanges) {
    const uint32_t log_n2 = log_n - log_n1;
    const uint64_t n1 = (uint64_t)1 << log_n1, n2 = (uint64_t)1 << log_n2, P = cm.world;
    const uint64_t r1 = n1 / P, r2 = n2 / P, row0 = (uint64_t)cm.rank * r1;
    const uint64_t total = k * r1 * n2;
    if (total * sizeof(Fe) > cm.buf_bytes) { set_last_error("prove_dist: exchange buffers too small"); return WS_ERR_SIZE; }
    int rc;
    const Fe *lo, *hi;
    int h;
    if ((rc = ntt_twiddle_tables((int)log_n, inverse, &lo, &hi, &h, s, /* internal form of the radix-2^29 field */ true))) return rc;
    Context* C = ctx();
    uint32_t lw = 0;
    while (((uint64_t)1 << lw) < P) lw++;
    const uint32_t lr1 = log_n1 - lw, lr2 = log_n2 - lw;
    const uint64_t per_vec = r1 * n2;
    // Round 5: the inter-step twiddle and the block order of the exchange ride on the column step's LAST pass (NttRowPost) -- the
    // packing pass below (two products, a read and a write of the whole stack, a launch) is left for the one shape without a
    // column step (n2 == 1)
    const NttRowPost post{cm.d_send, lo, hi, (uint32_t)h, lr1, lr2, (uint32_t)k, row0};
    const bool fused_pack = log_n2 >= 1;
    // x[t] *= w_2n^t (odd), then the column step.  Round 4: the factors are applied by the column step's own first load (row b of
    // the batch knows its t: NttRowCoset) -- the separate pass over the stack (saturated field, 64-bit divisions: 0.33 ms for two
    // 2^20 vectors in rocprofv3, more than the pack and unpack passes together) is kept for blocks the batch cannot describe
    if (odd && log_n2 >= 1 && (r1 & (r1 - 1)) == 0) {
        NttRowCoset pre;
        if ((rc = ntt_coset_tables_kernel_format((int)log_n, &pre.lo, &pre.hi, &pre.hc, s))) return rc;
        pre.shift = log_n1; pre.row0 = (uint32_t)row0; pre.row_mask = (uint32_t)(r1 - 1);
        if ((rc = ntt_run(L, x, nullptr, x, nullptr, n2, 0, inverse, s, k * r1, &pre, fused_pack ? &post : nullptr))) return rc;
    } else {
        if (odd && (rc = dist_scale_dev(x, k, r1, n2, row0, log_n1, log_n, 1, 0, s))) return rc;
        if (log_n2 >= 1 && (rc = ntt_run(L, x, nullptr, x, nullptr, n2, 0, inverse, s, k * r1, nullptr, fused_pack ? &post : nullptr))) return rc;   // column step
    }
    if (!fused_pack) {
        C->timer.begin("dist_pack", s);
        hipLaunchKernelGGL(dist_pack_kernel, dim3(ceil_div_u64(per_vec, 256), (uint32_t)k), dim3(256), 0, s, x, cm.d_send, (uint32_t)k, lr1, lr2, lw, row0, lo, hi, (uint32_t)h);
        C->timer.end(s);
        WS_HIP_CHECK(hipGetLastError());
    }
    const Fe* recv = cm.d_send;                                                                        // a world of one: the exchange is the identity
    if (P > 1) {
        if (!cm.all_to_all) { set_last_error("prove_dist: no all-to-all callback"); return WS_ERR_ARG; }
        const int xrc = cm.all_to_all(cm.user, k * r1 * r2 * sizeof(Fe), (void*)s);
        if (exchanges) ++*exchanges;                                                                   // posted (or attempted): the peers' matching call has its partner
        if (xrc != 0) { set_last_error("prove_dist: the all-to-all callback failed"); return WS_ERR_ARG; }
        recv = cm.d_recv;
    }
    // ... and the transposition rides on the row step's FIRST load (NttRowGather); the unpacking pass is left for n1 == 1
    if (log_n1 >= 1) {
        const NttRowGather gather{recv, lr1, lr2, (uint32_t)k};
        return ntt_run(L, recv, nullptr, y, nullptr, n1, 0, inverse, s, k * r2, nullptr, nullptr, &gather);   // row step
    }
    C->timer.begin("dist_unpack", s);
    hipLaunchKernelGGL(dist_unpack_kernel, dim3(ceil_div_u64(per_vec, 256), (uint32_t)k), dim3(256), 0, s, recv, y, (uint32_t)k, lr1, lr2, lw);
    C->timer.end(s);
    WS_HIP_CHECK(hipGetLastError());
    return WS_OK;
}

// Everything of calc_h_dist that can fail on ONE rank without a collective being involved: the geometry checks and the
// grow-only reserves of the lane's buffers.  groth16_prove_dist calls it before its first collective and ships the result.
int calc_h_dist_reserve(Lane& L, const DistComm& cm, uint32_t n_signals, uint32_t domain, uint32_t l2_expected) {
    if (domain < 2 || (domain & (domain - 1)) || domain > (1u << 27)) return WS_ERR_SIZE;
    uint32_t log_n = 0;
    while ((1u << log_n) < domain) log_n++;
    const uint32_t l1 = (log_n + 1) / 2, l2 = log_n - l1, P = cm.world;
    if ((P & (P - 1)) || ((uint64_t)1 << l2) < P) { set_last_error("prove_dist: needs a power-of-two world size <= 2^floor(log2(domain)/2)"); return WS_ERR_SIZE; }
    if (l2 != l2_expected) { set_last_error("prove_dist: the handle's hExps interleave does not match this domain (load the shard with h_interleave_log = floor(log2(domain) / 2))"); return WS_ERR_ARG; }
    const uint64_t n_loc = domain / P;
    if ((uint64_t)3 * n_loc * sizeof(Fe) > cm.buf_bytes) { set_last_error("prove_dist: exchange buffers too small (3 * 32 * domain / world bytes each)"); return WS_ERR_SIZE; }
    (void)n_signals;
    WS_HIP_CHECK(L.dist_buf[0].reserve((size_t)3 * n_loc * sizeof(Fe)));
    WS_HIP_CHECK(L.dist_buf[1].reserve((size_t)3 * n_loc * sizeof(Fe)));
    return WS_OK;
}

// CALC_H (src/bn128.js:139-164) for the rank's slice: h[(r, j)] = h[(rank rows + r) + 2^l2 j] in plain form
int calc_h_dist(Lane& L, const DistComm& cm, const Fe* d_signals_plain, uint32_t n_signals, const CsrMatrix& A, const CsrMatrix& B,
                uint32_t domain, uint32_t l2_expected, Fe* d_h_local, hipStream_t s, int* exchanges) {
    Context* C = ctx();
    if (!C) return WS_ERR_NOINIT;
    if (A.n_rows != domain || B.n_rows != domain || A.n_cols != n_signals || B.n_cols != n_signals) return WS_ERR_ARG;
    int rc0 = calc_h_dist_reserve(L, cm, n_signals, domain, l2_expected);      // (no-ops after the caller's preflight)
    if (rc0) return rc0;
    uint32_t log_n = 0;
    while ((1u << log_n) < domain) log_n++;
    const uint32_t l1 = (log_n + 1) / 2, l2 = log_n - l1, P = cm.world;
    const uint64_t n_loc = domain / P;
    ScratchGuard scratch_turn(L.calch_chain, s);
    Fe* X = L.dist_buf[0].as<Fe>();
    Fe* T = L.dist_buf[1].as<Fe>();
    KernelTimer& Tm = C->timer;
    int rc;
    const Fe* sigM = d_signals_plain;          // (plain signals against pre-scaled coefficients: calch.hip, pols_to_csr)
    // a, b: only the rows of this rank's slice, in the l1-interleaved layout (rows i1 in the rank's range, all i2)
    const uint64_t r1 = ((uint64_t)1 << l1) / P, c1 = (uint64_t)1 << (log_n - l1);
    uint32_t log_r1 = 0;
    while (((uint64_t)1 << log_r1) < r1) log_r1++;
    Tm.begin("lc_spmv", s);
    hipLaunchKernelGGL(lc_spmv_rows_kernel, dim3(ceil_div_u64(n_loc, 256)), dim3(256), 0, s, A.row_ptr.as<uint32_t>(), A.col.as<uint32_t>(),
                       A.coef.as<Fe>(), sigM, log_r1, c1, (uint64_t)cm.rank * r1, l1, X);
    hipLaunchKernelGGL(lc_spmv_rows_kernel, dim3(ceil_div_u64(n_loc, 256)), dim3(256), 0, s, B.row_ptr.as<uint32_t>(), B.col.as<uint32_t>(),
                       B.coef.as<Fe>(), sigM, log_r1, c1, (uint64_t)cm.rank * r1, l1, X + n_loc);
    Tm.end(s);
    WS_HIP_CHECK(hipGetLastError());
    if ((rc = fr_mul_dev(X, X + n_loc, X + 2 * n_loc, n_loc, s))) return rc;                           // E = A.B on the domain
    if ((rc = dist_ntt_native(L, cm, X, T, log_n, l1, 0, 1, 3, s, exchanges))) return rc;                         // coefficients of a, b; e   (l2-interleaved)
    if ((rc = dist_ntt_native(L, cm, T, X, log_n, l2, 1, 0, 2, s, exchanges))) return rc;                         // odd-coset evaluations     (l1-interleaved)
    if ((rc = fr_mul_dev(X, X + n_loc, X, n_loc, s))) return rc;                                       // O = A.B on the coset
    if ((rc = dist_ntt_native(L, cm, X, T, log_n, l1, 0, 1, 1, s, exchanges))) return rc;                         // o                          (l2-interleaved, like e)
    const uint64_t rows = ((uint64_t)1 << l2) / P;
    return dist_combine_dev(T + 2 * n_loc, T, d_h_local, rows, (uint64_t)1 << (log_n - l2), (uint64_t)cm.rank * rows, l2, log_n, s);
}

WS_DEFINE_WARM(dist)

}  // namespace wsnark
