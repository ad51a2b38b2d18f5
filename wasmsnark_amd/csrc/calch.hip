// calch.hip -- the CALC_H worker command on the GPU: h = upper half of the coefficients of
// A(x)*B(x), where A, B are the witness-weighted Lagrange combinations of the key's polsA/polsB.
//
// Replaces SURVEY.md section 8a rows a17-a19 (/root/reference src/bn128.js:126-166 CALC_H;
// src/build_pol.js:62-144 pol_constructLC; src/build_fft.js:374-547 copyNInterleaved / mulN /
// to/fromMontgomeryN).
//
// Differences from the reference's sequence (all exact field arithmetic => identical h):
//  * pol_constructLC walks a column-major record stream and scatter-adds; here the stream is
//    transposed ONCE at key load into row-major CSR and each output row is one lane's dot product
//    (modular sums are order independent).
//  * the reference interleaves evaluations on the n-domain and on its odd coset into a 2n array
//    and runs one size-2n inverse transform, keeping the upper half.  With E = A.B on the domain,
//    O = A.B on the coset, e = iNTT_n(E), o = iNTT_n(O):
//        c[t] = (e[t mod n] + w_2n^-t o[t mod n]) / 2   for t in [0, 2n)
//    so h[t] = c[n + t] = (e[t] - w_2n^-t o[t]) / 2: two size-n transforms, no 2n buffer.
#include <string.h>

#include <atomic>
#include <thread>

#include "internal.h"

namespace wsnark {

__global__ __launch_bounds__(256) void fr_map_kernel(const Fe* __restrict__ in, Fe* __restrict__ out, uint64_t n, int to_mont) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = to_mont ? Fr::to_mont(in[i]) : Fr::from_mont(in[i]);
}

// res[row] = sum_k coef[k] * sig[col[k]]   (pol_constructLC, row-major), for the two matrices of a proof in ONE launch (blockIdx.y
// selects A or B): the rows are gather-latency bound (1-3 terms per lane), so twice the lanes in flight take about as long as one
// matrix alone -- and CALC_H's transforms start one launch earlier, in the stretch of a proof where the grouping pass leaves most of the chip idle
struct SpmvPair { const uint32_t* row_ptr[2]; const uint32_t* col[2]; const Fe* coef[2]; Fe* res[2]; };
__global__ __launch_bounds__(256) void lc_spmv2_kernel(SpmvPair M, const Fe* __restrict__ sig, uint32_t n_rows) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const uint32_t* __restrict__ row_ptr = M.row_ptr[blockIdx.y];
    const uint32_t* __restrict__ col = M.col[blockIdx.y];
    const Fe* __restrict__ coef = M.coef[blockIdx.y];
    M.res[blockIdx.y][r] = lc_row_dot(coef, col, sig, row_ptr[r], row_ptr[r + 1]);
}

__global__ __launch_bounds__(256) void fr_mul_kernel(const Fe* __restrict__ a, const Fe* __restrict__ b, Fe* __restrict__ out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = Fr::mul(a[i], b[i]);
}

// ---- pieces of the distributed (four-step) transform: n = n1 * n2, element (r, c) of a rank's rows x cols block sits at
// global position t = (row0 + r) + n1 * c of the length-n vector (wasmsnark_amd/dist.py: dist_ntt) ----
//   mode 0: *= w_n^((row0 + r) * c)   the twiddle between the column step and the row step (inverse: w_n^-1)
//   mode 1: *= w_2n^t                 the coset pre-scale of fft_fft's odd = 1 (src/build_fft.js:159-187)
__global__ __launch_bounds__(256) void dist_scale_kernel(Fe* __restrict__ data, uint64_t stack, uint64_t rows, uint64_t cols, uint64_t row0,
                                                           uint32_t log_n1, int mode, const Fe* __restrict__ lo, const Fe* __restrict__ hi, uint32_t h) {
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= stack * rows * cols) return;
    const uint64_t rr = idx / cols, c = idx - rr * cols, r = rr % rows;      // `stack` blocks of the same layout, one after the other
    const uint64_t e = mode == 0 ? (row0 + r) * c : (row0 + r) + (c << log_n1);
    const Fe f = Fr::mul(hi[e >> h], lo[e & (((uint64_t)1 << h) - 1)]);
    data[idx] = Fr::mul(data[idx], f);
}
int dist_scale_dev(Fe* d_data, uint64_t stack, uint64_t rows, uint64_t cols, uint64_t row0, uint32_t log_n1, uint32_t log_n, int mode, int inverse, hipStream_t s) {
    Context* C = ctx();
    if (!C) return WS_ERR_NOINIT;
    if (!s) s = C->stream;
    if (stack == 0 || rows == 0 || cols == 0) return WS_OK;
    if (stack > 64) return WS_ERR_SIZE;
    if (!d_data) return WS_ERR_ARG;
    if (log_n < 1 || log_n > 27 || log_n1 > log_n || cols != ((uint64_t)1 << (log_n - log_n1)) || row0 + rows > ((uint64_t)1 << log_n1) ||
        (mode != 0 && mode != 1))
        return WS_ERR_SIZE;
    const Fe *lo, *hi;
    int h;
    int rc;
    if (mode == 0) {
        rc = ntt_twiddle_tables((int)log_n, inverse, &lo, &hi, &h, s);
    } else {
        Fe n_inv;
        rc = ntt_coset_tables((int)log_n, &lo, &hi, &h, &n_inv, s);
    }
    if (rc) return rc;
    hipLaunchKernelGGL(dist_scale_kernel, dim3(ceil_div_u64(stack * rows * cols, 256)), dim3(256), 0, s, d_data, stack, rows, cols, row0, log_n1, mode, lo, hi, (uint32_t)h);
    WS_HIP_CHECK(hipGetLastError());
    return WS_OK;
}

// the last step of CALC_H on a rank's block of the interleaved layout: h[t] = fromMontgomery((e[t] - w_2n^-t o[t]) / 2),
// t = (row0 + r) + n1 * c   (the formula of the single-GPU epilogue, ntt.hip: CombineEpilogue)
__global__ __launch_bounds__(256) void dist_combine_kernel(const Fe* __restrict__ e, const Fe* __restrict__ o, Fe* __restrict__ h,
                                                             uint64_t rows, uint64_t cols, uint64_t row0, uint32_t log_n1, uint32_t log_n,
                                                             const Fe* __restrict__ cs_lo, const Fe* __restrict__ cs_hi, uint32_t hc, Fe half) {
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * cols) return;
    const uint64_t r = idx / cols, c = idx - r * cols;
    const uint64_t t = (row0 + r) + (c << log_n1);
    Fe v;
    if (t == 0) {
        v = Fr::sub(e[idx], o[idx]);
    } else {
        const uint64_t x = ((uint64_t)1 << log_n) - t;
        v = Fr::add(e[idx], Fr::mul(Fr::mul(cs_hi[x >> hc], cs_lo[x & (((uint64_t)1 << hc) - 1)]), o[idx]));
    }
    h[idx] = Fr::from_mont(Fr::mul(v, half));
}
int dist_combine_dev(const Fe* d_e, const Fe* d_o, Fe* d_h, uint64_t rows, uint64_t cols, uint64_t row0, uint32_t log_n1, uint32_t log_n, hipStream_t s) {
    Context* C = ctx();
    if (!C) return WS_ERR_NOINIT;
    if (!s) s = C->stream;
    if (rows == 0 || cols == 0) return WS_OK;
    if (!d_e || !d_o || !d_h) return WS_ERR_ARG;
    if (log_n < 1 || log_n > 27 || log_n1 > log_n || cols != ((uint64_t)1 << (log_n - log_n1)) || row0 + rows > ((uint64_t)1 << log_n1)) return WS_ERR_SIZE;
    const Fe *lo, *hi;
    int hc;
    Fe n_inv;
    int rc = ntt_coset_tables((int)log_n, &lo, &hi, &hc, &n_inv, s);
    if (rc) return rc;
    const Fe half = Fr::inv(Fr::add(Fr::one(), Fr::one()));
    hipLaunchKernelGGL(dist_combine_kernel, dim3(ceil_div_u64(rows * cols, 256)), dim3(256), 0, s, d_e, d_o, d_h, rows, cols, row0, log_n1, log_n,
                       lo, hi, (uint32_t)hc, half);
    WS_HIP_CHECK(hipGetLastError());
    return WS_OK;
}
int fr_mul_dev(const Fe* d_a, const Fe* d_b, Fe* d_out, uint64_t n, hipStream_t s) {
    Context* C = ctx();
    if (!C) return WS_ERR_NOINIT;
    if (n == 0) return WS_OK;
    if (!d_a || !d_b || !d_out) return WS_ERR_ARG;
    if (!s) s = C->stream;
    hipLaunchKernelGGL(fr_mul_kernel, dim3(ceil_div_u64(n, 256)), dim3(256), 0, s, d_a, d_b, d_out, n);
    WS_HIP_CHECK(hipGetLastError());
    return WS_OK;
}
// a = A w, b = B w (pol_constructLC twice, bn128.js:139-145) from a device-resident plain witness; Montgomery outputs
int eval_ab_dev(Lane& L, const Fe* d_signals_plain, uint32_t n_signals, const CsrMatrix& A, const CsrMatrix& B, uint32_t domain,
                Fe* d_a, Fe* d_b, hipStream_t s) {
    Context* C = ctx();
    if (!C) return WS_ERR_NOINIT;
    if (!s) s = L.stream;
    if (A.n_rows != domain || B.n_rows != domain || A.n_cols != n_signals || B.n_cols != n_signals || !d_a || !d_b) return WS_ERR_ARG;
    (void)L;
    const dim3 blk(256), grd(ceil_div_u64(domain, 256));
    // (the resident coefficients are pre-scaled by R: plain signals in, Montgomery sums out -- see pols_to_csr)
    const SpmvPair M{{A.row_ptr.as<uint32_t>(), B.row_ptr.as<uint32_t>()}, {A.col.as<uint32_t>(), B.col.as<uint32_t>()}, {A.coef.as<Fe>(), B.coef.as<Fe>()}, {d_a, d_b}};
    hipLaunchKernelGGL(lc_spmv2_kernel, dim3(grd.x, 2), blk, 0, s, M, d_signals_plain, domain);
    WS_HIP_CHECK(hipGetLastError());
    return WS_OK;
}

int fr_map_dev(const Fe* d_in, Fe* d_out, uint64_t n, int to_mont, hipStream_t s) {
    Context* C = ctx();
    if (!C) return WS_ERR_NOINIT;
    if (n == 0) return WS_OK;
    if (!s) s = C->stream;
    hipLaunchKernelGGL(fr_map_kernel, dim3(ceil_div_u64(n, 256)), dim3(256), 0, s, d_in, d_out, n, to_mont);
    WS_HIP_CHECK(hipGetLastError());
    return WS_OK;
}

// ---- transposition of the record stream into row-major CSR, on the GPU (round 4) ----
// The records are variable-length, so ONE sequential walk over the per-signal headers stays on the host (where every signal
// starts, how many records precede it: ~2 ms per million signals, lengths validated on the way).  Everything else -- 36 bytes
// per non-zero -- is uploaded as it is and transposed by three kernels: count per row (one lane per record, the signal found
// by binary search in the record prefix), exclusive scan, fill through per-row cursors with the coefficient scaled by R on the
// way (see below).  Round 3 did the two passes over the records on host threads: 56-65 ms per 2^20 key against ~10 ms of
// upload + kernels here.  The order of the terms inside a row is whatever the atomics give: the modular sum is exact.
__global__ __launch_bounds__(256) void csr_count_kernel(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ start,
                                                          const uint32_t* __restrict__ rec_base, uint32_t n_signals, uint32_t nnz, uint32_t domain,
                                                          uint32_t* __restrict__ sig_of, uint32_t* __restrict__ row_of, uint32_t* __restrict__ row_cnt,
                                                          uint32_t* __restrict__ bad) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nnz) return;
    uint32_t lo = 0, hi = n_signals;                       // largest i with rec_base[i] <= t  (rec_base[n_signals] = nnz > t)
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (rec_base[mid] <= t) lo = mid; else hi = mid; }
    const uint64_t q = start[lo] + 4 + (uint64_t)36 * (t - rec_base[lo]);
    const uint32_t idx = *reinterpret_cast<const uint32_t*>(blob + q);
    sig_of[t] = lo;
    row_of[t] = idx;
    if (idx >= domain) { atomicAdd(bad, 1u); return; }
    atomicAdd(&row_cnt[idx], 1u);
}
// exclusive scan of n counters in three steps (tiles of 1024): tile sums, scan of the tile sums by one workgroup, tile scans
__global__ __launch_bounds__(256) void scan_tile_sums(const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ tile_sum) {
    __shared__ uint32_t red[256];
    const uint32_t base = blockIdx.x * 1024;
    uint32_t v = 0;
    for (uint32_t k = threadIdx.x; k < 1024; k += 256) if (base + k < n) v += in[base + k];
    red[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t d = 128; d >= 1; d >>= 1) { if (threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d]; __syncthreads(); }
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void scan_tile_bases(uint32_t* __restrict__ tile_sum, uint32_t ntiles) {
    __shared__ uint32_t part[256];
    const uint32_t per = (ntiles + 255) / 256, lo = threadIdx.x * per;
    uint32_t v = 0;
    for (uint32_t k = lo; k < lo + per && k < ntiles; k++) v += tile_sum[k];
    part[threadIdx.x] = v;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t run = 0; for (uint32_t k = 0; k < 256; k++) { const uint32_t x = part[k]; part[k] = run; run += x; } }
    __syncthreads();
    uint32_t run = part[threadIdx.x];
    for (uint32_t k = lo; k < lo + per && k < ntiles; k++) { const uint32_t x = tile_sum[k]; tile_sum[k] = run; run += x; }
}
__global__ __launch_bounds__(256) void scan_tiles(const uint32_t* __restrict__ in, uint32_t n, const uint32_t* __restrict__ tile_base,
                                                    uint32_t* __restrict__ out, uint32_t* __restrict__ cursor) {
    __shared__ uint32_t part[256];
    const uint32_t base = blockIdx.x * 1024, t0 = base + threadIdx.x * 4;
    uint32_t v[4], sum = 0;
    for (uint32_t k = 0; k < 4; k++) { v[k] = t0 + k < n ? in[t0 + k] : 0; sum += v[k]; }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {
        const uint32_t x = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += x;
        __syncthreads();
    }
    uint32_t run = tile_base[blockIdx.x] + part[threadIdx.x] - sum;
    for (uint32_t k = 0; k < 4; k++)
        if (t0 + k < n) { out[t0 + k] = run; cursor[t0 + k] = run; run += v[k]; }
}
__global__ __launch_bounds__(256) void csr_fill_kernel(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ start,
                                                         const uint32_t* __restrict__ rec_base, uint32_t nnz, const uint32_t* __restrict__ sig_of,
                                                         const uint32_t* __restrict__ row_of, uint32_t* __restrict__ cursor,
                                                         uint32_t* __restrict__ col, Fe* __restrict__ coef) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nnz) return;
    const uint32_t i = sig_of[t];
    const uint32_t* w = reinterpret_cast<const uint32_t*>(blob + start[i] + 8 + (uint64_t)36 * (t - rec_base[i]));   // (4-byte aligned)
    Fe c;
    for (int k = 0; k < 4; k++) c.l[k] = (uint64_t)w[2 * k] | ((uint64_t)w[2 * k + 1] << 32);
    const uint32_t k = atomicAdd(&cursor[row_of[t]], 1u);
    col[k] = i;
    // The resident coefficients carry one more factor R (c R^2 instead of the file's Montgomery form c R): the product with a
    // PLAIN signal s then is c R^2 s / R = (c s) R, the Montgomery form of the term.  The reference converts the nSignals
    // signals of every proof instead (fft_toMontgomeryN, src/bn128.js:139); scaling the key's coefficients once at load time
    // removes that pass -- and its buffer -- from every proof, and the sparse product reads the caller's witness as it is
    // (raw 256-bit values: the product of a reduced operand and any value below 2^256 reduces correctly).
    coef[k] = Fr::to_mont(c);
}

struct SlabPiece {      // a part of one device allocation, with DevBuf's accessor
    void* p = nullptr;
    template <class T> T* as() const { return (T*)p; }
};

int pols_to_csr(const uint8_t* pols, size_t len, uint32_t n_signals, uint32_t domain, CsrMatrix* out,
                size_t* consumed, hipStream_t s, void (*release)(const void* p, size_t n)) {
    const bool trace = tuning_get("TRACE", 0) == 1;
    const auto t_in = std::chrono::steady_clock::now();
    auto at = [&](const char* what) {
        if (trace) fprintf(stderr, "[wsnark trace] pols_to_csr (%u signals): %s at %.2f ms\n", n_signals, what,
                           std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_in).count());
    };
    // pass 0 (host, sequential): where every signal's records start and how many records precede it; lengths validated
    std::vector<uint64_t> start((size_t)n_signals + 1);
    std::vector<uint32_t> rec_base((size_t)n_signals + 1);
    size_t pp = 0;
    uint64_t nnz = 0;
    // Streaming form (a mapped key file, `release` set): the records go up in pieces of ~32 MiB AS the walk passes them and every
    // piece's pages are handed back at once, so the walk and the upload together never hold more than one piece of the file --
    // instead of all of it twice (walk, then upload).  The device copy of the records then needs its own allocation, sized by the
    // section's length, before the walk knows the record count.
    DevBuf d_blob_own;
    size_t sent = 0;
    const size_t piece = (size_t)tuning_get("POLS_PIECE_KB", 32768) << 10;       // (tests make the pieces small)
    if (release) WS_HIP_CHECK(d_blob_own.alloc(len + 64));
    for (uint32_t i = 0; i < n_signals; i++) {
        start[i] = pp;
        rec_base[i] = (uint32_t)nnz;
        if (pp + 4 > len) { set_last_error("pols: truncated record header"); return WS_ERR_FORMAT; }
        uint32_t nc; memcpy(&nc, pols + pp, 4); pp += 4;
        if ((uint64_t)nc * 36 > len - pp) { set_last_error("pols: truncated coefficient records"); return WS_ERR_FORMAT; }
        pp += (size_t)nc * 36;
        nnz += nc;
        if (nnz >= ((uint64_t)1 << 32)) return WS_ERR_SIZE;
        if (release && pp - sent >= piece) {
            const int rc_up = upload_staged((uint8_t*)d_blob_own.p + sent, pols + sent, pp - sent, s);
            if (rc_up) return rc_up;
            release(pols + sent, pp - sent);
            sent = pp;
        }
    }
    if (release && pp > sent) {
        const int rc_up = upload_staged((uint8_t*)d_blob_own.p + sent, pols + sent, pp - sent, s);
        if (rc_up) return rc_up;
        release(pols + sent, pp - sent);
        sent = pp;
    }
    start[n_signals] = pp;
    rec_base[n_signals] = (uint32_t)nnz;
    if (consumed) *consumed = pp;
    const size_t nz = (size_t)nnz ? (size_t)nnz : 1;
    out->n_rows = domain; out->n_cols = n_signals; out->nnz = nnz;
    WS_HIP_CHECK(out->row_ptr.alloc(((size_t)domain + 1) * 4));
    WS_HIP_CHECK(out->col.alloc(nz * 4));
    WS_HIP_CHECK(out->coef.alloc(nz * sizeof(Fe)));
    const uint32_t ntiles = ceil_div_u64((uint64_t)domain + 1, 1024);
    // the nine temporaries in ONE allocation (every hipMalloc / hipFree is a call into the driver -- and a free waits for the device:
    // with the ROCm 7.2 runtime the 2 x 12 allocations and 2 x 9 frees of the two matrices were half of this function's 30 ms).
    // Plain hipMalloc / hipFree on purpose: see prove.hip on the stream-ordered allocator
    SlabPiece d_blob, d_start, d_base, d_sig, d_row, d_cnt, d_cursor, d_tiles, d_bad;
    DevBuf d_tmp;
    {
        const size_t sizes[9] = {release ? 64 : pp + 64, start.size() * 8, rec_base.size() * 4, nz * 4, nz * 4, ((size_t)domain + 1) * 4, ((size_t)domain + 1) * 4, (size_t)ntiles * 4, 4};
        SlabPiece* const pieces[9] = {&d_blob, &d_start, &d_base, &d_sig, &d_row, &d_cnt, &d_cursor, &d_tiles, &d_bad};
        size_t total = 0;
        for (size_t b : sizes) total += (b + 255) & ~(size_t)255;
        WS_HIP_CHECK(d_tmp.alloc(total));
        size_t off = 0;
        for (int i = 0; i < 9; i++) { pieces[i]->p = (uint8_t*)d_tmp.p + off; off += (sizes[i] + 255) & ~(size_t)255; }
        if (release) d_blob.p = d_blob_own.p;
    }
    at("header walk + allocations done");
    int rc;
    if (!release && (rc = upload_staged(d_blob.p, pols, pp, s))) return rc;
    if ((rc = upload_staged(d_start.p, start.data(), start.size() * 8, s))) return rc;
    if ((rc = upload_staged(d_base.p, rec_base.data(), rec_base.size() * 4, s))) return rc;
    WS_HIP_CHECK(hipMemsetAsync(d_cnt.p, 0, ((size_t)domain + 1) * 4, s));
    WS_HIP_CHECK(hipMemsetAsync(d_bad.p, 0, 4, s));
    if (nnz)
        hipLaunchKernelGGL(csr_count_kernel, dim3(ceil_div_u64(nnz, 256)), dim3(256), 0, s, d_blob.as<uint8_t>(), d_start.as<uint64_t>(), d_base.as<uint32_t>(),
                           n_signals, (uint32_t)nnz, domain, d_sig.as<uint32_t>(), d_row.as<uint32_t>(), d_cnt.as<uint32_t>(), d_bad.as<uint32_t>());
    // (domain + 1 counters, the last one zero: the exclusive scan's last entry is the total, i.e. row_ptr[domain])
    hipLaunchKernelGGL(scan_tile_sums, dim3(ntiles), dim3(256), 0, s, d_cnt.as<uint32_t>(), domain + 1, d_tiles.as<uint32_t>());
    hipLaunchKernelGGL(scan_tile_bases, dim3(1), dim3(256), 0, s, d_tiles.as<uint32_t>(), ntiles);
    hipLaunchKernelGGL(scan_tiles, dim3(ntiles), dim3(256), 0, s, d_cnt.as<uint32_t>(), domain + 1, d_tiles.as<uint32_t>(), out->row_ptr.as<uint32_t>(), d_cursor.as<uint32_t>());
    WS_HIP_CHECK(hipGetLastError());
    uint32_t bad = 0;
    WS_HIP_CHECK(hipMemcpyAsync(&bad, d_bad.p, 4, hipMemcpyDeviceToHost, s));
    WS_HIP_CHECK(hipStreamSynchronize(s));
    at("records uploaded, counted, scanned");
    if (bad) { set_last_error("pols: coefficient index >= domainSize"); return WS_ERR_FORMAT; }
    if (nnz)
        hipLaunchKernelGGL(csr_fill_kernel, dim3(ceil_div_u64(nnz, 256)), dim3(256), 0, s, d_blob.as<uint8_t>(), d_start.as<uint64_t>(), d_base.as<uint32_t>(),
                           (uint32_t)nnz, d_sig.as<uint32_t>(), d_row.as<uint32_t>(), d_cursor.as<uint32_t>(), out->col.as<uint32_t>(), out->coef.as<Fe>());
    WS_HIP_CHECK(hipGetLastError());
    WS_HIP_CHECK(hipStreamSynchronize(s));
    at("filled");
    return WS_OK;
}

int calc_h_dev(Lane& L, const Fe* d_signals_plain, uint32_t n_signals, const CsrMatrix& A, const CsrMatrix& B, uint32_t domain,
               Fe* d_h_out, hipStream_t s) {
    Context* C = ctx();
    if (!C) return WS_ERR_NOINIT;
    if (!s) s = L.stream;
    // src/build_fft.js:137-154: the transforms trap unless the size is a power of two; the 2n-sized
    // inverse of the reference needs 2*domain <= 2^28
    if (domain < 2 || (domain & (domain - 1)) || domain > (1u << 27)) return WS_ERR_SIZE;
    if (A.n_rows != domain || B.n_rows != domain || A.n_cols != n_signals || B.n_cols != n_signals) return WS_ERR_ARG;
    int bits = 0;
    while ((1u << bits) < domain) bits++;
    const size_t nb = (size_t)domain * sizeof(Fe);
    ScratchGuard scratch_turn(L.calch_chain, s);   // the work arrays below are shared by every CALC_H on this lane
    // a and b back to back: their two inverse and their two coset transforms run as ONE batch of two each (round 6: a 2^20 pass
    // is exactly one resident set of workgroups, all in the same phase; twice the workgroups let loads overlap butterflies)
    WS_HIP_CHECK(L.calch_buf[1].reserve(2 * nb));
    WS_HIP_CHECK(L.calch_buf[3].reserve(nb));
    // the transforms' ping-pong buffer at the size of the LARGEST batch below, before the first transform takes it at half that: a
    // buffer that grows is freed first, and hipFree waits for the whole device -- in a key's first proof that is the background
    // table build (first proof 156 ms instead of 17: profiles/r06_s1 against r06_s2)
    if (tuning_get("CALCH_BATCH", 1)) WS_HIP_CHECK(L.ntt_scratch.reserve(2 * nb));
    Fe* a = L.calch_buf[1].as<Fe>();
    Fe* b = a + domain;
    Fe* e = L.calch_buf[3].as<Fe>();
    KernelTimer& T = C->timer;
    const dim3 blk(256), grd(ceil_div_u64(domain, 256));
    int rc;

    // bn128.js:139 (fft_toMontgomeryN of the signals) has no counterpart per proof: the key's coefficients were scaled once
    // at load time (pols_to_csr), so the sparse products read the plain witness and still leave Montgomery sums
    T.begin("lc_spmv", s);                                           // bn128.js:141-145
    const SpmvPair M{{A.row_ptr.as<uint32_t>(), B.row_ptr.as<uint32_t>()}, {A.col.as<uint32_t>(), B.col.as<uint32_t>()}, {A.coef.as<Fe>(), B.coef.as<Fe>()}, {a, b}};
    hipLaunchKernelGGL(lc_spmv2_kernel, dim3(grd.x, 2), blk, 0, s, M, d_signals_plain, domain);
    T.end(s);
    // The pointwise products and the final combination are fused into the transforms next to them: E = A.B is formed by the FIRST pass of its inverse transform while it loads,
    // O = A.B on the coset likewise, and the last pass of that second transform stores h directly.  Saves three
    // element-wise kernels (4 x 32 B x domain of traffic each) and their slow saturated-field products.
    {
        if ((rc = ntt_run(L, a, b, e, nullptr, domain, 0, 1, s))) return rc;   // e = iNTT(A.B)            (bn128.js:148, 160)
        if (tuning_get("CALCH_BATCH", 1)) {
            if ((rc = ntt_dev(L, a, domain, 0, 1, s, 2))) return rc;           // bn128.js:150-151  evaluations -> coefficients (a, b)
            if ((rc = ntt_dev(L, a, domain, 1, 0, s, 2))) return rc;           // bn128.js:152-153  -> odd-coset evaluations (a, b)
        } else {
            if ((rc = ntt_dev(L, a, domain, 0, 1, s))) return rc;
            if ((rc = ntt_dev(L, b, domain, 0, 1, s))) return rc;
            if ((rc = ntt_dev(L, a, domain, 1, 0, s))) return rc;
            if ((rc = ntt_dev(L, b, domain, 1, 0, s))) return rc;
        }
        return ntt_run(L, a, b, d_h_out, e, domain, 0, 1, s);                  // o = iNTT(A.B on the coset), h = combine(e, o)   (:158-164)
    }
}

WS_DEFINE_WARM(calch)

}  // namespace wsnark
