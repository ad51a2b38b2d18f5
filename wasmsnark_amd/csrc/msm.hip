// msm.hip -- bucket (Pippenger) multi-scalar multiplication on BN128 G1 / G2 for gfx950.
//
// Replaces SURVEY.md section 8a rows a11-a14 and the sharding of a20
// (/root/reference src/build_multiexp.js: __packbits :30-155, subset table :295-429,
// g1m_multiexp2 :651-744, g2m_multiexp :498-580; host split/gather src/bn128.js:353-415).
// The reference's method (w=7 bit-sliced subset tables, 256 per-bit accumulators) is NOT
// reproduced: any correct MSM yields the same group element, and results are compared
// after affine normalisation (src/bn128.js:706-712).
//
// Pipeline (all device-resident; a plan depends on the scalars only and is shared by every point set that
// uses the same scalars -- the prover's A, B1, B2 and C sums):
//   plan  1 presort_count/scan/scatter  scalar (32 B coalesced) -> reduced mod r -> signed c-bit digits on the
//                      fly -> ONE scatter of (index | sign | low bucket bits) entries into coarse bins
//                      (window x top bucket bits), block-reserved ranges, ranks from LDS atomics
//         2 presort_bins   one workgroup per bin: LDS counting sort over the low bucket bits, bucket bounds,
//                      task-length histogram (wavefront-aggregated counters for hot buckets)
//         3 msm_plan_*     buckets cut into tasks of <= lmax entries, ordered longest-first (255-level
//                      counting sort); very hot buckets get their task list from a workgroup each
//   exec  4 msm_accumulate  one lane per task: mixed additions (XYZZ += affine, 8M+2S) of the task's points,
//                      gathered 64/128 B per lane, next point in flight during the current addition
//         5 msm_combine_*   partial sums of split buckets: one lane, one wavefront (LDS tree), or -- very hot
//                      buckets -- one wavefront per 512 partial sums and a second stage per bucket
//         6 msm_chunks      one lane per 8 buckets: S_j = sum B, A_j = sum (i-i0+1) B
//         7 msm_tree        per (window, bit q): LDS tree sums U_q = sum_{j: bit q} S_j, and sum A_j
//         8 host            sum_w 2^(c w) (A_w + 8 sum_q 2^q U_{w,q}): ~W*(log J + 1) points, a serial
//                      doubling chain that is faster on one CPU core than on one GPU lane.
#include <string.h>

#include <algorithm>
#include <chrono>

#include "internal.h"
namespace wsnark {

#ifndef WS_ACC_WAVES
#define WS_ACC_WAVES 3
#endif
#ifndef WS_ACC_WAVES_G2
#define WS_ACC_WAVES_G2 2
#endif
#ifndef WS_G2_PREFETCH
#define WS_G2_PREFETCH 0
#endif
#ifndef WS_G1_PREFETCH
#define WS_G1_PREFETCH 1   // G1 accumulation: next point's gather in flight during the current addition (0: gather just before use, 16 VGPRs fewer)
#endif
#ifndef WS_MADD_WIDE
#define WS_MADD_WIDE 1     // accumulation loop keeps X wide between additions (curve.h: madd_wide); 0 = strict madd, for A/B builds
#endif
static const uint32_t CHUNK = 8;          // buckets per msm_chunks lane
#ifndef WS_TAIL_L2_DEFAULT
#define WS_TAIL_L2_DEFAULT 4      // second chunk level (msm_chunks2) of full-size table plans: chunk pairs folded per lane (1 = off)
#endif

struct MsmScratch {
    DevBuf vals_out, entries, hot, hot_sums;
    DevBuf bstart, bend, buckets, counters, tasks, multi, partials;
    DevBuf chunkS, chunkA, sums, points_conv;
    DevBuf hot_done;                 // device-scope completion counters of msm_combine_all (zero between launches)
    DevBuf tree_half, tree_done;     // msm_tree: halves of the unmasked rows and their completion counters (zero between launches)
};

// ---------------------------------------------------------------------------
// 1'/2'. bucket grouping without a general sort ("presort", the default).
// Only the grouping matters (order inside a bucket is free), and keys have structure: window (known per
// digit position) x 15-bit bucket.  So: (a) digits are extracted on the fly from the scalars -- no key
// arrays are written; (b) ONE scatter into W*HB coarse bins (bin = window, top bits of the bucket) with
// block-reserved ranges -- no stability needed, so ranks come from LDS atomics; (c) one workgroup per bin
// finishes with an LDS counting sort over the low 7 bucket bits and writes the bucket bounds directly.
// HBM traffic at 2^20: 2 x 32 MB scalars + 128 MB bins written + 128 MB read + 64 MB values written
// (vs 160 MB digits + ~830 MB three-pass radix sort + 64 MB bounds).
// ---------------------------------------------------------------------------
static const uint32_t PRESORT_MAX_BINS = 4096; // W * HB, 16 KiB of LDS counters
// layout of a plan's counter block (u32 words): [0] partial slots, [1] multi-task buckets, [3] total tasks, [4] hot buckets,
// [5] hot slices, [6] tasks of segment 0; task-length histograms of the two task segments at CNT_HIST (2 x 256), their
// rank cursors at CNT_CURSOR (2 x 256), the coarse-bin counts / starts / cursors of the grouping pass from CNT_BINS on
static const uint32_t CNT_HIST = 16, CNT_CURSOR = 16 + 512, CNT_BINS = 2048;

// calls f(k, d, neg) for every owned window k (local index) with a non-zero signed digit d in [1, NB]
template <class Fn>
__device__ __forceinline__ void for_each_digit(const Fe& raw, uint32_t c, uint32_t Wall, uint32_t w_off, uint32_t w_stride, Fn f) {
    const Fe s = Fr::reduce_full(raw);
    const uint32_t NB = 1u << (c - 1);
    uint32_t carry = 0, k = 0;
    for (uint32_t w = 0; w < Wall; w++) {
        const uint32_t bit = w * c;
        const uint32_t limb = bit >> 6, off = bit & 63;
        uint64_t v = limb < 4 ? (s.l[limb] >> off) : 0;
        if (off + c > 64 && limb + 1 < 4) v |= s.l[limb + 1] << (64 - off);
        uint32_t d = (uint32_t)(v & ((1u << c) - 1)) + carry;
        uint32_t neg = 0;
        if (d > NB) { d = (1u << c) - d; neg = 1; carry = 1; } else carry = 0;
        if (w >= w_off && (w - w_off) % w_stride == 0) {
            if (d) f(k, d, neg);
            k++;
        }
    }
}

struct PresortArgs {
    const Fe* scalars;
    uint32_t n, c, Wall, w_off, w_stride;
    uint32_t i0, i_end;                // the scalars [i0, i_end) this launch covers (the digit histogram may be taken chunk by chunk)
    uint32_t lo_bits, HB, nbins;       // bin = k*HB + ((d-1) >> lo_bits)
    uint32_t tile;                     // scalars per workgroup
    uint32_t idx_bits;                 // packed 4-byte entries: idx | neg << idx_bits | lo << (idx_bits+1)
    const uint8_t* mask;               // optional: pairs with mask[i] == 0 are left out (their point is infinity)
    uint32_t flat;                     // fixed-base table plans: ONE bucket set, bin = (d-1) >> lo_bits, entries index the table (window * n + i)
};
__device__ __forceinline__ uint32_t presort_bin(const PresortArgs& A, uint32_t k, uint32_t d) {
    return (A.flat ? 0u : k * A.HB) + ((d - 1) >> A.lo_bits);
}

// 8-byte entry = (low bucket bits << 32) | (index | sign << 31); 4-byte entries when bits(n)+1+lo_bits <= 32
template <class E> struct PresortEntry;
template <> struct PresortEntry<uint64_t> {
    static __device__ __forceinline__ uint64_t make(uint32_t i, uint32_t neg, uint32_t lo, uint32_t) {
        return ((uint64_t)lo << 32) | (uint64_t)(i | (neg << 31));
    }
    static __device__ __forceinline__ uint32_t lo(uint64_t e, uint32_t) { return (uint32_t)(e >> 32); }
    static __device__ __forceinline__ uint32_t val(uint64_t e, uint32_t) { return (uint32_t)e; }
};
template <> struct PresortEntry<uint32_t> {
    static __device__ __forceinline__ uint32_t make(uint32_t i, uint32_t neg, uint32_t lo, uint32_t ib) {
        return i | (neg << ib) | (lo << (ib + 1));
    }
    static __device__ __forceinline__ uint32_t lo(uint32_t e, uint32_t ib) { return e >> (ib + 1); }
    static __device__ __forceinline__ uint32_t val(uint32_t e, uint32_t ib) {
        return (e & ((1u << ib) - 1)) | (((e >> ib) & 1u) << 31);
    }
};

__global__ __launch_bounds__(1024) void presort_count(PresortArgs A, uint32_t* __restrict__ bin_count) {
    __shared__ uint32_t cnt[PRESORT_MAX_BINS];
    for (uint32_t b = threadIdx.x; b < A.nbins; b += blockDim.x) cnt[b] = 0;
    __syncthreads();
    const uint32_t base = A.i0 + blockIdx.x * A.tile;
    for (uint32_t i = base + threadIdx.x; i < base + A.tile && i < A.i_end; i += blockDim.x) {
        if (A.mask && !A.mask[i]) continue;
        for_each_digit(A.scalars[i], A.c, A.Wall, A.w_off, A.w_stride, [&](uint32_t k, uint32_t d, uint32_t) {
            atomicAdd(&cnt[presort_bin(A, k, d)], 1u);
        });
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < A.nbins; b += blockDim.x)
        if (cnt[b]) atomicAdd(&bin_count[b], cnt[b]);
}

// exclusive scan of the (<= 4096) bin counts: bin_start[0..nbins], cursors = starts
__global__ __launch_bounds__(256) void presort_scan(const uint32_t* __restrict__ bin_count, uint32_t nbins,
                                                      uint32_t* __restrict__ bin_start, uint32_t* __restrict__ bin_cursor) {
    __shared__ uint32_t part[256];
    const uint32_t per = (nbins + 255) / 256, lo = threadIdx.x * per;
    uint32_t sum = 0;
    for (uint32_t b = lo; b < lo + per && b < nbins; b++) sum += bin_count[b];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t t = 0; t < 256; t++) { const uint32_t v = part[t]; part[t] = run; run += v; }
        bin_start[nbins] = run;
    }
    __syncthreads();
    uint32_t run = part[threadIdx.x];
    for (uint32_t b = lo; b < lo + per && b < nbins; b++) {
        bin_start[b] = run;
        bin_cursor[b] = run;
        run += bin_count[b];
    }
}

template <class E>
__global__ __launch_bounds__(1024) void presort_scatter(PresortArgs A, uint32_t* __restrict__ bin_cursor, E* __restrict__ entries) {
    __shared__ uint32_t cnt[PRESORT_MAX_BINS];
    __shared__ uint32_t gbase[PRESORT_MAX_BINS];
    for (uint32_t b = threadIdx.x; b < A.nbins; b += blockDim.x) cnt[b] = 0;
    __syncthreads();
    const uint32_t base = A.i0 + blockIdx.x * A.tile;
    for (uint32_t i = base + threadIdx.x; i < base + A.tile && i < A.i_end; i += blockDim.x) {
        if (A.mask && !A.mask[i]) continue;
        for_each_digit(A.scalars[i], A.c, A.Wall, A.w_off, A.w_stride, [&](uint32_t k, uint32_t d, uint32_t) {
            atomicAdd(&cnt[presort_bin(A, k, d)], 1u);
        });
    }
    __syncthreads();
    // reserve this workgroup's range in every bin it touches; then reuse cnt[] as the local rank counters
    for (uint32_t b = threadIdx.x; b < A.nbins; b += blockDim.x) {
        const uint32_t c0 = cnt[b];
        gbase[b] = c0 ? atomicAdd(&bin_cursor[b], c0) : 0;
        cnt[b] = 0;
    }
    __syncthreads();
    const uint32_t lo_mask = (1u << A.lo_bits) - 1;
    for (uint32_t i = base + threadIdx.x; i < base + A.tile && i < A.i_end; i += blockDim.x) {
        if (A.mask && !A.mask[i]) continue;
        for_each_digit(A.scalars[i], A.c, A.Wall, A.w_off, A.w_stride, [&](uint32_t k, uint32_t d, uint32_t neg) {
            const uint32_t b = presort_bin(A, k, d);
            const uint32_t r = atomicAdd(&cnt[b], 1u);
            const uint32_t idx = A.flat ? (A.w_off + k * A.w_stride) * A.n + i : i;
            entries[gbase[b] + r] = PresortEntry<E>::make(idx, neg, (d - 1) & lo_mask, A.idx_bits);
        });
    }
}

// The same scatter for plans of <= 16 windows (every plan of 2^14 pairs and more), one scalar per thread: the digits are taken
// ONCE, statically unrolled -- no run-time indexing of the scalar's limbs (48 bytes of scratch per lane in the kernel above), no
// second extraction -- and the LDS atomic that counts an entry is also the one that ranks it; entries wait in registers for
// the workgroup's bin reservations.  The exclusive scan of the bin counts is taken by every workgroup for itself (4096 counters,
// a few microseconds beside the digit work; one launch and one dependency less than a scan kernel in between); workgroup 0
// leaves the starts behind for presort_bins.  bin_cursor: zero at launch, counts RELATIVE to the bin's start.
static const uint32_t PRESORT_ONCE_W = 16;
// exclusive scan of one value per thread over a workgroup of up to 1024 threads; *total = the sum.  wsum: 17 words of LDS.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* wsum, uint32_t* total) {
    const uint32_t lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    uint32_t x = v;
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl(x, (int)(lane >= d ? lane - d : lane));
        if (lane >= d) x += y;
    }
    if (lane == 63) wsum[wid] = x;
    __syncthreads();
    if (wid == 0) {
        const uint32_t w = lane < nw ? wsum[lane] : 0;
        uint32_t t = w;
        for (uint32_t d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl(t, (int)(lane >= d ? lane - d : lane));
            if (lane >= d) t += y;
        }
        if (lane < nw) wsum[lane] = t - w;
        if (lane == 63) wsum[16] = t;
    }
    __syncthreads();
    *total = wsum[16];
    return wsum[wid] + x - v;
}
template <class E>
__global__ __launch_bounds__(1024) void presort_scatter_once(PresortArgs A, const uint32_t* __restrict__ bin_count,
                                                               uint32_t* __restrict__ bin_start, uint32_t* __restrict__ bin_cursor,
                                                               E* __restrict__ entries) {
    __shared__ uint32_t cnt[PRESORT_MAX_BINS];
    __shared__ uint32_t gbase[PRESORT_MAX_BINS];
    __shared__ uint32_t wsum[17];
    {
        // gbase[b] = start of bin b (thread t owns `per` consecutive bins)
        const uint32_t per = (A.nbins + blockDim.x - 1) / blockDim.x, b0 = threadIdx.x * per;
        uint32_t mine = 0, total;
        for (uint32_t b = b0; b < b0 + per && b < A.nbins; b++) mine += bin_count[b];
        uint32_t run = block_exclusive_scan(mine, wsum, &total);
        for (uint32_t b = b0; b < b0 + per && b < A.nbins; b++) {
            gbase[b] = run;
            if (blockIdx.x == 0) bin_start[b] = run;
            run += bin_count[b];
            cnt[b] = 0;
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) bin_start[A.nbins] = total;
    }
    __syncthreads();
    const uint32_t i = A.i0 + blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < A.i_end && (!A.mask || A.mask[i]);
    const uint32_t NONE = 0xFFFFFFFFu;
    E ent[PRESORT_ONCE_W];
    uint32_t where[PRESORT_ONCE_W];          // bin | rank within this workgroup's share of the bin << 12
    {
        const Fe s = live ? Fr::reduce_full(A.scalars[i]) : Fe{{0, 0, 0, 0}};
        const uint32_t c = A.c, NB = 1u << (c - 1), lo_mask = (1u << A.lo_bits) - 1;
        uint32_t carry = 0, k = 0, next_own = A.w_off;
#pragma unroll
        for (uint32_t w = 0; w < PRESORT_ONCE_W; w++) {
            where[w] = NONE;
            ent[w] = 0;
            if (w < A.Wall) {
                const uint32_t bit = w * c, limb = bit >> 6, off = bit & 63;
                const uint64_t lo = limb == 0 ? s.l[0] : limb == 1 ? s.l[1] : limb == 2 ? s.l[2] : limb == 3 ? s.l[3] : 0;
                const uint64_t hi = limb == 0 ? s.l[1] : limb == 1 ? s.l[2] : limb == 2 ? s.l[3] : 0;
                uint64_t v = lo >> off;
                if (off + c > 64) v |= hi << (64 - off);
                uint32_t d = (uint32_t)(v & ((1u << c) - 1)) + carry;
                uint32_t neg = 0;
                if (d > NB) { d = (1u << c) - d; neg = 1; carry = 1; } else carry = 0;
                if (w == next_own) {
                    if (live && d) {
                        const uint32_t b = presort_bin(A, k, d);
                        where[w] = b | (atomicAdd(&cnt[b], 1u) << 12);
                        ent[w] = PresortEntry<E>::make(A.flat ? w * A.n + i : i, neg, (d - 1) & lo_mask, A.idx_bits);
                    }
                    k++;
                    next_own += A.w_stride;
                }
            }
        }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < A.nbins; b += blockDim.x) {
        const uint32_t c0 = cnt[b];
        if (c0) gbase[b] += atomicAdd(&bin_cursor[b], c0);
    }
    __syncthreads();
#pragma unroll
    for (uint32_t w = 0; w < PRESORT_ONCE_W; w++)
        if (where[w] != NONE) entries[gbase[where[w] & 0xFFFu] + (where[w] >> 12)] = ent[w];
}

// 1..255, monotone in len: the task planner's length key (see msm_plan_* below)
__device__ __forceinline__ uint32_t len_key(uint32_t len, uint32_t lmax) {
    uint32_t k = (len * 255u + lmax - 1) / lmax;
    return k > 255u ? 255u : (k < 1u ? 1u : k);
}

// one workgroup per bin: LDS counting sort over the low bucket bits, bucket bounds written directly;
// the bucket lengths are known here, so the planner's task-length histogram is taken on the way
static const uint32_t PRESORT_MAX_LO = 10;
// counters[key] += 1 for every active lane, returning the lane's rank.  When the whole wavefront hits ONE counter
// (a hot bucket: boolean-heavy witnesses put half their entries into digit 1 of window 0) a single atomic serves
// all 64 lanes instead of a 64-way LDS conflict.  Every lane of the wavefront must call it.
__device__ __forceinline__ uint32_t wave_rank_add(uint32_t* counters, uint32_t key, bool active) {
    const unsigned long long act = __ballot(active);
    if (act == 0) return 0;
    const uint32_t lane = threadIdx.x & 63;
    const int first = __ffsll(act) - 1;
    const uint32_t k0 = __shfl(key, first);
    const unsigned long long same = __ballot(active && key == k0);
    uint32_t r = 0;
    if (same == act) {
        uint32_t base = 0;
        if (lane == (uint32_t)first) base = atomicAdd(&counters[k0], (uint32_t)__popcll(act));
        base = __shfl(base, first);
        r = base + (uint32_t)__popcll(act & ((1ull << lane) - 1));
    } else if (active) {
        r = atomicAdd(&counters[key], 1u);
    }
    return r;
}

template <class E>
__global__ __launch_bounds__(1024) void presort_bins(const E* __restrict__ entries, const uint32_t* __restrict__ bin_start,
                                                       uint32_t lo_bits, uint32_t idx_bits, uint32_t* __restrict__ vals_out,
                                                       uint32_t* __restrict__ bstart, uint32_t* __restrict__ bend,
                                                       uint32_t lmax, uint32_t* __restrict__ hist,
                                                       const uint8_t* __restrict__ mask, uint32_t mask_mod) {
    __shared__ uint32_t sub[1u << PRESORT_MAX_LO];
    __shared__ uint32_t off[1u << PRESORT_MAX_LO];
    __shared__ uint32_t wsum[17];
    __shared__ uint32_t lhist[256];
    const uint32_t bin = blockIdx.x, SUB = 1u << lo_bits;
    const uint32_t s = bin_start[bin], e = bin_start[bin + 1];
    for (uint32_t t = threadIdx.x; t < SUB; t += blockDim.x) sub[t] = 0;
    for (uint32_t t = threadIdx.x; t < 256; t += blockDim.x) lhist[t] = 0;
    __syncthreads();
    // four independent loads in flight per thread (the bin loops are latency-bound otherwise); the trip count is
    // uniform over the workgroup because the counter updates are wavefront-cooperative
    const uint32_t stride = 4 * blockDim.x;
    const uint32_t iters = (e - s + stride - 1) / stride;
    for (uint32_t it = 0; it < iters; it++) {
        const uint32_t i = s + threadIdx.x + it * stride;
        E v[4];
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) {
            const uint32_t j = i + u * blockDim.x;
            v[u] = j < e ? entries[j] : (E)0;
        }
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) {
            bool active = i + u * blockDim.x < e;
            if (mask && active) {                                                                           // plan variant: pair left out
                const uint32_t idx = PresortEntry<E>::val(v[u], idx_bits) & 0x7FFFFFFFu;
                active = mask[mask_mod ? idx % mask_mod : idx] != 0;                                        // (table plans: idx = window * n + i)
            }
            (void)wave_rank_add(sub, PresortEntry<E>::lo(v[u], idx_bits), active);
        }
    }
    __syncthreads();
    // exclusive scan of sub[0..SUB): thread t owns `per` consecutive counters
    const uint32_t per = (SUB + blockDim.x - 1) / blockDim.x, t0 = threadIdx.x * per;
    uint32_t mine = 0, bin_total;
    for (uint32_t t = t0; t < t0 + per && t < SUB; t++) mine += sub[t];
    uint32_t run = s + block_exclusive_scan(mine, wsum, &bin_total);
    for (uint32_t t = t0; t < t0 + per && t < SUB; t++) {
        const uint32_t cnt = sub[t];
        const uint32_t bucket = (bin << lo_bits) + t;   // == k*NB + (hi << lo_bits | lo)
        off[t] = run;
        bstart[bucket] = run;
        bend[bucket] = run + cnt;
        run += cnt;
        sub[t] = 0;                                     // becomes the placement cursor
        if (cnt) {                                      // == msm_plan_hist
            const uint32_t nt = (cnt + lmax - 1) / lmax, rem = cnt - (nt - 1) * lmax;
            if (nt > 1) atomicAdd(&lhist[255], nt - 1);
            atomicAdd(&lhist[len_key(rem, lmax)], 1u);
        }
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < 256; t += blockDim.x)
        if (lhist[t]) atomicAdd(&hist[t], lhist[t]);
    for (uint32_t it = 0; it < iters; it++) {
        const uint32_t i = s + threadIdx.x + it * stride;
        E v[4];
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) {
            const uint32_t j = i + u * blockDim.x;
            v[u] = j < e ? entries[j] : (E)0;
        }
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) {
            bool active = i + u * blockDim.x < e;
            if (mask && active) {
                const uint32_t idx = PresortEntry<E>::val(v[u], idx_bits) & 0x7FFFFFFFu;
                active = mask[mask_mod ? idx % mask_mod : idx] != 0;
            }
            const uint32_t lo = PresortEntry<E>::lo(v[u], idx_bits);
            const uint32_t r = wave_rank_add(sub, lo, active);
            if (active) vals_out[off[lo] + r] = PresortEntry<E>::val(v[u], idx_bits);
        }
    }
}

// ---------------------------------------------------------------------------
// 4/5. accumulation
// ---------------------------------------------------------------------------
// Round 6: the loop is split in two.  FAST loop: one straight path per addition (curve.h: madd_fast) for what a task is made of --
// a finite point added to a finite sum that is neither it nor its opposite; the cases the reference branches on
// (build_curve_jacobian_a0.js:322-356: infinity operands, P = +-Q) make the LANE leave that loop, and the GENERIC loop below (madd_wide,
// every case) finishes its task.  On a proof no lane ever leaves (the plan's variants drop the key's infinity points; equal points
// need a 2^-26 coincidence first); the vectors with planted infinities, duplicates and P / -P pairs run the generic loop.
// The index of the entry after next is loaded one addition early, so that no gather waits for its own index load (the round-5
// loop issued `global_load_dword; s_waitcnt vmcnt(0)` in front of every gather).
#ifndef WS_ACC_FAST
#define WS_ACC_FAST 1      // 0: the round-5 loop (madd_wide for every entry), for A/B builds
#endif
#ifndef WS_ACC_MMADD
#define WS_ACC_MMADD 1     // the second entry of a task by the affine + affine formulas
#endif
template <class C>
__device__ __forceinline__ typename C::Pt accumulate_range(const typename C::AffP* __restrict__ points,
                                                           const uint32_t* __restrict__ vals, uint32_t s, uint32_t len) {
    typedef typename C::Field F;
    typename C::Pt acc = C::infinity();
    if (len == 0) return acc;
    constexpr bool kPrefetch = sizeof(typename C::AffP) > 64 ? WS_G2_PREFETCH != 0 : WS_G1_PREFETCH != 0;
    uint32_t k = 0;
#if WS_ACC_FAST && WS_MADD_WIDE
    {
        uint32_t v = vals[s];
        typename C::AffP cp = points[v & 0x7FFFFFFFu];
        if (!F::packed_is_zero(cp.x)) {
            // the first entry starts the sum
            typename C::Aff a0 = C::unpack_aff(cp);
            a0.y = F::cneg(a0.y, (v >> 31) != 0);
            acc = typename C::Pt{a0.x, a0.y, F::one(), F::one()};
            k = 1;
            const uint32_t last = len - 1;
#if WS_ACC_MMADD
            // the second meets an affine sum: four products fewer (curve.h: mmadd_fast)
            if (len > 1) {
                v = vals[s + 1];
                cp = points[v & 0x7FFFFFFFu];
                if (!F::packed_is_zero(cp.x) && C::mmadd_fast(acc, a0, C::unpack_aff(cp), (v >> 31) != 0)) k = 2;
            }
#endif
            if (kPrefetch) {
                // G1: the next point's gather is in flight while the current one is added.  No branches around the loads: indices
                // past the task's end are clamped to its last entry (the final addition gathers that point once more)
                uint32_t v1 = vals[s + (k < last ? k : last)], v2 = vals[s + (k + 1 < last ? k + 1 : last)];
                cp = points[v1 & 0x7FFFFFFFu];
                while (k < len) {
                    const bool inf = F::packed_is_zero(cp.x);
                    typename C::Aff cur = C::unpack_aff(cp);
                    const bool neg = (v1 >> 31) != 0;
                    F::keep(cur.x);                                    // limbs taken before the registers are loaded again: no copies
                    F::keep(cur.y);
                    v1 = v2;
                    cp = points[v1 & 0x7FFFFFFFu];
                    v2 = vals[s + (k + 2 < last ? k + 2 : last)];
                    if (inf) break;
                    if (!C::madd_fast(acc, cur, neg)) break;
                    k++;
                }
            } else {
                // G2: the accumulator alone is 72 VGPRs; holding a prefetched 128-byte point as well costs a wavefront of
                // occupancy, so the gather is issued just before use (touching only the next point's cache line one addition
                // ahead was measured too: +4 %, profiles/r02_ab_g2_variants.txt) -- but its INDEX is one register and comes early
                uint32_t vn = vals[s + (k < last ? k : last)];
                while (k < len) {
                    v = vn;
                    cp = points[v & 0x7FFFFFFFu];
                    vn = vals[s + (k + 1 < last ? k + 1 : last)];
                    if (F::packed_is_zero(cp.x)) break;
                    if (!C::madd_fast(acc, C::unpack_aff(cp), (v >> 31) != 0)) break;
                    k++;
                }
            }
        }
    }
#endif
    // generic loop: every case of the reference's addition; entries from k on (all of them when the task's first point is infinity)
    if (k < len) {
        if (!kPrefetch) {
            for (; k < len; k++) {
                const uint32_t v = vals[s + k];
                const typename C::Aff cur = C::unpack_aff(points[v & 0x7FFFFFFFu]);
#if WS_MADD_WIDE
                C::madd_wide(acc, cur, (v >> 31) != 0);
#else
                C::madd(acc, cur, (v >> 31) != 0);
#endif
            }
        } else {
            // software pipeline: the next point's gather is in flight while the current one is added
            uint32_t v = vals[s + k];
            typename C::AffP nxt = points[v & 0x7FFFFFFFu];
            for (; k < len; k++) {
                const typename C::Aff cur = C::unpack_aff(nxt);
                const bool neg = (v >> 31) != 0;
                if (k + 1 < len) {
                    v = vals[s + k + 1];
                    nxt = points[v & 0x7FFFFFFFu];
                }
#if WS_MADD_WIDE
                C::madd_wide(acc, cur, neg);
#else
                C::madd(acc, cur, neg);
#endif
            }
        }
    }
    C::narrow_x(acc);
    return acc;
}

// reference-format affine points -> the device field's internal domain (identity for the 4x64 field)
template <class C>
__global__ __launch_bounds__(256) void msm_convert_points(const typename C::AffP* __restrict__ in,
                                                            typename C::AffP* __restrict__ out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = C::pack_aff(C::aff_to_internal(in[i]));
}

// ---- task planning: every bucket is cut into tasks of <= lmax entries; tasks are ordered
// longest-first (counting sort on a 255-level length key) so that (a) the lanes of one wavefront
// run (almost) the same trip count and (b) long tasks start first and never form the tail.
static const uint32_t PARTIAL_FLAG = 0x80000000u;
struct Task { uint32_t dst, start, len; };          // dst: bucket index, or PARTIAL_FLAG | partial slot
struct MultiBucket { uint32_t bucket, first_partial, ntasks; };
// Very hot buckets (>= HOT_MIN tasks: one value shared by a large part of the scalars, e.g. the ones of a
// boolean-heavy witness) get their task list written and their partial sums combined by many workgroups.
static const uint32_t HOT_MIN = 1024;      // tasks
static const uint32_t HOT_SLICE = 512;     // partial sums per first-stage wavefront
struct HotBucket { uint32_t bucket, first_partial, ntasks, task_base, rem_index, start, rem, slice_base; };

// Tasks are laid out longest-first: the slots of key k start after all tasks with a longer key.  Every workgroup
// derives those starts from the (complete) histogram itself -- 256 entries -- instead of a separate one-thread
// kernel; `cursor` (zeroed) only hands out ranks within a key.  counters[3] = total tasks.
__global__ __launch_bounds__(256) void msm_plan_emit(const uint32_t* __restrict__ bstart, const uint32_t* __restrict__ bend,
                                                       uint32_t nbuckets, uint32_t lmax, const uint32_t* __restrict__ hist,
                                                       uint32_t* __restrict__ cursor,
                                                       Task* __restrict__ tasks, uint32_t* __restrict__ counters,
                                                       MultiBucket* __restrict__ multi, HotBucket* __restrict__ hot,
                                                       uint32_t hot_min) {
    __shared__ uint32_t lcnt[256];
    __shared__ uint32_t lbase[256];
    __shared__ uint32_t first[256];          // start of key k's slots = number of tasks with a longer key
    lcnt[threadIdx.x] = 0;
    first[threadIdx.x] = hist[threadIdx.x];
    __syncthreads();
    // inclusive suffix sum over the 256 keys (Hillis-Steele), then shift to exclusive
    for (uint32_t d = 1; d < 256; d <<= 1) {
        const uint32_t v = threadIdx.x + d < 256 ? first[threadIdx.x + d] : 0;
        __syncthreads();
        first[threadIdx.x] += v;
        __syncthreads();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) counters[3] = first[0];
    const uint32_t excl = threadIdx.x + 1 < 256 ? first[threadIdx.x + 1] : 0;
    __syncthreads();
    first[threadIdx.x] = excl;
    __syncthreads();
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t cnt = 0, s = 0, nt = 0, rem = 0, krem = 0, r255 = 0, rrem = 0;
    if (b < nbuckets) {
        s = bstart[b];
        cnt = bend[b] - s;
        if (cnt) {
            nt = (cnt + lmax - 1) / lmax;
            rem = cnt - (nt - 1) * lmax;
            krem = len_key(rem, lmax);
            if (nt > 1) r255 = atomicAdd(&lcnt[255], nt - 1);
            rrem = atomicAdd(&lcnt[krem], 1u);
        }
    }
    __syncthreads();
    lbase[threadIdx.x] = lcnt[threadIdx.x] ? first[threadIdx.x] + atomicAdd(&cursor[threadIdx.x], lcnt[threadIdx.x]) : 0;
    __syncthreads();
    if (!cnt) return;
    // the full tasks were ranked before the remainder when krem == 255: keep the two ranges disjoint
    if (nt == 1) {
        tasks[lbase[krem] + rrem] = Task{b, s, rem};
    } else {
        const uint32_t pbase = atomicAdd(&counters[0], nt);
        if (nt >= hot_min) {      // task list and combine are spread over many workgroups (msm_plan_emit_hot, msm_combine_hot*)
            const uint32_t nsl = (nt + HOT_SLICE - 1) / HOT_SLICE;
            hot[atomicAdd(&counters[4], 1u)] = HotBucket{b, pbase, nt, lbase[255] + r255, lbase[krem] + rrem, s, rem,
                                                         atomicAdd(&counters[5], nsl)};
            return;
        }
        multi[atomicAdd(&counters[1], 1u)] = MultiBucket{b, pbase, nt};
        for (uint32_t k = 0; k + 1 < nt; k++) tasks[lbase[255] + r255 + k] = Task{PARTIAL_FLAG | (pbase + k), s + k * lmax, lmax};
        tasks[lbase[krem] + rrem] = Task{PARTIAL_FLAG | (pbase + nt - 1), s + (nt - 1) * lmax, rem};
    }
}

__global__ __launch_bounds__(256) void msm_plan_emit_hot(const HotBucket* __restrict__ hot, const uint32_t* __restrict__ counters,
                                                           uint32_t lmax, Task* __restrict__ tasks) {
    const uint32_t nhot = counters[4];
    for (uint32_t hb = blockIdx.x; hb < nhot; hb += gridDim.x) {
        const HotBucket h = hot[hb];
        for (uint32_t k = threadIdx.x; k + 1 < h.ntasks; k += blockDim.x)
            tasks[h.task_base + k] = Task{PARTIAL_FLAG | (h.first_partial + k), h.start + k * lmax, lmax};
        if (threadIdx.x == 0)
            tasks[h.rem_index] = Task{PARTIAL_FLAG | (h.first_partial + h.ntasks - 1), h.start + (h.ntasks - 1) * lmax, h.rem};
    }
}

// Up to 4 point sets per launch (blockIdx.y selects one): the sets of a proof that share a plan -- or run on variants of it --
// are accumulated and combined by ONE launch each, so that small sums, which fill a fraction of the chip each, run beside
// each other instead of behind each other (a rank's share of a points-sharded key; circuits up to 2^19).
template <class C>
struct AccSets {
    const typename C::AffP* points[4];
    const uint32_t* vals[4];
    const Task* tasks[4];
    const uint32_t* counters[4];
    const MultiBucket* multi[4];
    const HotBucket* hot[4];
    typename C::PtP* buckets[4];
    typename C::PtP* partials[4];
    typename C::PtP* hot_sums[4];
    uint32_t* hot_done[4];      // per hot bucket: slices folded so far (msm_combine_all; zero between launches)
};

// 4. one lane per task: mixed additions of the task's points
template <class C>
__global__ __launch_bounds__(256, (sizeof(typename C::PtP) > 128 ? WS_ACC_WAVES_G2 : WS_ACC_WAVES)) void msm_accumulate(AccSets<C> as) {
    const typename C::AffP* __restrict__ points = as.points[blockIdx.y];
    const uint32_t* __restrict__ vals = as.vals[blockIdx.y];
    const Task* __restrict__ tasks = as.tasks[blockIdx.y];
    const uint32_t* __restrict__ counters = as.counters[blockIdx.y];
    typename C::PtP* __restrict__ buckets = as.buckets[blockIdx.y];
    typename C::PtP* __restrict__ partials = as.partials[blockIdx.y];
    // the grid is sized for the worst case; the real task count stays on the device (no host round trip)
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= counters[3]) return;
    const Task k = tasks[t];
    const typename C::PtP acc = C::pack_pt(accumulate_range<C>(points, vals, k.start, k.len));
    if (k.dst & PARTIAL_FLAG) partials[k.dst & ~PARTIAL_FLAG] = acc;
    else buckets[k.dst] = acc;
}

// 5. partial sums of split buckets -> buckets.  Three shapes by the number of tasks a bucket was cut into: a few (one lane sums
// them), >= WAVE_COMBINE_MIN (one wavefront per bucket: lanes stride over the partial sums, an LDS tree folds the 64 lane sums),
// >= HOT_MIN (very hot buckets, e.g. the ones of a boolean-heavy witness: one wavefront per HOT_SLICE partial sums, then a second
// stage over the slice sums).
static const uint32_t WAVE_COMBINE_MIN = 17;
// Round 5: ONE launch (four before, three of which found nothing to do on a uniform witness and still cost a dependent launch
// each on the proof's critical queue).  Workgroups of one wavefront; blockIdx.x selects the role:
//   [0, CB_SMALL)           buckets cut into < WAVE_COMBINE_MIN tasks: one lane sums the partials
//   [.., + CB_WAVE)         buckets cut into more tasks: one wavefront per bucket
//   [.., + CB_HOT)          very hot buckets: one wavefront per HOT_SLICE partial sums (stage 1); the wavefront that completes a
//                           bucket's LAST slice -- a device-scope counter per hot bucket, zero between launches -- folds the slice
//                           sums (stage 2) and clears the counter
// (few workgroups: a launch costs ~5 us plus ~10 us per thousand workgroups it starts, whether they find work or not -- the four
//  launches this one replaces started 3 400 workgroups per sum to find nothing on a uniform witness)
// (cb_small: 256 workgroups -- a uniform full-size sum cuts next to no bucket --, but one lane per bucket where the plan cuts MOST of
//  them: a rank's share of a sharded key runs 2^16 buckets at a task cap of 16 under ~30 entries each, and 256 workgroups would
//  leave every lane a chain of four buckets: +0.09 ms on the rank's four sums, profiles/r05_s1_shard_probe_2p20.json vs r05_s2)
static const uint32_t CB_SMALL_MIN = 256, CB_SMALL_MAX = 4096, CB_WAVE = 256, CB_HOT = 256;
template <class C>
__device__ __forceinline__ typename C::Pt wave_tree_sum(typename C::PtP* sh, uint32_t lane, const typename C::Pt& mine) {
    sh[lane] = C::pack_pt(mine);
    __syncthreads();
    for (uint32_t step = 32; step >= 1; step >>= 1) {
        if (lane < step) sh[lane] = C::pack_pt(C::add(C::unpack_pt(sh[lane]), C::unpack_pt(sh[lane + step])));
        __syncthreads();
    }
    // (the sum leaves in REGISTERS: returned as the packed struct it stayed in private memory across the caller's `if (lane == 0)` --
    //  272 / 528 bytes of scratch per lane, the only per-proof kernel with any.  Its time did not change: profiles/r06_tail_l2_ab.txt, c57)
    const typename C::Pt r = C::unpack_pt(sh[0]);
    __syncthreads();
    return r;
}
template <class C>
__global__ __launch_bounds__(64) void msm_combine_all(AccSets<C> as, uint32_t CB_SMALL) {
    const uint32_t* __restrict__ counters = as.counters[blockIdx.y];
    const typename C::PtP* __restrict__ partials = as.partials[blockIdx.y];
    typename C::PtP* __restrict__ buckets = as.buckets[blockIdx.y];
    __shared__ typename C::PtP sh[64];
    const uint32_t lane = threadIdx.x;
    if (blockIdx.x < CB_SMALL) {
        const MultiBucket* __restrict__ mbs = as.multi[blockIdx.y];
        const uint32_t gt = blockIdx.x * 64 + lane, gn = CB_SMALL * 64;
        // (empty buckets are NOT cleared here any more: msm_chunks reads the plan's bounds instead -- the scan over 4 MB of bounds
        //  was 36-70 us of the round-4 combine step on a uniform witness, a chain of memory latencies)
        const uint32_t nmb = counters[1];
        for (uint32_t i = gt; i < nmb; i += gn) {
            const MultiBucket h = mbs[i];
            if (h.ntasks >= WAVE_COMBINE_MIN) continue;
            typename C::Pt acc = C::unpack_pt(partials[h.first_partial]);
            for (uint32_t k = 1; k < h.ntasks; k++) acc = C::add(acc, C::unpack_pt(partials[h.first_partial + k]));
            buckets[h.bucket] = C::pack_pt(acc);
        }
        return;
    }
    if (blockIdx.x < CB_SMALL + CB_WAVE) {
        const MultiBucket* __restrict__ mbs = as.multi[blockIdx.y];
        const uint32_t nmb = counters[1];
        for (uint32_t hb = blockIdx.x - CB_SMALL; hb < nmb; hb += CB_WAVE) {      // uniform per workgroup
            const MultiBucket h = mbs[hb];
            if (h.ntasks < WAVE_COMBINE_MIN) continue;
            typename C::Pt acc = C::infinity();
            for (uint32_t k = lane; k < h.ntasks; k += 64) acc = C::add(acc, C::unpack_pt(partials[h.first_partial + k]));
            const typename C::Pt r = wave_tree_sum<C>(sh, lane, acc);
            if (lane == 0) buckets[h.bucket] = C::pack_pt(r);
        }
        return;
    }
    const uint32_t nhot = counters[4];
    if (nhot == 0) return;
    const HotBucket* __restrict__ hot = as.hot[blockIdx.y];
    typename C::PtP* slice_sums = as.hot_sums[blockIdx.y];          // (written and read in this launch: no __restrict__)
    uint32_t* done = as.hot_done[blockIdx.y];
    const uint32_t bx = blockIdx.x - CB_SMALL - CB_WAVE;
    for (uint32_t hb = 0; hb < nhot; hb++) {                          // few entries; every workgroup walks them all
        const HotBucket h = hot[hb];
        const uint32_t nsl = (h.ntasks + HOT_SLICE - 1) / HOT_SLICE;
        const uint32_t sl0 = (bx + CB_HOT - h.slice_base % CB_HOT) % CB_HOT;       // slices dealt round-robin by their GLOBAL number
        for (uint32_t sl = sl0; sl < nsl; sl += CB_HOT) {            // uniform per workgroup
            const uint32_t lo = sl * HOT_SLICE, hi = lo + HOT_SLICE < h.ntasks ? lo + HOT_SLICE : h.ntasks;
            typename C::Pt acc = C::infinity();
            for (uint32_t k = lo + lane; k < hi; k += 64) acc = C::add(acc, C::unpack_pt(partials[h.first_partial + k]));
            const typename C::Pt r = wave_tree_sum<C>(sh, lane, acc);
            uint32_t last = 0;
            if (lane == 0) {
                slice_sums[h.slice_base + sl] = C::pack_pt(r);
                __threadfence();
                last = atomicAdd(&done[hb], 1u) + 1 == nsl ? 1u : 0u;
            }
            last = __shfl(last, 0);
            if (!last) continue;
            __threadfence();                                          // the other wavefronts' slice sums are visible from here on
            typename C::Pt tot = C::infinity();
            for (uint32_t k = lane; k < nsl; k += 64) tot = C::add(tot, C::unpack_pt(slice_sums[h.slice_base + k]));
            const typename C::Pt rr = wave_tree_sum<C>(sh, lane, tot);
            if (lane == 0) { buckets[h.bucket] = C::pack_pt(rr); done[hb] = 0; }
        }
    }
}

// ---------------------------------------------------------------------------
// 6. chunk sums: for CHUNK consecutive buckets of one window
//    S = sum B_i,  A = sum (i - i0 + 1) B_i   (descending running sum)
// ---------------------------------------------------------------------------
// the reduction tail is latency-bound (a dependent chain of ~20-30 full additions), so up to 4 point sets that
// share a plan (the prover's A, B1, C) go through it in ONE launch: blockIdx.y / .z selects the set.
// The tail kernels are written over PointIO<C> (curve.h): one lane per point, or -- the lane-paired G2 curve -- two.
template <class C>
struct TailSets {
    typedef typename PointIO<C>::Stored St;
    const St* buckets[4];
    St* chunkS[4];       // what the trees read: msm_chunks' pairs, or msm_chunks2's when there is a second level
    St* chunkA[4];
    const St* l1S[4];    // msm_chunks2's input: msm_chunks' pairs
    const St* l1A[4];
    St* chunkW[4];       // msm_chunks2's third output (the W row's elements)
    St* rows[4];         // per piece: U_0 .. U_{logJ-1}, A [, T]; in the internal domain when a piece-reduction pass follows
    St* sums[4];         // what leaves for the host (reference format)
    const uint32_t* bstart[4];   // the plan's bucket bounds: an EMPTY bucket (no entry) is infinity whatever its slot still holds from the
    const uint32_t* bend[4];     // launch before -- msm_chunks checks the bounds, so no pass over the buckets has to clear them
    St* half[4];                 // msm_tree: the two halves of the unmasked rows (per piece: [A half 0, A half 1, T half 0, T half 1, W half 0, W half 1])
    uint32_t* half_done[4];      // ... and their completion counters (per piece and row; zero between launches)
};

template <class C>
__global__ __launch_bounds__(256) void msm_chunks(TailSets<C> ts, uint32_t nchunks, uint32_t m) {
    typedef PointIO<C> IO;
    const typename IO::Stored* __restrict__ buckets = ts.buckets[blockIdx.y];
    typename IO::Stored* __restrict__ chunkS = ts.chunkS[blockIdx.y];
    typename IO::Stored* __restrict__ chunkA = ts.chunkA[blockIdx.y];
    const uint32_t j = (blockIdx.x * blockDim.x + threadIdx.x) / IO::LPP;
    if (j >= nchunks) return;
    typename C::Pt run = C::infinity(), acc = C::infinity();
    const typename IO::Stored* B = buckets + (uint64_t)j * m;
    const uint32_t* __restrict__ bs = ts.bstart[blockIdx.y] + (uint64_t)j * m;
    const uint32_t* __restrict__ be = ts.bend[blockIdx.y] + (uint64_t)j * m;
    for (int i = (int)m - 1; i >= 0; i--) {
        if (bs[i] != be[i]) run = C::add(run, IO::load(B, (uint64_t)i));      // (uniform over a lane pair: both lanes read the same bounds)
        acc = C::add(acc, run);
    }
    IO::store(chunkS, j, run);
    IO::store(chunkA, j, acc);
}

// 6'. second chunk level (round 6, closing): the m2 chunk pairs (S_j, A_j), j = k m2 + i, of msm_chunks (m1 buckets each) become the
//     pair of ONE chunk of m = m1 m2 buckets:   S'_k = sum_i S_j,   A'_k = sum_i A_j,   W_k = sum_i i S_j
//     sum_b (b + 1) B_b over the m buckets, b = i m1 + u, is A'_k + m1 W_k: the chunk's own A_j carries (u + 1), the offset i m1 falls on
//     S_j.  The factor m1 is not applied here: the W_k are summed like the A'_k (one more unmasked row of msm_tree, one more row of
//     msm_rows), and the host's Horner chain, which doubles its way from the U rows (weight m 2^q) down to the A row (weight 1) anyway,
//     adds R_W when it passes weight m1: no doubling anywhere on the GPU.
//     Latency-bound like the trees, so it runs on their curve (one point on two / four lanes), and the two chains sit on different
//     lanes: slots [0, npairs) run S' and W (2 m2 - 1 dependent additions), slots [npairs, 2 npairs) run A' (m2 - 1).
template <class C>
__global__ __launch_bounds__(256) void msm_chunks2(TailSets<C> ts, uint32_t npairs, uint32_t m2) {
    typedef PointIO<C> IO;
    const typename IO::Stored* __restrict__ S1 = ts.l1S[blockIdx.y];
    const typename IO::Stored* __restrict__ A1 = ts.l1A[blockIdx.y];
    uint32_t k = (blockIdx.x * blockDim.x + threadIdx.x) / IO::LPP;
    if (k >= 2 * npairs) return;
    if (k >= npairs) {
        k -= npairs;
        const uint64_t base = (uint64_t)k * m2;
        typename C::Pt aa = IO::load(A1, base);
        for (uint32_t i = 1; i < m2; i++) aa = C::add(aa, IO::load(A1, base + i));
        IO::store(ts.chunkA[blockIdx.y], k, aa);
        return;
    }
    const uint64_t base = (uint64_t)k * m2;
    typename C::Pt run = C::infinity(), acc = C::infinity();
    for (uint32_t i = m2 - 1; i >= 1; i--) {
        run = C::add(run, IO::load(S1, base + i));
        acc = C::add(acc, run);                      // -> sum_{i >= 1} i S_i
    }
    run = C::add(run, IO::load(S1, base));
    IO::store(ts.chunkS[blockIdx.y], k, run);
    IO::store(ts.chunkW[blockIdx.y], k, acc);
}

// ---------------------------------------------------------------------------
// 7. masked tree sums per piece: blockIdx.y = piece ("window"), blockIdx.x = q
//    q < logJ : U_q = sum_{j : bit q of j set} S_j ;  q == logJ : sum_j A_j ;  q == logJ + 1 (pieces of one bucket set) : sum_j S_j
//    to_ref: the rows leave in the reference format (they go straight to the host); otherwise they stay in the internal
//    domain for msm_rows
// ---------------------------------------------------------------------------
// (one lane per 256-byte G2 point: 256 threads at most, so that a wavefront may use the whole register file)
template <class C> struct TreeBound { static constexpr int value = (PointIO<C>::LPP == 1 && sizeof(typename PointIO<C>::Stored) > 128) ? 256 : 512; };
template <class B> struct TreeBound<CurvePairG1<B, 1>> { static constexpr int value = 1024; };      // G1: 512 slots of two lanes (115 VGPRs)
template <class B> struct TreeBound<CurvePairG1<B, 2>> { static constexpr int value = 512; };       // G2: 128 slots of four lanes (1024 threads cap the kernel at 128 VGPRs: it spills)
template <class C>
__global__ __launch_bounds__(TreeBound<C>::value) void msm_tree(TailSets<C> ts, uint32_t J, uint32_t logJ, uint32_t to_ref, uint32_t nsum) {
    typedef PointIO<C> IO;
    typedef typename IO::Stored St;
    const St* __restrict__ chunkS = ts.chunkS[blockIdx.z];
    const St* __restrict__ chunkA = ts.chunkA[blockIdx.z];
    St* __restrict__ rows = to_ref ? ts.sums[blockIdx.z] : ts.rows[blockIdx.z];
    WS_DYN_SMEM(St, sh);
    // blockIdx.x < logJ: the masked row q.  Beyond: the UNMASKED rows (sum of all A_j; sum of all S_j), which add up twice as many
    // elements as a masked one -- each is cut into two halves on two workgroups (round 5: a launch lasted as long as these rows'
    // 2 J / slots + log2(slots) additions; now every workgroup has J / 2 elements), and the workgroup that finishes second adds the
    // halves: a device-scope counter per piece and row, left at zero.
    const uint32_t w = blockIdx.y;
    const bool masked = blockIdx.x < logJ;
    const uint32_t q = masked ? blockIdx.x : logJ + ((blockIdx.x - logJ) >> 1), half = masked ? 0 : (blockIdx.x - logJ) & 1u;
    const uint32_t slot = threadIdx.x / IO::LPP, nslots = blockDim.x / IO::LPP;
    const St* src = (q == logJ ? chunkA : q == logJ + 2 ? (const St*)ts.chunkW[blockIdx.z] : chunkS) + (uint64_t)w * J;      // A; T = sum of S; W
    typename C::Pt acc = C::infinity();
    if (!masked) {
        const uint32_t Jh = (J + 1) >> 1, lo = half ? Jh : 0, hi = half ? J : Jh;
        for (uint32_t j = lo + slot; j < hi; j += nslots) acc = C::add(acc, IO::load(src, j));
    } else {
        // enumerate only the J/2 indices whose bit q is set (insert a 1 at bit q): every lane stays busy
        const uint32_t low = (1u << q) - 1;
        for (uint32_t i = slot; i < (J >> 1); i += nslots) {
            const uint32_t j = ((i & ~low) << 1) | (1u << q) | (i & low);
            acc = C::add(acc, IO::load(src, j));
        }
    }
    IO::store(sh, slot, acc);
    __syncthreads();
    for (uint32_t step = nslots >> 1; step >= 1; step >>= 1) {
        if (slot < step) IO::store(sh, slot, C::add(IO::load(sh, slot), IO::load(sh, slot + step)));
        __syncthreads();
    }
    // (every lane stays to the end: the wavefront-wide exchange below wants all of them)
    typename C::Pt r = IO::load(sh, 0);
    uint32_t finish = masked ? 1u : 0u;
    if (!masked) {
        const uint32_t hrow = q - logJ;                              // 0: A, 1: T, 2: W
        St* hv = ts.half[blockIdx.z] + ((uint64_t)w * 3 + hrow) * 2;
        uint32_t* done = ts.half_done[blockIdx.z] + (uint64_t)w * 3 + hrow;
        uint32_t second = 0;
        if (slot == 0) {
            IO::store(hv, half, r);
            __threadfence();
            if (threadIdx.x == 0) second = atomicAdd(done, 1u);
        }
        second = __shfl(second, 0);                                  // (lane 0 of the first wavefront asked; the others' copies are not used)
        if (slot == 0 && second) {                                   // the other half is there: this workgroup finishes the row
            __threadfence();
            r = C::add(r, IO::load(hv, half ^ 1u));
            if (threadIdx.x == 0) *done = 0;
            finish = 1;
        }
    }
    if (slot == 0 && finish) {
        if (to_ref) IO::store_ref(rows, (uint64_t)w * nsum + q, r);
        else IO::store(rows, (uint64_t)w * nsum + q, r);
    }
}

// ---------------------------------------------------------------------------
// 7'. piece reduction (round 4): a bucket set of NB buckets is cut into P pieces of tNB buckets (short trees above), and the
//     rows of the pieces are folded HERE instead of on the host:  sum_b (b + 1) B_b,  b = v tNB + u,
//        = sum_v [ A_v + m sum_q 2^q U_{v,q} ] + tNB sum_v v T_v
//        = R_A + m sum_q 2^q R_q + tNB sum_p 2^p V_p,   R_q = sum_v U_{v,q},  R_A = sum_v A_v,  V_p = sum_{v : bit p of v} T_v.
//     blockIdx.x = output row r (r < logJ: R_r; r == logJ: R_A; r > logJ: V_{r - logJ - 1}), blockIdx.y = group (the whole bucket
//     set of a table plan; one window of a per-window plan), blockIdx.z = point set.  One slot per piece, LDS tree, the result
//     leaves in the reference format: logJ + 1 + log2 P rows per group for the host instead of P (logJ + 2).
// ---------------------------------------------------------------------------
template <class C>
__global__ __launch_bounds__(TreeBound<C>::value) void msm_rows(TailSets<C> ts, uint32_t P, uint32_t logJ, uint32_t nsum_in, uint32_t nrows_out, uint32_t w_row) {
    typedef PointIO<C> IO;
    typedef typename IO::Stored St;
    const St* __restrict__ rows = ts.rows[blockIdx.z];
    St* __restrict__ out = ts.sums[blockIdx.z];
    WS_DYN_SMEM(St, sh);
    const uint32_t r = blockIdx.x, g = blockIdx.y;
    const uint32_t slot = threadIdx.x / IO::LPP, nslots = blockDim.x / IO::LPP;
    // (w_row: the output row of the second chunk level's W sums -- R_W = sum_v W_v, input row logJ + 2, unmasked -- or none)
    const uint32_t in_row = r == w_row ? logJ + 2 : r <= logJ ? r : logJ + 1;
    const uint32_t bit = (r > logJ && r != w_row) ? r - logJ - 1 : 0xFFFFFFFFu;
    typename C::Pt acc = C::infinity();
    for (uint32_t v = slot; v < P; v += nslots)
        if (bit == 0xFFFFFFFFu || ((v >> bit) & 1u)) acc = C::add(acc, IO::load(rows, ((uint64_t)g * P + v) * nsum_in + in_row));
    IO::store(sh, slot, acc);
    __syncthreads();
    for (uint32_t step = nslots >> 1; step >= 1; step >>= 1) {
        if (slot < step) IO::store(sh, slot, C::add(IO::load(sh, slot), IO::load(sh, slot + step)));
        __syncthreads();
    }
    if (slot == 0) IO::store_ref(out, (uint64_t)g * nrows_out + r, IO::load(sh, 0));
}


// ---------------------------------------------------------------------------
// host driver
// ---------------------------------------------------------------------------
static uint32_t pick_window(uint64_t n) {
    { const long c = tuning_get("MSM_C", 0); if (c >= 4 && c <= 16) return (uint32_t)c; }
    int lg = 0;
    while (((uint64_t)1 << (lg + 1)) <= n) lg++;
    int c = lg - 4;
    if (c < 4) c = 4;
    if (c > 16) c = 16;
    return (uint32_t)c;
}

// ---- phase 1 (independent of the points and of the curve): digits, sort, bounds, task list ----
struct MsmPlanInfo {
    uint64_t n = 0;
    uint32_t c = 0, W = 0, NB = 0, nbuckets = 0, m = 0, J = 0, logJ = 0, nsum = 0, lmax = 0, hot_cap = 0;
    uint32_t hot_min = 1024;
    uint32_t ps_lo_bits = 0, ps_idx_bits = 0, ps_nbins = 0, ps_bthr = 0;   // grouping pass geometry (presort plans)
    uint32_t ps_HB = 0, ps_tile = 0, ps_thr = 0;
    bool ps_valid = false, ps_e32 = false;
    bool building = false;        // between msm_plan_begin and msm_plan_finish: the digit histogram is being taken
    uint32_t Wall = 0, w_off = 0, w_stride = 1;   // W = owned windows; Wall = windows of the whole scalar
    // fixed-base table plans (flat): the points array holds Wall rows of n points, row w = 2^(c w) * row 0, so every window's
    // digits go into ONE set of NB buckets and no doubling is left for the tail.  For the reduction tail the bucket set is
    // cut into tW pieces of tNB buckets ("windows" of the chunk/tree kernels; classic plans: tW = W, tNB = NB).
    bool flat = false;
    uint32_t tW = 0, tNB = 0;
    // round 4: a group (the one bucket set of a table plan; a window of a per-window plan) is cut into tP pieces of tNB buckets;
    // with `reduce` the pieces' rows are folded on the GPU (msm_rows) and nrows = logJ + 1 + log2(tP) rows per GROUP reach the
    // host, otherwise nsum rows per PIECE do (the round-2 / 3 arrangement, TAIL_REDUCE=0)
    uint32_t tP = 1, groups = 1, nrows = 0;
    bool reduce = false;
    // round 6 (closing): a SECOND chunk level (msm_chunks2).  msm_chunks leaves J1 = tNB / m1 chunk pairs per piece; msm_chunks2 folds m2
    // of them into one, so that the trees see J = J1 / m2 pairs of m = m1 m2 buckets.  m2 = 1: no second level (m = m1, J = J1).
    uint32_t m1 = 0, m2 = 1, J1 = 0;
    uint32_t ntasks = 0, nmulti = 0;
    bool valid = false;
};
// ---- asynchronous completion: the last kernel's W x nsum window sums are copied to pinned host memory
// and an event is recorded; the serial host tail (a 240-doubling Horner chain, faster on a CPU core than
// on a GPU lane) runs later in msm_finish, so the GPU can already work on the next point set.
struct MsmPending {
    bool active = false;
    int which = 0;
    int plan_id = 0;
    MsmPlanInfo info;
    DevBuf d_sums, d_rows;        // what goes to the host; the pieces' rows before msm_rows folds them
    MsmScratch S;                 // this launch's accumulation buffers (buckets, partials, chunk sums)
    void* h_sums = nullptr;
    size_t h_bytes = 0;
    hipEvent_t ev = nullptr;
    const void* d_points_used = nullptr;            // the (converted) point array the accumulation reads
    void release() {
        if (h_sums) (void)hipHostFree(h_sums);
        if (ev) (void)hipEventDestroy(ev);
        h_sums = nullptr; ev = nullptr; h_bytes = 0; active = false;
        d_sums.release(); d_rows.release();
        S.buckets.release(); S.partials.release(); S.chunkS.release(); S.chunkA.release(); S.points_conv.release(); S.hot_sums.release();
        S.hot_done.release(); S.tree_half.release(); S.tree_done.release();
    }
};

// Everything the MSMs of ONE lane share: up to four plans alive at once (the prover builds the H plan on the second
// queue while the sums over the witness still read theirs; A and B1/B2 may run on variants of the witness plan),
// the launch slots (a proof has five launches in flight), the conversion buffer of caller-format points.
static const int kPlans = 4;
static const int kPendingSlots = 8;
struct MsmWorkspace {
    struct Plan { MsmPlanInfo info; MsmScratch S; } plan[kPlans];
    int cur = 0;
    MsmPending slot[kPendingSlots];
    hipEvent_t host_ev = nullptr;                      // msm_host_t: points uploaded (second queue)
    hipEvent_t conv_ev[2] = {nullptr, nullptr};
    DevBuf conv_buf;
};
static MsmWorkspace& ws(Lane& L) {
    if (!L.msm) L.msm = new MsmWorkspace();
    return *L.msm;
}
void msm_workspace_free(Lane& L) {
    if (!L.msm) return;
    MsmWorkspace& M = *L.msm;
    if (M.host_ev) (void)hipEventDestroy(M.host_ev);
    for (auto& e : M.conv_ev) if (e) (void)hipEventDestroy(e);
    for (auto& p : M.slot) p.release();
    delete L.msm;
    L.msm = nullptr;
}
void msm_select_plan(Lane& L, int id) { ws(L).cur = (id >= 0 && id < kPlans) ? id : 0; }

// error paths: forget launches whose results will never be collected (only the caller's own slots)
void msm_abort_slots(Lane& L, const int* slots, int nslots, hipStream_t a, hipStream_t b) {
    if (a) (void)hipStreamSynchronize(a);
    if (b && b != a) (void)hipStreamSynchronize(b);
    if (!L.msm) return;
    for (int k = 0; k < nslots; k++)
        if (slots[k] >= 0 && slots[k] < kPendingSlots) L.msm->slot[slots[k]].active = false;
}

template <class H>
static int msm_finish_t(MsmPending& P, typename H::Pt* out_host) {
    typedef typename H::Pt HPt;
    const MsmPlanInfo& I = P.info;
    if (I.n == 0) { *out_host = H::infinity(); P.active = false; return WS_OK; }
    struct Done { MsmPending& p; ~Done() { p.active = false; } } done{P};   // the slot is free again whatever happens
    const bool trace = tuning_get("TRACE", 0) == 1;
    const auto t_wait = std::chrono::steady_clock::now();
    WS_HIP_CHECK(hipEventSynchronize(P.ev));
    const auto t_tail = std::chrono::steady_clock::now();
    const HPt* sums = reinterpret_cast<const HPt*>(P.h_sums);
    if (I.flat && I.reduce) {
        // table plans with the pieces folded on the GPU (msm_rows): R_A + m sum_q 2^q R_q + tNB sum_p 2^p V_p -- two short Horner
        // chains over logJ + 1 + log2(tP) rows, c - 1 doublings in all
        uint32_t logm = 0, logt = 0, logP = 0;
        while ((1u << logm) < I.m) logm++;
        while ((1u << logt) < I.tNB) logt++;
        while ((1u << logP) < I.tP) logP++;
        HPt acc = H::infinity();
        for (int q = (int)I.logJ - 1; q >= 0; q--) acc = H::add(H::dbl(acc), sums[q]);
        if (I.m2 > 1) {
            // second chunk level: R_A' + m1 R_W + m sum_q 2^q R_q -- the W row (the last one) joins the chain at weight m1
            uint32_t logm1 = 0;
            while ((1u << logm1) < I.m1) logm1++;
            for (uint32_t k = logm1; k < logm; k++) acc = H::dbl(acc);
            acc = H::add(acc, sums[I.logJ + 1 + logP]);
            for (uint32_t k = 0; k < logm1; k++) acc = H::dbl(acc);
        } else {
            for (uint32_t k = 0; k < logm; k++) acc = H::dbl(acc);
        }
        acc = H::add(acc, sums[I.logJ]);
        HPt tot = H::infinity();
        for (int pb = (int)logP - 1; pb >= 0; pb--) tot = H::add(H::dbl(tot), sums[I.logJ + 1 + (uint32_t)pb]);
        for (uint32_t k = 0; k < logt; k++) tot = H::dbl(tot);
        *out_host = H::add(acc, tot);
        if (trace)
            fprintf(stderr, "[wsnark trace]   msm finish (%s, table, rows folded on the GPU): waited %.3f ms for the GPU, host tail %.3f ms\n", sizeof(HPt) > 128 ? "G2" : "G1",
                    std::chrono::duration<double, std::milli>(t_tail - t_wait).count(),
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_tail).count());
        return WS_OK;
    }
    if (I.flat) {
        // table plans: sum_b (b+1) B_b over ONE bucket set, b = v*tNB + u:  sum_v [ A_v + m sum_q 2^q U_{v,q} ] + tNB sum_v v T_v
        // (rows per piece v: U_{v,0..logJ-1}, A_v, T_v = sum of the piece's buckets).  The pieces are summed row-wise first,
        // so the doubling chain is c - 1 long whatever the number of pieces.
        uint32_t logm = 0, logt = 0;
        while ((1u << logm) < I.m) logm++;
        while ((1u << logt) < I.tNB) logt++;
        HPt acc = H::infinity();
        for (int q = (int)I.logJ - 1; q >= 0; q--) {
            acc = H::dbl(acc);
            for (uint32_t v = 0; v < I.tW; v++) acc = H::add(acc, sums[(size_t)v * I.nsum + q]);
        }
        for (uint32_t k = 0; k < logm; k++) acc = H::dbl(acc);
        for (uint32_t v = 0; v < I.tW; v++) acc = H::add(acc, sums[(size_t)v * I.nsum + I.logJ]);
        if (I.tW > 1) {
            HPt run = H::infinity(), tot = H::infinity();
            for (uint32_t v = I.tW - 1; v >= 1; v--) {
                run = H::add(run, sums[(size_t)v * I.nsum + I.logJ + 1]);
                tot = H::add(tot, run);
            }
            for (uint32_t k = 0; k < logt; k++) tot = H::dbl(tot);
            acc = H::add(acc, tot);
        }
        *out_host = acc;
        if (trace)
            fprintf(stderr, "[wsnark trace]   msm finish (%s, table): waited %.3f ms for the GPU, host tail %.3f ms\n", sizeof(HPt) > 128 ? "G2" : "G1",
                    std::chrono::duration<double, std::milli>(t_tail - t_wait).count(),
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_tail).count());
        return WS_OK;
    }
    // 8. host tail: result = sum_w 2^(c w) [ A_w + m * sum_q 2^q U_{w,q} ].  ONE Horner chain over the bit
    // positions of the whole scalar (MSB first): U_{w,q} sits at bit c*w + log2(m) + q, A_w at bit c*w, and
    // log2(m) + logJ = c - 1, so every window's terms fall inside its own c positions: c*Wall doublings and
    // (logJ + 1) additions per owned window -- no per-window inner chain.
    uint32_t logm = 0, logt = 0, logP = 0;
    while ((1u << logm) < I.m) logm++;
    while ((1u << logt) < I.tNB) logt++;
    while ((1u << logP) < I.tP) logP++;
    // (windows cut into pieces, rows folded by msm_rows: V_{w,p} = sum_{v: bit p} T_{w,v} sits at bit c*w + log2(tNB) + p)
    const uint32_t rows_per_window = I.reduce ? I.nrows : I.nsum;
    HPt acc = H::infinity();
    bool started = false;                                  // leading doublings of infinity are skipped
    for (int wg = (int)I.Wall - 1; wg >= 0; wg--) {      // global window index; rows exist for the owned ones
        const bool owned = (uint32_t)wg >= I.w_off && ((uint32_t)wg - I.w_off) % I.w_stride == 0;
        const uint32_t krow = owned ? ((uint32_t)wg - I.w_off) / I.w_stride : 0;
        const HPt* row = owned ? &sums[(size_t)krow * rows_per_window] : nullptr;
        for (int k = (int)I.c - 1; k >= 0; k--) {
            if (started) acc = H::dbl(acc);
            if (!owned) continue;
            if (I.reduce && (uint32_t)k >= logt && (uint32_t)k - logt < logP) { acc = H::add(acc, row[I.logJ + 1 + (uint32_t)k - logt]); started = true; }
            if ((uint32_t)k >= logm && (uint32_t)k - logm < I.logJ) { acc = H::add(acc, row[(uint32_t)k - logm]); started = true; }
            if (k == 0) { acc = H::add(acc, row[I.logJ]); started = true; }
        }
    }
    *out_host = acc;
    if (trace)
        fprintf(stderr, "[wsnark trace]   msm finish (%s): waited %.3f ms for the GPU, host Horner %.3f ms\n", sizeof(HPt) > 128 ? "G2" : "G1",
                std::chrono::duration<double, std::milli>(t_tail - t_wait).count(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_tail).count());
    return WS_OK;
}

// launches that still read plan `plan_id`'s buffers on their own streams must finish before the buffers are rewritten
static int wait_plan_users(MsmWorkspace& M, int plan_id, hipStream_t s) {
    for (auto& p : M.slot)
        if (p.active && p.ev && p.info.n && p.plan_id == plan_id) WS_HIP_CHECK(hipStreamWaitEvent(s, p.ev, 0));
    return WS_OK;
}

uint32_t msm_table_rows(uint32_t tc) { return tc ? (255 + tc - 1) / tc : 1; }

// window width of the fixed-base tables for n pairs: c = log2 n, so that a bucket of the ONE bucket set receives about
// 2 * rows entries (26 at 2^20: 13 rows, 2^19 buckets) -- the same load per lane as the per-window method has, with
// 13 passes over the points instead of 16.  Narrower windows were measured at 2^20 (profiles/r02_sweep_key_table.txt):
// their few, heavily loaded buckets must be cut into several tasks each and the partial sums folded again, which costs
// more than the shorter tail saves (c = 18: 12.5 ms per proof, c = 20: 10.2-10.5 ms, plain sections 11.25 ms).
// Between powers of two the width goes up at n = 1.25 * 2^k: there rows(c) * n + ~3 * 2^(c-1) additions (passes + bucket
// reduction) break even, and the wider window keeps one task per bucket (14-32 entries each).
// Capped at 20: 2^21 buckets would need 8-byte grouping entries.  WSNARK_TABLE_C overrides.
uint32_t msm_table_window(uint64_t n) {
    { const long c = tuning_get("TABLE_C", 0); if (c >= 4 && c <= 22) return (uint32_t)c; }
    int lg = 0;
    while (((uint64_t)1 << (lg + 1)) <= n) lg++;
    int c = (n * 4 >= ((uint64_t)5 << lg)) ? lg + 1 : lg;
    if (c < 4) c = 4;
    if (c > 20) c = 20;
    return (uint32_t)c;
}

// The plan in three steps, so that the FIRST pass over the scalars -- the digit histogram -- can run on parts of the vector
// while the rest is still on its way to the device (prove.hip: a witness uploaded chunk by chunk):
//   msm_plan_begin   geometry, buffers, cleared counters
//   msm_plan_count   histogram of the scalars [i0, i1) (any number of calls that together cover [0, n), on queues ordered
//                    before the one msm_plan_finish runs on)
//   msm_plan_finish  scan, scatter, per-bin sort, task list
// msm_plan_dev is the three in a row.
int msm_plan_begin(Lane& L, uint64_t n, WindowShard sh, hipStream_t s, uint32_t table_c) {
    Context* X = ctx();
    if (!X) return WS_ERR_NOINIT;
    if (!s) s = L.stream;
    MsmWorkspace& M = ws(L);
    MsmPlanInfo& I = M.plan[M.cur].info;
    I = MsmPlanInfo();
    if (n == 0) { I.valid = true; return WS_OK; }
    if (n > ((uint64_t)1 << 28)) return WS_ERR_SIZE;
    if (sh.stride == 0) return WS_ERR_ARG;
    I.n = n;
    I.flat = table_c != 0;
    I.c = I.flat ? table_c : pick_window(n);
    if (I.c < 2 || I.c > 22) return WS_ERR_ARG;
    I.Wall = (255 + I.c - 1) / I.c;
    if (I.flat && (uint64_t)I.Wall * n >= ((uint64_t)1 << 31)) return WS_ERR_SIZE;   // entries index the table, sign in bit 31
    I.w_off = sh.off;
    I.w_stride = sh.stride;
    I.W = I.w_off < I.Wall ? (I.Wall - I.w_off + I.w_stride - 1) / I.w_stride : 0;
    if (I.W == 0) { I.n = 0; I.valid = true; return WS_OK; }   // this rank owns no window: partial = infinity
    I.NB = 1u << (I.c - 1);
    I.nbuckets = I.flat ? I.NB : I.W * I.NB;
    const uint64_t total = n * I.W;
    if (total >= ((uint64_t)1 << 32)) return WS_ERR_SIZE;
    // Reduction-tail geometry.  The tail is a chain of dependent additions -- 2 m in a chunk lane, then log2 of what is left in
    // trees -- on few wavefronts; its WORK only matters at full size, its DEPTH matters whenever nothing else fills the chip
    // (small sums, a rank's share of a sharded key, a stand-alone MSM).  So: chunks of 8 buckets and 2^15-bucket pieces for
    // large bucket sets (the round-2 / 3 shape), chunks of 4 and 2^11-bucket pieces below 2^18 buckets, and the pieces' rows are
    // folded on the GPU (msm_rows) so that short pieces do not turn into host work.  WSNARK_MSM_CHUNK / WSNARK_TAIL_BITS
    // override (tests of the geometries; A/B: profiles/r04_*).
    const bool big_set = I.NB >= (1u << 18);
    uint32_t chunk = (I.flat && !big_set) ? 4u : CHUNK;
    { const long v = tuning_get("MSM_CHUNK", 0); if (v == 2 || v == 4 || v == 8 || v == 16 || v == 32) chunk = (uint32_t)v; }
    {
        uint32_t tbits = I.flat ? (big_set ? 15u : 11u) : 31u;      // buckets per tail piece; per-window plans: the whole window
        { const long v = tuning_get("TAIL_BITS", 0); if (v >= 3 && v <= 20) tbits = (uint32_t)v; }
        I.tNB = (tbits >= 31 || I.NB < (1u << tbits)) ? I.NB : (1u << tbits);
        I.tP = I.NB / I.tNB;
        if (I.tP > 256) { I.tP = 256; I.tNB = I.NB / 256; }        // (msm_rows folds a group's pieces in one workgroup)
        I.groups = I.flat ? 1 : I.W;
        I.tW = I.groups * I.tP;
    }
    I.m1 = I.tNB < chunk ? I.tNB : chunk;
    I.J1 = I.tNB / I.m1;
    I.m2 = 1;
    {
        // Second chunk level for the bucket sets of table plans from 2^17 buckets on: the masked tree rows cost (logJ / 2 + 2) J additions per piece,
        // i.e. half of what msm_chunks itself does at J = 4096, on workgroups that spend most of their time in a nine-step LDS tree; a
        // second running sum over m2 chunk pairs costs 3 additions per pair and leaves the trees 1 / m2 of their elements.
        // WSNARK_TAIL_L2 = 1 / 2 / 4 / 8 (1: off); A/B: profiles/r06_tail_l2_ab.txt
        // (table plans whose pieces' rows are folded on the GPU: the host branch that knows the W row, msm_finish_t)
        long v = tuning_get("TAIL_L2", (I.flat && I.NB >= (1u << 17)) ? WS_TAIL_L2_DEFAULT : 1);      // (2^16 buckets and fewer: no gain measured, call c47 / c48)
        if ((v == 2 || v == 4 || v == 8) && I.flat && I.tP > 1 && I.J1 >= 8u * (uint32_t)v) I.m2 = (uint32_t)v;      // (the trees keep at least eight pairs)
    }
    I.m = I.m1 * I.m2;
    I.J = I.J1 / I.m2;
    while ((1u << I.logJ) < I.J) I.logJ++;
    I.nsum = I.logJ + 1 + ((I.flat || I.tP > 1) ? 1 : 0) + (I.m2 > 1 ? 1 : 0);      // U_0 .. U_{logJ-1}, A [, T [, W]]
    I.reduce = I.tP > 1;
    {
        uint32_t logP = 0;
        while ((1u << logP) < I.tP) logP++;
        I.nrows = I.reduce ? I.logJ + 1 + logP + (I.m2 > 1 ? 1 : 0) : I.nsum;      // (the W row of a second chunk level comes last)
    }
    // task length cap: twice the mean bucket load
    I.lmax = (uint32_t)(2 * (((I.flat ? total : n) + I.NB - 1) / I.NB));
    if (I.flat && I.NB < (1u << 19) && total >= ((uint64_t)1 << 22) && total / I.NB > 64) {
        // one SMALL bucket set under many entries (a forced narrow window; the default width keeps 14-32 entries per
        // bucket): a lane per bucket would leave most of the chip idle (2^15 buckets at 2^20 pairs = 2 wavefronts per CU:
        // measured 4.0 ms instead of 1.26 ms), so the buckets are cut until about 2^19 tasks exist
        const uint32_t by_tasks = (uint32_t)((total + (1u << 19) - 1) >> 19);
        if (I.lmax > by_tasks) I.lmax = by_tasks;
    }
    // Very small sums (round 3): fewer bucket runs than the chip has SIMDs x 64 lanes (2^16).  One lane per run then leaves SIMDs
    // without any wavefront, so the runs are cut until about 2^17 tasks exist and the partial sums are folded by
    // msm_combine_small.  2^16-constraint proofs: 2.43 -> 1.95 ms together with the batched accumulation of msm_g1_launch_batch.
    // With 2^16 runs or more every SIMD has work, and sums that run beside each other already fill the issue slots -- cutting
    // further only adds combine work (measured on a rank's share of a 2^20 key over 8 ranks: 2.07 vs 2.12 ms,
    // profiles/r03_s14_small_sums.txt).
    uint32_t lmin = 32;
    {
        if (I.nbuckets < (1u << 16) && total >= ((uint64_t)1 << 16)) {
            const uint32_t by_tasks = (uint32_t)((total + (1u << 17) - 1) >> 17);
            if (by_tasks < I.lmax) I.lmax = by_tasks;
            lmin = 8;
        } else if (I.flat && I.nbuckets == (1u << 16) && I.lmax > 16) {
            // Round 4, exactly 2^16 bucket runs (a rank's share of a 2^20 key over 8 ranks, 2^17 proofs): with the shorter reduction
            // tails the accumulations are what is left of such a sum, and a lane's chain of ~30 additions (G2: 0.7 ms whatever the
            // parallelism) is their floor -- runs cut at 16 entries + the G1 sets in one launch (msm_g1_launch_batch): the rank's four
            // witness sums 1.80 -> 1.66 ms (profiles/r04_s4_shard_sweep.txt; round 3 had measured the opposite with its longer tails).
            I.lmax = 16;
            lmin = 16;
        }
    }
    { const long v = tuning_get("MSM_LMAX", 0); if (v >= 4 && v <= 65536) { I.lmax = (uint32_t)v; lmin = 4; } }
    if (I.lmax < lmin) I.lmax = lmin;
    I.hot_cap = (uint32_t)(total / I.lmax) + I.nbuckets + 16;
    const uint32_t c = I.c, W = I.W, nbuckets = I.nbuckets, lmax = I.lmax;

    int rc = wait_plan_users(M, M.cur, s);
    if (rc) return rc;
    MsmScratch& S = M.plan[M.cur].S;
    WS_HIP_CHECK(S.vals_out.reserve(total * 4));
    WS_HIP_CHECK(S.bstart.reserve((size_t)nbuckets * 4));
    WS_HIP_CHECK(S.bend.reserve((size_t)nbuckets * 4));
    WS_HIP_CHECK(S.counters.reserve((size_t)CNT_BINS * 4 + ((size_t)PRESORT_MAX_BINS + 1) * 4 * 3));
    WS_HIP_CHECK(S.tasks.reserve((size_t)I.hot_cap * sizeof(Task)));
    WS_HIP_CHECK(S.multi.reserve((size_t)I.hot_cap * sizeof(MultiBucket)));
    uint32_t hot_min = HOT_MIN;      // (WSNARK_MSM_HOT_MIN: lets small tests reach the hot-bucket path)
    { const long v = tuning_get("MSM_HOT_MIN", 0); if (v >= 2) hot_min = (uint32_t)v; }
    I.hot_min = hot_min;
    WS_HIP_CHECK(S.hot.reserve(((size_t)I.hot_cap / hot_min + 16) * sizeof(HotBucket)));

    // counters: [0] partial slots, [1] multi-task buckets, [3] total tasks; [16..271] length histogram,
    // [272..527] cursors
    // (the coarse-bin counts of the grouping pass follow at [1024 ..]: one memset clears both)
    uint32_t* d_cnt = S.counters.as<uint32_t>();
    // grouping-pass geometry (swept in rounds 1-2, profiles/r01_sweep_presort_*.txt, r02_sweep_presort_geometry.txt): 1024 scalars per
    // workgroup of 1024 threads, 8 low bucket bits sorted per bin
    const uint32_t env_lo = 8, env_tile = 1024, env_thr = 1024;
    const bool env_e64 = tuning_get("MSM_ENTRY64", 0) != 0;      // (tests: the 8-byte grouping entries a 2^24 table key needs, at small sizes)
    uint32_t lo_bits = env_lo > PRESORT_MAX_LO ? PRESORT_MAX_LO : env_lo;
    if (lo_bits > c - 1) lo_bits = c - 1;
    const uint32_t Wb = I.flat ? 1 : W;        // bucket sets the bins are spread over
    while ((uint64_t)Wb * (I.NB >> lo_bits) > PRESORT_MAX_BINS && lo_bits < c - 1 && lo_bits < PRESORT_MAX_LO) lo_bits++;
    // one bucket set: keep about as many bins (= workgroups of the per-bin sort) as the 16-window plans have
    // (not below 7 low bits: with 64 LDS counters per bin the counting sort's atomics collide -- 0.94 ms instead of 0.15-0.22 ms)
    while (I.flat && (I.NB >> lo_bits) < 2048 && lo_bits > 7) lo_bits--;
    uint32_t idx_bits = 1;
    while (((uint64_t)1 << idx_bits) < (I.flat ? (uint64_t)I.Wall * n : n)) idx_bits++;
    // large inputs (2^24 pairs): give up low bucket bits while that keeps the entries at 4 bytes and the bin count in range
    while (!env_e64 && idx_bits + 1 + lo_bits > 32 && lo_bits > 1 && (uint64_t)Wb * (I.NB >> (lo_bits - 1)) <= PRESORT_MAX_BINS) lo_bits--;
    const uint32_t HB = I.NB >> lo_bits, nbins = Wb * HB;
    if (nbins > PRESORT_MAX_BINS) { set_last_error("msm: grouping pass geometry out of range"); return WS_ERR_SIZE; }
    {
        // ---- grouping by coarse bins + per-bin LDS counting sort (see the kernels above) ----
        const bool e32 = !env_e64 && idx_bits + 1 + lo_bits <= 32;
        WS_HIP_CHECK(S.entries.reserve(total * (e32 ? 4 : 8)));
        WS_HIP_CHECK(hipMemsetAsync(S.counters.p, 0, (CNT_BINS + 3 * ((size_t)nbins + 1)) * 4, s));   // (counts, starts, cursors)
        I.ps_valid = true;
        I.ps_lo_bits = lo_bits; I.ps_idx_bits = idx_bits; I.ps_nbins = nbins; I.ps_e32 = e32;
        I.ps_HB = HB; I.ps_tile = env_tile; I.ps_thr = env_thr;
    }
    (void)d_cnt; (void)hot_min; (void)lmax; (void)nbuckets;
    I.building = true;
    return WS_OK;
}

static PresortArgs plan_presort_args(const MsmPlanInfo& I, const Fe* d_scalars, uint32_t i0, uint32_t i_end) {
    return PresortArgs{d_scalars, (uint32_t)I.n, I.c, I.Wall, I.w_off, I.w_stride, i0, i_end, I.ps_lo_bits, I.ps_HB, I.ps_nbins, I.ps_tile,
                       I.ps_idx_bits, nullptr, I.flat ? 1u : 0u};
}

int msm_plan_count(Lane& L, const Fe* d_scalars, uint64_t i0, uint64_t i1, hipStream_t s) {
    Context* X = ctx();
    if (!X) return WS_ERR_NOINIT;
    if (!s) s = L.stream;
    MsmWorkspace& M = ws(L);
    MsmPlanInfo& I = M.plan[M.cur].info;
    if (I.n == 0 && I.valid) return WS_OK;                 // (nothing to group: n == 0, or a rank that owns no window)
    if (!I.building) { set_last_error("msm_plan_count: no plan is being built"); return WS_ERR_ARG; }
    if (!d_scalars || i0 > i1 || i1 > I.n) return WS_ERR_ARG;
    if (!I.ps_valid || i0 == i1) return WS_OK;             // (the hipCUB pipeline has no histogram pass)
    MsmScratch& S = M.plan[M.cur].S;
    uint32_t* bin_count = S.counters.as<uint32_t>() + CNT_BINS;
    const PresortArgs PA = plan_presort_args(I, d_scalars, (uint32_t)i0, (uint32_t)i1);
    X->timer.begin("msm_presort_count", s);
    hipLaunchKernelGGL(presort_count, dim3(ceil_div_u64(i1 - i0, I.ps_tile)), dim3(I.ps_thr), 0, s, PA, bin_count);
    X->timer.end(s);
    WS_HIP_CHECK(hipGetLastError());
    return WS_OK;
}

int msm_plan_finish(Lane& L, const Fe* d_scalars, hipStream_t s) {
    Context* X = ctx();
    if (!X) return WS_ERR_NOINIT;
    if (!s) s = L.stream;
    MsmWorkspace& M = ws(L);
    MsmPlanInfo& I = M.plan[M.cur].info;
    if (I.n == 0 && I.valid) return WS_OK;
    if (!I.building) { set_last_error("msm_plan_finish: no plan is being built"); return WS_ERR_ARG; }
    if (!d_scalars) return WS_ERR_ARG;
    I.building = false;
    MsmScratch& S = M.plan[M.cur].S;
    KernelTimer& T = X->timer;
    const uint64_t n = I.n, total = n * I.W;
    const uint32_t nbuckets = I.nbuckets, lmax = I.lmax, hot_min = I.hot_min;
    uint32_t* d_cnt = S.counters.as<uint32_t>();
    if (I.ps_valid) {
        const uint32_t nbins = I.ps_nbins, lo_bits = I.ps_lo_bits, idx_bits = I.ps_idx_bits;
        const bool e32 = I.ps_e32;
        uint32_t* bin_count = d_cnt + CNT_BINS;
        uint32_t* bin_start = bin_count + (nbins + 1);
        uint32_t* bin_cursor = bin_start + (nbins + 1);
        const PresortArgs PA = plan_presort_args(I, d_scalars, 0u, (uint32_t)n);
        const dim3 grid(ceil_div_u64(n, I.ps_tile)), blk(I.ps_thr);
        const bool once = I.Wall <= PRESORT_ONCE_W && I.ps_tile == I.ps_thr;
        if (!once) {
            T.begin("msm_presort_scan", s);
            hipLaunchKernelGGL(presort_scan, dim3(1), dim3(256), 0, s, bin_count, nbins, bin_start, bin_cursor);
            T.end(s);
        }
        T.begin("msm_presort_scatter", s);
        if (once && e32) hipLaunchKernelGGL(presort_scatter_once<uint32_t>, grid, blk, 0, s, PA, bin_count, bin_start, bin_cursor, S.entries.as<uint32_t>());
        else if (once) hipLaunchKernelGGL(presort_scatter_once<uint64_t>, grid, blk, 0, s, PA, bin_count, bin_start, bin_cursor, S.entries.as<uint64_t>());
        else if (e32) hipLaunchKernelGGL(presort_scatter<uint32_t>, grid, blk, 0, s, PA, bin_cursor, S.entries.as<uint32_t>());
        else hipLaunchKernelGGL(presort_scatter<uint64_t>, grid, blk, 0, s, PA, bin_cursor, S.entries.as<uint64_t>());
        T.end(s);
        // one workgroup per bin, about eight entries per thread (two rounds of four loads in flight)
        uint32_t bthr = (uint32_t)((total / nbins / 8 + 63) / 64 * 64);
        bthr = bthr < 64 ? 64 : bthr > 1024 ? 1024 : bthr;
        const dim3 bblk(bthr);
        T.begin("msm_presort_bins", s);
        if (e32)
            hipLaunchKernelGGL(presort_bins<uint32_t>, dim3(nbins), bblk, 0, s, S.entries.as<uint32_t>(), bin_start, lo_bits,
                               idx_bits, S.vals_out.as<uint32_t>(), S.bstart.as<uint32_t>(), S.bend.as<uint32_t>(), lmax, d_cnt + CNT_HIST, nullptr, 0u);
        else
            hipLaunchKernelGGL(presort_bins<uint64_t>, dim3(nbins), bblk, 0, s, S.entries.as<uint64_t>(), bin_start, lo_bits,
                               idx_bits, S.vals_out.as<uint32_t>(), S.bstart.as<uint32_t>(), S.bend.as<uint32_t>(), lmax, d_cnt + CNT_HIST, nullptr, 0u);
        T.end(s);
        WS_HIP_CHECK(hipGetLastError());
        I.ps_bthr = bthr;
    } else {
        set_last_error("msm_plan_finish: the plan has no grouping pass");
        return WS_ERR_ARG;
    }
    T.begin("msm_plan", s);
    hipLaunchKernelGGL(msm_plan_emit, dim3(ceil_div_u64(nbuckets, 256)), dim3(256), 0, s, S.bstart.as<uint32_t>(),
                       S.bend.as<uint32_t>(), nbuckets, lmax, d_cnt + CNT_HIST, d_cnt + CNT_CURSOR, S.tasks.as<Task>(), d_cnt,
                       S.multi.as<MultiBucket>(), S.hot.as<HotBucket>(), hot_min);
    hipLaunchKernelGGL(msm_plan_emit_hot, dim3(64), dim3(256), 0, s, S.hot.as<HotBucket>(), d_cnt, lmax, S.tasks.as<Task>());
    T.end(s);
    WS_HIP_CHECK(hipGetLastError());
    // task / partial counts stay on the device (counters[3], [1], [0]); by construction
    // tasks <= total/lmax + nbuckets <= hot_cap, so the worst-case grids below always cover them
    I.ntasks = I.hot_cap;
    I.nmulti = 0;
    I.valid = true;
    return WS_OK;
}


int msm_plan_dev(Lane& L, const Fe* d_scalars, uint64_t n, WindowShard sh, hipStream_t s, uint32_t table_c) {
    if (n && !d_scalars) return WS_ERR_ARG;
    int rc = msm_plan_begin(L, n, sh, s, table_c);
    if (!rc) rc = msm_plan_count(L, d_scalars, 0, n, s);
    if (!rc) rc = msm_plan_finish(L, d_scalars, s);
    return rc;
}

int msm_plan_variant(Lane& L, int src_id, int dst_id, const uint8_t* d_mask, hipStream_t s) {
    Context* X = ctx();
    if (!X) return WS_ERR_NOINIT;
    if (!s) s = L.stream;
    if (src_id < 0 || src_id >= kPlans || dst_id < 0 || dst_id >= kPlans || src_id == dst_id || !d_mask) return WS_ERR_ARG;
    MsmWorkspace& M = ws(L);
    const MsmPlanInfo src = M.plan[src_id].info;
    MsmScratch& SS = M.plan[src_id].S;
    MsmPlanInfo& I = M.plan[dst_id].info;
    MsmScratch& S = M.plan[dst_id].S;
    if (!src.valid || !src.ps_valid || src.n == 0) { set_last_error("msm_plan_variant: the source plan has no shared grouping pass"); return WS_ERR_ARG; }
    int rc = wait_plan_users(M, dst_id, s);
    if (rc) return rc;
    I = src;
    I.ps_valid = false;
    I.valid = false;
    const uint64_t total = src.n * src.W;
    WS_HIP_CHECK(S.vals_out.reserve(total * 4));
    WS_HIP_CHECK(S.bstart.reserve((size_t)src.nbuckets * 4));
    WS_HIP_CHECK(S.bend.reserve((size_t)src.nbuckets * 4));
    WS_HIP_CHECK(S.counters.reserve((size_t)CNT_BINS * 4 + ((size_t)PRESORT_MAX_BINS + 1) * 4 * 3));
    WS_HIP_CHECK(S.tasks.reserve((size_t)src.hot_cap * sizeof(Task)));
    WS_HIP_CHECK(S.multi.reserve((size_t)src.hot_cap * sizeof(MultiBucket)));
    WS_HIP_CHECK(S.hot.reserve(((size_t)src.hot_cap / src.hot_min + 16) * sizeof(HotBucket)));
    uint32_t* d_cnt = S.counters.as<uint32_t>();
    WS_HIP_CHECK(hipMemsetAsync(S.counters.p, 0, (size_t)CNT_BINS * 4, s));
    const uint32_t* bin_start = SS.counters.as<uint32_t>() + CNT_BINS + (src.ps_nbins + 1);
    KernelTimer& T = X->timer;
    T.begin("msm_presort_bins", s);
    if (src.ps_e32)
        hipLaunchKernelGGL(presort_bins<uint32_t>, dim3(src.ps_nbins), dim3(src.ps_bthr), 0, s, SS.entries.as<uint32_t>(), bin_start,
                           src.ps_lo_bits, src.ps_idx_bits, S.vals_out.as<uint32_t>(), S.bstart.as<uint32_t>(), S.bend.as<uint32_t>(),
                           src.lmax, d_cnt + CNT_HIST, d_mask, src.flat ? (uint32_t)src.n : 0u);
    else
        hipLaunchKernelGGL(presort_bins<uint64_t>, dim3(src.ps_nbins), dim3(src.ps_bthr), 0, s, SS.entries.as<uint64_t>(), bin_start,
                           src.ps_lo_bits, src.ps_idx_bits, S.vals_out.as<uint32_t>(), S.bstart.as<uint32_t>(), S.bend.as<uint32_t>(),
                           src.lmax, d_cnt + CNT_HIST, d_mask, src.flat ? (uint32_t)src.n : 0u);
    T.end(s);
    T.begin("msm_plan", s);
    hipLaunchKernelGGL(msm_plan_emit, dim3(ceil_div_u64(src.nbuckets, 256)), dim3(256), 0, s, S.bstart.as<uint32_t>(),
                       S.bend.as<uint32_t>(), src.nbuckets, src.lmax, d_cnt + CNT_HIST, d_cnt + CNT_CURSOR, S.tasks.as<Task>(), d_cnt,
                       S.multi.as<MultiBucket>(), S.hot.as<HotBucket>(), src.hot_min);
    hipLaunchKernelGGL(msm_plan_emit_hot, dim3(64), dim3(256), 0, s, S.hot.as<HotBucket>(), d_cnt, src.lmax, S.tasks.as<Task>());
    T.end(s);
    WS_HIP_CHECK(hipGetLastError());
    I.valid = true;
    return WS_OK;
}

// ---- phase 2: bucket accumulation and reduction for one point set, against the current plan ----
// accumulation + combine of up to 4 launches (one plan or its variants: same geometry) in ONE launch each
template <class C>
static int msm_acc_sets(Lane& L, MsmPending* const* Ps, int nsets, int which, hipStream_t s) {
    typedef typename C::PtP Pt;
    Context* X = ctx();
    MsmWorkspace& M = ws(L);
    KernelTimer& T = X->timer;
    AccSets<C> as;
    uint32_t ntasks = 0, cb_small = CB_SMALL_MIN;
    for (int k = 0; k < 4; k++) {
        MsmPending& Q = *Ps[k < nsets ? k : 0];
        MsmScratch& QS = M.plan[Q.plan_id].S;
        as.points[k] = reinterpret_cast<const typename C::AffP*>(Q.d_points_used);
        as.vals[k] = QS.vals_out.as<uint32_t>();
        as.tasks[k] = QS.tasks.as<Task>();
        as.counters[k] = QS.counters.as<uint32_t>();
        as.multi[k] = QS.multi.as<MultiBucket>();
        as.hot[k] = QS.hot.as<HotBucket>();
        as.buckets[k] = Q.S.buckets.template as<Pt>();
        as.partials[k] = Q.S.partials.template as<Pt>();
        as.hot_sums[k] = Q.S.hot_sums.template as<Pt>();
        as.hot_done[k] = Q.S.hot_done.template as<uint32_t>();
        if (k < nsets && Q.info.ntasks > ntasks) ntasks = Q.info.ntasks;
        if (k < nsets) {
            const MsmPlanInfo& QI = Q.info;
            const uint64_t mean = (QI.n * QI.W + QI.nbuckets - 1) / (QI.nbuckets ? QI.nbuckets : 1);
            if (2 * (uint64_t)QI.lmax < 3 * mean) {        // task cap below 1.5 x the mean load: most buckets are cut
                const uint32_t want = (QI.nbuckets + 63) / 64;
                if (want > cb_small) cb_small = want > CB_SMALL_MAX ? CB_SMALL_MAX : want;
            }
        }
    }
    const uint32_t ny = (uint32_t)nsets;
    T.begin(which ? "msm_accumulate_g2" : "msm_accumulate_g1", s);
    // (Round 6, measured and dropped: capping the kernel at two workgroups per CU with unused dynamic LDS -- 2 wavefronts per SIMD, 240
    //  VGPRs per SIMD left for the other queue's tail / grouping kernels to move in beside it -- changed neither the kernel alone nor
    //  the two-queue proof; the G2 kernel at ONE wavefront per SIMD (280 registers left) was 0.1-0.4 ms slower: profiles/r06_acc_occupancy_ab.txt.)
    hipLaunchKernelGGL(msm_accumulate<C>, dim3(ceil_div_u64(ntasks, 256), ny), dim3(256), 0, s, as);
    T.end(s);
    WS_HIP_CHECK(hipGetLastError());
    T.begin("msm_combine", s);
    hipLaunchKernelGGL(msm_combine_all<C>, dim3(cb_small + CB_WAVE + CB_HOT, ny), dim3(64), 0, s, as, cb_small);
    T.end(s);
    WS_HIP_CHECK(hipGetLastError());
    return WS_OK;
}

// C = device curve (arithmetic of the kernels), H = host curve (reference-format results, host tail).
// `prepared`: d_points are already in C's internal domain (msm_prepare_points).
// defer_kernels: only the launch slot and its buffers are set up; the caller accumulates several such launches in ONE batched
// launch (msm_acc_sets)
template <class C, class H>
static int msm_launch_acc(Lane& L, int which, const typename H::Aff* d_points_ref, bool prepared, int* slot_out, hipStream_t s, bool defer_kernels = false) {
    typedef typename C::PtP Pt;      // packed accumulator in global memory (same bytes as H::Pt)
    static_assert(sizeof(typename C::PtP) == sizeof(typename H::Pt) && sizeof(typename C::AffP) == sizeof(typename H::Aff), "layouts");
    Context* X = ctx();
    if (!X) return WS_ERR_NOINIT;
    if (!s) s = L.stream;
    MsmWorkspace& M = ws(L);
    const MsmPlanInfo& I = M.plan[M.cur].info;
    if (!I.valid) { set_last_error("msm: no plan"); return WS_ERR_ARG; }
    if (I.n && !d_points_ref) return WS_ERR_ARG;
    int slot = -1;
    for (int i = 0; i < kPendingSlots; i++) if (!M.slot[i].active) { slot = i; break; }
    if (slot < 0) { set_last_error("msm: too many unfinished launches"); return WS_ERR_ARG; }
    MsmPending& P = M.slot[slot];
    P.which = which;
    P.info = I;
    P.plan_id = M.cur;
    if (I.n == 0) { P.active = true; *slot_out = slot; return WS_OK; }   // multiexp with n=0 leaves pr unchanged
    const typename C::AffP* d_points = reinterpret_cast<const typename C::AffP*>(d_points_ref);
    const uint64_t n = I.n;
    const uint32_t W = I.tW, nbuckets = I.nbuckets, J = I.J, nsum = I.nsum;

    MsmScratch& S = P.S;                        // this launch's accumulation buffers (the plan's own are read by the kernels)
    // Everything runs in order on the caller's stream.  Tried and measured slower on MI355X (rounds 1, 3, 5): accumulations on
    // concurrent streams (cache thrash), and the reduction tail on another (or a high-priority) stream: docs/HISTORY.md, DESIGN.md section 5.
    WS_HIP_CHECK(S.buckets.reserve((size_t)nbuckets * sizeof(Pt)));
    WS_HIP_CHECK(S.partials.reserve((size_t)I.hot_cap * sizeof(Pt)));
    // (level-1 chunk pairs first; with a second chunk level its W * J pairs follow them in the same buffers)
    WS_HIP_CHECK(S.chunkS.reserve(((size_t)W * I.J1 + (I.m2 > 1 ? (size_t)2 * W * J : 0)) * sizeof(Pt)));      // S, then S' and W
    WS_HIP_CHECK(S.chunkA.reserve(((size_t)W * I.J1 + (I.m2 > 1 ? (size_t)W * J : 0)) * sizeof(Pt)));          // A, then A'
    // rows that reach the host: nrows per group when the pieces are folded on the GPU (their nsum rows per piece then stay in
    // d_rows), nsum per piece otherwise
    const size_t sums_bytes = (I.reduce ? (size_t)I.groups * I.nrows : (size_t)W * nsum) * sizeof(Pt);
    WS_HIP_CHECK(P.d_sums.reserve(sums_bytes));
    if (I.reduce) WS_HIP_CHECK(P.d_rows.reserve((size_t)W * nsum * sizeof(Pt)));
    WS_HIP_CHECK(S.tree_half.reserve((size_t)W * 6 * sizeof(Pt)));
    {
        const size_t db = (size_t)W * 3 * sizeof(uint32_t);
        if (!S.tree_done.p || S.tree_done.bytes < db) { WS_HIP_CHECK(S.tree_done.alloc(db)); WS_HIP_CHECK(hipMemsetAsync(S.tree_done.p, 0, db, s)); }
    }
    if (P.h_bytes < sums_bytes) {
        if (P.h_sums) (void)hipHostFree(P.h_sums);
        P.h_sums = nullptr; P.h_bytes = 0;
        WS_HIP_CHECK(hipHostMalloc(&P.h_sums, sums_bytes, 0));
        P.h_bytes = sums_bytes;
    }
    if (!P.ev) WS_HIP_CHECK(hipEventCreate(&P.ev));
    // (slots wait for hot-bucket slice sums: at most one per HOT_SLICE tasks plus one per hot bucket)
    WS_HIP_CHECK(S.hot_sums.reserve(((size_t)I.hot_cap / HOT_SLICE + (size_t)I.hot_cap / I.hot_min + 32) * sizeof(Pt)));
    // completion counters of the combine kernel's hot-bucket stage: cleared when (re)allocated, left at zero by every launch
    {
        const size_t hot_bytes = ((size_t)I.hot_cap / I.hot_min + 16) * 4;
        if (!S.hot_done.p || S.hot_done.bytes < hot_bytes) { WS_HIP_CHECK(S.hot_done.alloc(hot_bytes)); WS_HIP_CHECK(hipMemsetAsync(S.hot_done.p, 0, hot_bytes, s)); }
    }

    KernelTimer& T = X->timer;
    if (I.flat && !prepared) { set_last_error("msm: a table plan needs a prepared fixed-base table"); return WS_ERR_ARG; }
    if (C::Field::kInternalDomain && !prepared) {
        WS_HIP_CHECK(S.points_conv.reserve((size_t)n * sizeof(typename C::AffP)));
        T.begin("msm_convert_points", s);
        hipLaunchKernelGGL(msm_convert_points<C>, dim3(ceil_div_u64(n, 256)), dim3(256), 0, s, d_points,
                           S.points_conv.as<typename C::AffP>(), n);
        T.end(s);
        d_points = S.points_conv.as<typename C::AffP>();
    }
    P.d_points_used = d_points;
    if (!defer_kernels) {
        MsmPending* one[1] = {&P};
        int rc = msm_acc_sets<C>(L, one, 1, which, s);
        if (rc) return rc;
    }
    // the slot is taken only now: an error above (null points, a failed reserve) leaves it free
    P.active = true;
    *slot_out = slot;
    return WS_OK;
}

// reduction tail (chunks, trees, [piece reduction,] copy of the rows, completion event) for up to 4 launches of one geometry.
// G1 runs on the curve variant with inlined products, G2 on the lane-paired curve (fp2.h; same buffers, same results).
template <class C> struct TailCurve { typedef C type; typedef C paired; };
template <> struct TailCurve<G1R29> { typedef G1R29I type; typedef G1P29 paired; };      // (round 6: one G1 point on two lanes, curve_pair.h)
template <> struct TailCurve<G2R29> { typedef G2P29 type; typedef G2Q29 paired; };       // (round 6: one G2 point on FOUR lanes for msm_tree / msm_rows)
// WSNARK_TAIL_PAIR_G1 (default 1): the G1 reduction tails on lane pairs -- seven product steps per addition instead of fourteen
static bool tail_pair_g1() { return tuning_get("TAIL_PAIR_G1", 1) != 0; }
static bool tail_quad_g2() { return tuning_get("TAIL_QUAD_G2", 1) != 0; }
template <class C> static bool tail_split() { return sizeof(typename C::AffP) > 64 ? tail_quad_g2() : tail_pair_g1(); }
// CC: the curve of msm_chunks (throughput-bound: 2 additions per bucket on every lane of the chip -- the one-lane form is the cheaper
// one there), C: the curve of msm_tree / msm_rows (latency-bound LDS reductions: the lane-paired forms halve every step).  Same buffers.
template <class C, class CC = C>
static int msm_launch_tail(Lane& L, const int* slot_ids, int nslots, hipStream_t s) {
    typedef PointIO<C> IO;
    typedef typename IO::Stored St;
    static_assert(sizeof(typename PointIO<CC>::Stored) == sizeof(St), "the chunk and tree kernels share their buffers");
    Context* X = ctx();
    if (!s) s = L.stream;
    MsmPending* slots = ws(L).slot;
    const MsmPlanInfo& I = slots[slot_ids[0]].info;
    if (I.n == 0) return WS_OK;
    TailSets<C> ts;
    for (int k = 0; k < 4; k++) {
        MsmPending& P = slots[slot_ids[k < nslots ? k : 0]];
        ts.buckets[k] = P.S.buckets.template as<St>();
        ts.l1S[k] = P.S.chunkS.template as<St>();
        ts.l1A[k] = P.S.chunkA.template as<St>();
        ts.chunkS[k] = P.S.chunkS.template as<St>() + (I.m2 > 1 ? (size_t)I.tW * I.J1 : 0);
        ts.chunkA[k] = P.S.chunkA.template as<St>() + (I.m2 > 1 ? (size_t)I.tW * I.J1 : 0);
        ts.chunkW[k] = P.S.chunkS.template as<St>() + (size_t)I.tW * I.J1 + (size_t)I.tW * I.J;
        ts.rows[k] = P.d_rows.template as<St>();
        ts.sums[k] = P.d_sums.template as<St>();
        ts.bstart[k] = ws(L).plan[P.plan_id].S.bstart.template as<uint32_t>();
        ts.bend[k] = ws(L).plan[P.plan_id].S.bend.template as<uint32_t>();
        ts.half[k] = P.S.tree_half.template as<St>();
        ts.half_done[k] = P.S.tree_done.template as<uint32_t>();
    }
    const uint32_t J = I.J, logJ = I.logJ, nsum = I.nsum, LPP = IO::LPP, W = I.tW;
    KernelTimer& T = X->timer;
    T.begin("msm_chunks", s);
    {
        TailSets<CC> tc;
        static_assert(sizeof tc == sizeof ts, "same pointers");
        memcpy(&tc, &ts, sizeof tc);
        for (int k = 0; k < 4; k++) { tc.chunkS[k] = const_cast<typename PointIO<CC>::Stored*>(tc.l1S[k]); tc.chunkA[k] = const_cast<typename PointIO<CC>::Stored*>(tc.l1A[k]); }
        hipLaunchKernelGGL(msm_chunks<CC>, dim3(ceil_div_u64((uint64_t)W * I.J1 * PointIO<CC>::LPP, 256), nslots), dim3(256), 0, s, tc, W * I.J1, I.m1);
    }
    T.end(s);
    WS_HIP_CHECK(hipGetLastError());
    if (I.m2 > 1) {
        T.begin("msm_chunks2", s);
        hipLaunchKernelGGL(msm_chunks2<C>, dim3(ceil_div_u64((uint64_t)2 * W * J * LPP, 256), nslots), dim3(256), 0, s, ts, W * J, I.m2);
        T.end(s);
        WS_HIP_CHECK(hipGetLastError());
    }
    // one slot per element of the longest strided sum (J/2 for the masked rows); 256 slots per workgroup (at 152 VGPRs a 512-thread
    // workgroup is one per CU and a full-size tree launch takes three rounds of them; 256 threads fit three per CU: profiles/r04_s6_*)
    uint32_t tslots = 1;
    // (512 slots where the whole launch is one round of workgroups anyway -- a single G1 set at full size: 16 rows x 16 pieces --:
    //  4 + 9 dependent additions per lane instead of 8 + 8)
    const uint32_t tree_wgs = (logJ + 2 * (nsum - logJ)) * W * (uint32_t)nslots;
    const uint32_t smax = ((LPP == 1 || TreeBound<C>::value >= 1024) && tree_wgs <= 256) ? 512 : 256;
    while (tslots < (J > 1 ? J / 2 : 1) && tslots < smax && tslots * LPP < (uint32_t)TreeBound<C>::value) tslots <<= 1;
    T.begin("msm_tree", s);
    hipLaunchKernelGGL(msm_tree<C>, dim3(logJ + 2 * (nsum - logJ), W, nslots), dim3(tslots * LPP), (size_t)tslots * sizeof(St), s, ts, J, logJ, I.reduce ? 0u : 1u,
                       nsum);
    T.end(s);
    WS_HIP_CHECK(hipGetLastError());
    size_t sums_bytes;
    if (I.reduce) {
        uint32_t rslots = 1;
        while (rslots < I.tP && rslots < smax && rslots * LPP < (uint32_t)TreeBound<C>::value) rslots <<= 1;
        T.begin("msm_rows", s);
        hipLaunchKernelGGL(msm_rows<C>, dim3(I.nrows, I.groups, nslots), dim3(rslots * LPP), (size_t)rslots * sizeof(St), s, ts, I.tP, logJ, nsum, I.nrows, I.m2 > 1 ? I.nrows - 1 : 0xFFFFFFFFu);
        T.end(s);
        WS_HIP_CHECK(hipGetLastError());
        sums_bytes = (size_t)I.groups * I.nrows * sizeof(St);
    } else {
        sums_bytes = (size_t)W * nsum * sizeof(St);
    }
    for (int k = 0; k < nslots; k++) {
        MsmPending& P = slots[slot_ids[k]];
        WS_HIP_CHECK(hipMemcpyAsync(P.h_sums, P.d_sums.p, sums_bytes, hipMemcpyDeviceToHost, s));
        WS_HIP_CHECK(hipEventRecord(P.ev, s));
    }
    return WS_OK;
}

template <class C, class H>
static int msm_launch(Lane& L, int which, const typename H::Aff* d_points_ref, bool prepared, int* slot_out, hipStream_t s) {
    if (!s) s = L.stream;
    int rc = msm_launch_acc<C, H>(L, which, d_points_ref, prepared, slot_out, s);
    if (rc) return rc;
    rc = tail_split<C>() ? msm_launch_tail<typename TailCurve<C>::paired, typename TailCurve<C>::type>(L, slot_out, 1, s)
                         : msm_launch_tail<typename TailCurve<C>::type>(L, slot_out, 1, s);
    if (rc) msm_abort_slots(L, slot_out, 1, s, nullptr);
    return rc;
}

// several G1 point sets against the current plan (or, with plan_ids, against variants of one plan: same geometry): accumulations
// back to back -- small sums: in ONE launch --, then ONE batched tail
int msm_g1_launch_batch(Lane& L, const Affine<Fq>* const* d_points, int nsets, bool prepared, int* slots, hipStream_t s, const int* plan_ids) {
    if (!ctx()) return WS_ERR_NOINIT;
    if (nsets < 1 || nsets > 4) return WS_ERR_ARG;
    if (!s) s = L.stream;
    MsmWorkspace& M = ws(L);
    const int keep = M.cur;
    for (int k = 0; k < nsets; k++) slots[k] = -1;
    int rc = WS_OK;
    // Very small sums (fewer bucket runs than the chip has lanes in one wavefront per SIMD, see msm_plan_begin): the sets are
    // accumulated by ONE launch (blockIdx.y = set) and run beside each other; from 2^16 runs on one launch per set, back to
    // back, measures better (same box, profiles/r03_s14_small_sums.txt: a rank's share of a 2^20 key over 8 ranks 2.08 vs 2.33 ms,
    // 2^18 proofs 4.15 vs 4.45 ms, 2^19 6.2 vs 6.43 ms; 2^14 / 2^16 proofs the other way: 1.55 vs 2.1 ms, 1.93 vs 2.4 ms with the
    // shorter tasks)
    const MsmPlanInfo& I0 = M.plan[plan_ids ? plan_ids[0] : M.cur].info;
    const bool batched = nsets > 1 && I0.valid && I0.n && I0.nbuckets <= (1u << 16);
    for (int k = 0; k < nsets && !rc; k++) {
        if (plan_ids) msm_select_plan(L, plan_ids[k]);
        rc = msm_launch_acc<G1R29, G1>(L, 0, d_points[k], prepared, &slots[k], s, batched);
        if (plan_ids) msm_select_plan(L, keep);
    }
    if (!rc && batched) {
        MsmPending* Ps[4];
        for (int k = 0; k < nsets; k++) Ps[k] = &M.slot[slots[k]];
        rc = msm_acc_sets<G1R29>(L, Ps, nsets, 0, s);
    }
    if (!rc) rc = tail_pair_g1() ? msm_launch_tail<TailCurve<G1R29>::paired, TailCurve<G1R29>::type>(L, slots, nsets, s)
                                 : msm_launch_tail<TailCurve<G1R29>::type>(L, slots, nsets, s);
    if (rc) msm_abort_slots(L, slots, nsets, s, nullptr);
    return rc;
}

int msm_g1_launch(Lane& L, const Affine<Fq>* d_points, bool prepared, int* slot, hipStream_t s) {
    if (!ctx()) return WS_ERR_NOINIT;
    return msm_launch<G1R29, G1>(L, 0, d_points, prepared, slot, s);
}
int msm_g2_launch(Lane& L, const Affine<Fq2>* d_points, bool prepared, int* slot, hipStream_t s) {
    if (!ctx()) return WS_ERR_NOINIT;
    return msm_launch<G2R29, G2>(L, 1, d_points, prepared, slot, s);
}
static MsmPending* live_slot(Lane& L, int slot, int which) {
    if (!L.msm || slot < 0 || slot >= kPendingSlots) return nullptr;
    MsmPending& P = L.msm->slot[slot];
    return (P.active && (which < 0 || P.which == which)) ? &P : nullptr;
}
bool msm_ready(Lane& L, int slot) {
    MsmPending* P = live_slot(L, slot, -1);
    return P && (P->info.n == 0 || (P->ev && hipEventQuery(P->ev) == hipSuccess));
}
int msm_g1_finish(Lane& L, int slot, XYZZ<Fq>* out_host) {
    MsmPending* P = live_slot(L, slot, 0);
    return P ? msm_finish_t<G1>(*P, out_host) : WS_ERR_ARG;
}
int msm_g2_finish(Lane& L, int slot, XYZZ<Fq2>* out_host) {
    MsmPending* P = live_slot(L, slot, 1);
    return P ? msm_finish_t<G2>(*P, out_host) : WS_ERR_ARG;
}

// Which pairs can be left out of a sum whatever the scalar: those whose point is infinity in EVERY given set
// (x == 0, reference format).  Real circuits leave many variables out of the B matrix, so B1 / B2 are sparse;
// the prover groups the witness once more with this mask for those two sums.
__global__ __launch_bounds__(256) void msm_points_mask_kernel(const Affine<Fq>* __restrict__ g1, const Affine<Fq2>* __restrict__ g2,
                                                                uint64_t n, uint8_t* __restrict__ mask, uint32_t* __restrict__ skipped) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool inf = true;
    if (g1) inf = inf && Fq::is_zero(g1[i].x);
    if (g2) inf = inf && Fq::is_zero(g2[i].x.c0) && Fq::is_zero(g2[i].x.c1);
    mask[i] = inf ? 0 : 1;
    if (inf) atomicAdd(skipped, 1u);
}
int msm_points_mask(const Affine<Fq>* d_g1, const Affine<Fq2>* d_g2, uint64_t n, uint8_t* d_mask, uint32_t* skipped_host, hipStream_t s) {
    Context* X = ctx();
    if (!X) return WS_ERR_NOINIT;
    if (!s) s = X->stream;
    *skipped_host = 0;
    if (n == 0) return WS_OK;
    DevBuf cnt;
    WS_HIP_CHECK(cnt.alloc(4));
    WS_HIP_CHECK(hipMemsetAsync(cnt.p, 0, 4, s));
    hipLaunchKernelGGL(msm_points_mask_kernel, dim3(ceil_div_u64(n, 256)), dim3(256), 0, s, d_g1, d_g2, n, d_mask, cnt.as<uint32_t>());
    WS_HIP_CHECK(hipGetLastError());
    WS_HIP_CHECK(hipMemcpyAsync(skipped_host, cnt.p, 4, hipMemcpyDeviceToHost, s));
    WS_HIP_CHECK(hipStreamSynchronize(s));
    return WS_OK;
}

// in-place conversion of a resident point array (the proving key's sections) to the device field's
// internal domain, so that proofs skip the per-MSM conversion pass
int msm_prepare_points(int which, void* d_points, uint64_t n, hipStream_t s) {
    Context* X = ctx();
    if (!X) return WS_ERR_NOINIT;
    if (!s) s = X->stream;
    if (n == 0) return WS_OK;
    if (which == 0) {
        hipLaunchKernelGGL(msm_convert_points<G1R29>, dim3(ceil_div_u64(n, 256)), dim3(256), 0, s,
                           (const G1R29::AffP*)d_points, (G1R29::AffP*)d_points, n);
    } else {
        hipLaunchKernelGGL(msm_convert_points<G2R29>, dim3(ceil_div_u64(n, 256)), dim3(256), 0, s,
                           (const G2R29::AffP*)d_points, (G2R29::AffP*)d_points, n);
    }
    WS_HIP_CHECK(hipGetLastError());
    return WS_OK;
}

// ---- fixed-base window tables (resident proving keys) ----
// table = rows x n affine points in the device field's domain (row 0 given, msm_prepare_points): row w = 2^(c w) * row 0.
// One lane per point walks the rows with c doublings each (XYZZ, never normalised in between) and normalises
// TABLE_GROUP rows with one shared inversion (Montgomery's trick on the ZZZ coordinates; 1/ZZ = ZZ^2 / ZZZ^2).
// Key-load time only.
// Rows normalised per shared inversion.  12 = all the rows of a 13-row table (c = 20) behind ONE Fermat inversion per point instead
// of two (8 + 4): the inversion is ~380 products against 12 x 20 doublings x 9 = 2 160, so about -12 % of the build (the round-3
// review's item 7; the group lives in scratch memory either way -- this kernel runs at key-load time only).
#ifndef WS_TABLE_GROUP
#define WS_TABLE_GROUP 12
#endif
static const uint32_t TABLE_GROUP = WS_TABLE_GROUP;
template <class C>
__global__ __launch_bounds__(256) void msm_table_kernel(typename C::AffP* __restrict__ table, uint64_t n, uint32_t c, uint32_t rows) {
    typedef typename C::Field F;
    typedef typename F::El El;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const typename C::AffP p0p = table[i];
    const typename C::Aff p0 = C::unpack_aff(p0p);
    if (C::aff_is_inf(p0)) {
        for (uint32_t w = 1; w < rows; w++) table[(uint64_t)w * n + i] = p0p;     // x == 0: infinity in every row
        return;
    }
    typename C::Pt P = C::from_affine(p0);
    typename C::Pt pts[TABLE_GROUP];
    El pre[TABLE_GROUP];
    for (uint32_t w0 = 1; w0 < rows; w0 += TABLE_GROUP) {
        const uint32_t g = rows - w0 < TABLE_GROUP ? rows - w0 : TABLE_GROUP;
        El run = F::one();
        for (uint32_t k = 0; k < g; k++) {
            for (uint32_t d = 0; d < c; d++) P = C::dbl(P);       // (a point of odd prime order never doubles to infinity)
            pts[k] = P;
            pre[k] = run;
            run = F::mul(run, P.zzz);
        }
        El inv = F::inv(run);
        for (int k = (int)g - 1; k >= 0; k--) {
            const El izzz = F::mul(inv, pre[k]);
            inv = F::mul(inv, pts[k].zzz);
            const El izz = F::mul(F::sqr(izzz), F::sqr(pts[k].zz));
            table[(uint64_t)(w0 + k) * n + i] = C::pack_aff(typename C::Aff{F::mul(pts[k].x, izz), F::mul(pts[k].y, izzz)});
        }
    }
}
// The same build cut into short launches (round 4, for the background build: a workgroup of the kernel above lives for ~2 ms --
// 240 doublings and an inversion per lane -- and a proof kernel that arrives meanwhile waits for such workgroups to retire before
// its own fit; below a workgroup lives for one row, 20 doublings).  Step k takes the running XYZZ point of every lane from slot
// `src` of the scratch slab (row 0 of the table when src < 0), doubles it c times and leaves it in slot `dst`; the closing launch
// normalises the g slots behind one inversion per lane, exactly as the one-kernel build does (same products in the same order:
// the tables come out bit-identical).  Scratch: TABLE_GROUP slots x S lanes of XYZZ points.
template <class C>
__global__ __launch_bounds__(256) void msm_table_step_kernel(const typename C::AffP* __restrict__ row0, typename C::Pt* __restrict__ tmp, uint64_t cnt,
                                                             uint64_t S, uint32_t c, int src, uint32_t dst) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cnt) return;
    const typename C::Aff p0 = C::unpack_aff(row0[i]);
    if (C::aff_is_inf(p0)) return;
    typename C::Pt P = src < 0 ? C::from_affine(p0) : tmp[(uint64_t)src * S + i];
    for (uint32_t d = 0; d < c; d++) P = C::dbl(P);
    tmp[(uint64_t)dst * S + i] = P;
}
template <class C>
__global__ __launch_bounds__(256) void msm_table_norm_kernel(typename C::AffP* __restrict__ table, uint64_t n, uint64_t base, uint64_t cnt,
                                                             const typename C::Pt* __restrict__ tmp, uint64_t S, uint32_t w0, uint32_t g) {
    typedef typename C::Field F;
    typedef typename F::El El;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cnt) return;
    const typename C::AffP p0p = table[base + i];
    if (C::aff_is_inf(C::unpack_aff(p0p))) {
        for (uint32_t k = 0; k < g; k++) table[(uint64_t)(w0 + k) * n + base + i] = p0p;
        return;
    }
    El pre[TABLE_GROUP];
    El run = F::one();
    for (uint32_t k = 0; k < g; k++) {
        pre[k] = run;
        run = F::mul(run, tmp[(uint64_t)k * S + i].zzz);
    }
    El inv = F::inv(run);
    for (int k = (int)g - 1; k >= 0; k--) {
        const typename C::Pt q = tmp[(uint64_t)k * S + i];
        const El izzz = F::mul(inv, pre[k]);
        inv = F::mul(inv, q.zzz);
        const El izz = F::mul(F::sqr(izzz), F::sqr(q.zz));
        table[(uint64_t)(w0 + k) * n + base + i] = C::pack_aff(typename C::Aff{F::mul(q.x, izz), F::mul(q.y, izzz)});
    }
}
size_t msm_table_scratch_bytes(uint64_t lanes) { return (size_t)TABLE_GROUP * lanes * sizeof(G2R29::Pt); }
template <class C>
static void table_stepped(typename C::AffP* table, uint64_t n, uint32_t tc, uint32_t rows, hipStream_t s, void* d_tmp, size_t tmp_bytes) {
    typename C::Pt* tmp = (typename C::Pt*)d_tmp;
    uint64_t S = tmp_bytes / ((size_t)TABLE_GROUP * sizeof(typename C::Pt));
    S &= ~(uint64_t)63;
    for (uint64_t base = 0; base < n; base += S) {
        const uint64_t cnt = n - base < S ? n - base : S;
        const dim3 grid(ceil_div_u64(cnt, 256));
        int src = -1;
        for (uint32_t w0 = 1; w0 < rows; w0 += TABLE_GROUP) {
            const uint32_t g = rows - w0 < TABLE_GROUP ? rows - w0 : TABLE_GROUP;
            for (uint32_t k = 0; k < g; k++) {
                hipLaunchKernelGGL(msm_table_step_kernel<C>, grid, dim3(256), 0, s, table + base, tmp, cnt, S, tc, src, k);
                src = (int)k;
            }
            hipLaunchKernelGGL(msm_table_norm_kernel<C>, grid, dim3(256), 0, s, table, n, base, cnt, tmp, S, w0, g);
        }
    }
}
// d_table: rows x n points, row 0 already in the device domain (msm_prepare_points).  d_tmp (msm_table_scratch_bytes of at least
// 64 lanes; may be NULL): build in short launches through that scratch instead of one long kernel
int msm_build_table(int which, void* d_table, uint64_t n, uint32_t tc, hipStream_t s, void* d_tmp, size_t tmp_bytes) {
    Context* X = ctx();
    if (!X) return WS_ERR_NOINIT;
    if (!s) s = X->stream;
    const uint32_t rows = msm_table_rows(tc);
    if (n == 0 || rows < 2) return WS_OK;
    if (d_tmp && tmp_bytes >= msm_table_scratch_bytes(64)) {
        if (which == 0) table_stepped<G1R29>((G1R29::AffP*)d_table, n, tc, rows, s, d_tmp, tmp_bytes);
        else table_stepped<G2R29>((G2R29::AffP*)d_table, n, tc, rows, s, d_tmp, tmp_bytes);
    } else if (which == 0)
        hipLaunchKernelGGL(msm_table_kernel<G1R29>, dim3(ceil_div_u64(n, 256)), dim3(256), 0, s, (G1R29::AffP*)d_table, n, tc, rows);
    else
        hipLaunchKernelGGL(msm_table_kernel<G2R29>, dim3(ceil_div_u64(n, 256)), dim3(256), 0, s, (G2R29::AffP*)d_table, n, tc, rows);
    WS_HIP_CHECK(hipGetLastError());
    return WS_OK;
}

// The conversion of caller points to the device field's internal domain does not depend on the plan: it runs on
// the lane's second queue while the first one groups the digits (returns the array the accumulation should read).
template <class CD, class AffT>
static int convert_beside_plan(Lane& L, const AffT* d_points, uint64_t n, hipStream_t s, const AffT** out, bool* prepared) {
    Context* X = ctx();
    *out = d_points;
    if (*prepared || !CD::Field::kInternalDomain || n < (1u << 14) || s == L.stream2) return WS_OK;
    MsmWorkspace& M = ws(L);
    for (auto& e : M.conv_ev) if (!e) WS_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    WS_HIP_CHECK(M.conv_buf.reserve((size_t)n * sizeof(typename CD::AffP)));
    WS_HIP_CHECK(hipEventRecord(M.conv_ev[0], s));                 // the caller's points are ready on s
    WS_HIP_CHECK(hipStreamWaitEvent(L.stream2, M.conv_ev[0], 0));
    X->timer.begin("msm_convert_points", L.stream2);
    hipLaunchKernelGGL(msm_convert_points<CD>, dim3(ceil_div_u64(n, 256)), dim3(256), 0, L.stream2,
                       reinterpret_cast<const typename CD::AffP*>(d_points), M.conv_buf.as<typename CD::AffP>(), n);
    X->timer.end(L.stream2);
    WS_HIP_CHECK(hipEventRecord(M.conv_ev[1], L.stream2));
    *out = M.conv_buf.as<AffT>();
    *prepared = true;
    return WS_OK;
}

// one whole MSM from device pointers on lane L: conversion beside the plan, launch, host tail
template <class CD, class H>
static int msm_dev_t(Lane& L, int which, const Fe* d_scalars, const typename H::Aff* d_points, uint64_t n, WindowShard sh,
                     typename H::Pt* out_host, hipStream_t s) {
    if (!ctx()) return WS_ERR_NOINIT;
    if (n && !d_points) return WS_ERR_ARG;
    if (!s) s = L.stream;
    const typename H::Aff* pts = d_points;
    bool prepared = false;
    msm_select_plan(L, 0);
    int rc = convert_beside_plan<CD>(L, d_points, n, s, &pts, &prepared);
    if (rc) return rc;
    if ((rc = msm_plan_dev(L, d_scalars, n, sh, s, 0))) return rc;
    if (prepared) WS_HIP_CHECK(hipStreamWaitEvent(s, ws(L).conv_ev[1], 0));
    int slot = -1;
    rc = which == 0 ? msm_g1_launch(L, reinterpret_cast<const Affine<Fq>*>(pts), prepared, &slot, s)
                    : msm_g2_launch(L, reinterpret_cast<const Affine<Fq2>*>(pts), prepared, &slot, s);
    if (rc) return rc;
    return msm_finish_t<H>(ws(L).slot[slot], out_host);
}
int msm_g1_dev(Lane& L, const Fe* d_scalars, const Affine<Fq>* d_points, uint64_t n, WindowShard sh, Jac<Fq>* out_host, hipStream_t s) {
    XYZZ<Fq> r;
    int rc = msm_dev_t<G1R29, G1>(L, 0, d_scalars, d_points, n, sh, &r, s);
    if (rc) return rc;
    *out_host = G1::to_affine_jac(r);
    return WS_OK;
}
int msm_g2_dev(Lane& L, const Fe* d_scalars, const Affine<Fq2>* d_points, uint64_t n, WindowShard sh, Jac<Fq2>* out_host, hipStream_t s) {
    XYZZ<Fq2> r;
    int rc = msm_dev_t<G2R29, G2>(L, 1, d_scalars, d_points, n, sh, &r, s);
    if (rc) return rc;
    *out_host = G2::to_affine_jac(r);
    return WS_OK;
}
// Host-pointer boundary (what the reference's g1_multiexp / g2_multiexp callers hold: plain host buffers).
// Scalars go up first; the plan is built on the GPU while worker threads stage the points on the second queue.
template <class H, class JacT>
static int msm_host_t(Lane& L, int which, const void* h_scalars, const void* h_points, uint64_t n, WindowShard sh, JacT* out_host) {
    if (!ctx()) return WS_ERR_NOINIT;
    if (n == 0) { *out_host = H::to_affine_jac(H::infinity()); return WS_OK; }
    if (!h_scalars || !h_points) return WS_ERR_ARG;
    hipStream_t s = L.stream, s2 = L.stream2;
    MsmWorkspace& M = ws(L);
    WS_HIP_CHECK(L.host_in[0].reserve((size_t)n * 32));
    WS_HIP_CHECK(L.host_in[1].reserve((size_t)n * sizeof(typename H::Aff)));
    if (!M.host_ev) WS_HIP_CHECK(hipEventCreateWithFlags(&M.host_ev, hipEventDisableTiming));
    int rc = upload_staged(L.host_in[0].p, h_scalars, (size_t)n * 32, s);
    if (rc) return rc;
    msm_select_plan(L, 0);
    if ((rc = msm_plan_dev(L, L.host_in[0].as<Fe>(), n, sh, s, 0))) return rc;
    if ((rc = upload_staged(L.host_in[1].p, h_points, (size_t)n * sizeof(typename H::Aff), s2))) return rc;
    WS_HIP_CHECK(hipEventRecord(M.host_ev, s2));
    WS_HIP_CHECK(hipStreamWaitEvent(s, M.host_ev, 0));
    typename H::Pt r;
    int slot = -1;
    rc = which == 0 ? msm_g1_launch(L, reinterpret_cast<const Affine<Fq>*>(L.host_in[1].p), false, &slot, s)
                    : msm_g2_launch(L, reinterpret_cast<const Affine<Fq2>*>(L.host_in[1].p), false, &slot, s);
    if (rc) return rc;
    if ((rc = msm_finish_t<H>(M.slot[slot], &r))) return rc;
    *out_host = H::to_affine_jac(r);
    return WS_OK;
}
int msm_g1_host(Lane& L, const void* h_scalars, const void* h_points, uint64_t n, WindowShard sh, Jac<Fq>* out_host) {
    return msm_host_t<G1, Jac<Fq>>(L, 0, h_scalars, h_points, n, sh, out_host);
}
int msm_g2_host(Lane& L, const void* h_scalars, const void* h_points, uint64_t n, WindowShard sh, Jac<Fq2>* out_host) {
    return msm_host_t<G2, Jac<Fq2>>(L, 1, h_scalars, h_points, n, sh, out_host);
}

WS_DEFINE_WARM(msm)

}  // namespace wsnark
