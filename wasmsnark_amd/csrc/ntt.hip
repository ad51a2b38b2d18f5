// ntt.hip -- radix-2 NTT / iNTT over BN128 Fr for gfx950.
//
// Replaces SURVEY.md section 8a rows a15-a17 (/root/reference src/build_fft.js:
// __rawfft :223-372, fft_fft :159-187, __finalInverse :550-648, fft_ifft :189-221,
// __reversePermutation :650-715).  Same function, different algorithm: the reference
// bit-reverses and then runs log2(n) in-place DIT stages over the whole array from one
// thread; here the transform is factored n = L0*L1*...(<= 4 digits of <= 2^10) and each
// digit is one kernel pass whose length-L sub-transforms run entirely in LDS:
//
//   pass p (not last): tile = L_p x T elements {hi*L*S + j*S + lo0+t}: T-element (T*32 B)
//        coalesced runs, DIF butterflies in LDS, inter-digit twiddle w_N^(c*k*i_rest) applied
//        on the way out, stored back to the same places;
//   last pass: tile = T values of the first digit x L contiguous elements; results are
//        written to their final natural-order (digit-reversed) position in T-element runs,
//        so no separate permutation pass exists.
//   first pass reads the caller's buffer and writes a scratch buffer, the last pass reads
//   scratch and writes the caller's buffer => in place for the caller, no extra copy.
//
// HBM traffic: np passes x (32 B read + 32 B write) per element, np = 3 at 2^22.
// Arithmetic is exact, so outputs are bit-identical to the reference's.
#include <atomic>

#include "internal.h"
#include "field29.h"

namespace wsnark {

struct alignas(16) Q128 {
    uint64_t a, b;
};

struct PassArgs {
    const Fe* in;
    Fe* out;
    uint32_t log_n, log_L, log_S, log_T;
    uint32_t is_last, log_S0, log_L0;         // last pass: stride and size of digit 0
    uint32_t np, kdig[4];
    const Fe* tw_small; uint32_t log_lmax;    // w_Lmax^j, j < Lmax/2 (direction-specific)
    const struct TwLimbs* tw_small29;         // the same table as nine 29-bit limbs per entry (radix-2^29 path: no unpacking per butterfly)
    uint32_t lds_tw;                          // round 6: the L/2 small twiddles of this pass are staged in LDS behind the tile (L <= 256)
    const Fe* tw_lo; const Fe* tw_hi; uint32_t h;   // two-level w_N^e = tw_hi[e>>h] * tw_lo[e & mask]
    uint32_t apply_twiddle;
    const Fe* cs_lo; const Fe* cs_hi; uint32_t hc; uint32_t prescale;   // coset factors w_2N^i
    // prescale == 2 (distributed column step, dist.hip): transform b of the batch is row (b & pre_row_mask) of a rank's block of a
    // LONGER vector, its element g sits at position pre_row0 + row + (g << pre_shift) of that vector, and cs_* are THAT length's tables
    uint32_t pre_shift, pre_row0, pre_row_mask;
    uint32_t scale, first;
    Fe out_scale;   // reference-format Montgomery form of 1/n (inverse) or 1
    // radix-2^29 path, >= 2 passes: one full inter-digit twiddle table per pass (index kk*S + i_rest) that also
    // carries the domain conversions: x2^5 in pass 0 (fold_in: the loaded values are used as they are) and
    // x2^-5 [and 1/n] in the second-to-last pass (fold_out: the last pass stores without a product)
    const Fe* tw_full;
    uint32_t fold_in, out_plain;
    // CALC_H fusions (calch.hip): first pass loads in[g] * in2[g] (the pointwise product of two evaluation vectors);
    // last pass stores h[t] = fromMontgomery((e[t] - w_2n^-t * v) / 2) instead of v.  fold_in == 2: the loaded product is
    // 2^-5 short of the reference form AND 2^5 short of the internal one: the pass-0 table carries 2^10.
    const Fe* in2;
    const Fe* comb_e; const Fe* comb_lo; const Fe* comb_hi; uint32_t comb_hc;
    Fe k271;        // 2^10 in the internal domain (raw 2^271 mod r): product-on-load without a folding table
    Fe comb_c;      // epilogue constant: radix-2^29 path raw 16 (x 2^-257 = halve + leave Montgomery), 4x64 path 1/2 (Montgomery)
    // POST instantiation (distributed column step, dist.hip): the last pass multiplies element c of transform b = (vector, row r)
    // by the four-step twiddle w_N^(+-(post_row0 + r) c) of the LONGER length N (post_lo / post_hi: two-level, internal form) and
    // stores it where the exchange wants it -- post_out[q][vector][r][c2], c = q 2^lr2 + c2 -- instead of a pass of its own.
    Fe* post_out; const Fe* post_lo; const Fe* post_hi;
    uint32_t post_h, post_lr1, post_lr2, post_k;
    uint64_t post_row0;
    // GATHER instantiation (distributed row step): transform b = (vector << lr2 | c2) loads its element i1 = q 2^lr1 + r from the
    // exchange's receive buffer gather_in[q][vector][r][c2] -- the transposition without a pass (and a buffer) of its own
    const Fe* gather_in;
};
enum { NTT_PLAIN = 0, NTT_POST = 1, NTT_GATHER = 2 };

// h[t] from the inverse transform's value v (reference Montgomery form, canonical) and e[t]: the CALC_H epilogue
//   h[t] = fromMontgomery((e[t] - w_2n^-t o[t]) / 2),   w_2n^-t = -w_2n^(n-t) for t >= 1   (derivation: calch.hip header)
template <class F> struct CombineEpilogue;
template <class P> struct CombineEpilogue<Field<P>> {
    __device__ static __forceinline__ Fe run(const PassArgs& A, const Fe& v, uint64_t t) {
        typedef Field<P> F;
        const Fe e = A.comb_e[t];
        Fe x;
        if (t == 0) {
            x = F::sub(e, v);
        } else {
            const uint64_t k = ((uint64_t)1 << A.log_n) - t;
            x = F::add(e, F::mul(F::mul(A.comb_hi[k >> A.comb_hc], A.comb_lo[k & (((uint64_t)1 << A.comb_hc) - 1)]), v));
        }
        return F::from_mont(F::mul(x, A.comb_c));
    }
};
template <class P> struct CombineEpilogue<Field29<P>> {
    // v: canonical reference-form value held in limbs.  comb_hi / comb_lo: w_2n^k two-level, both in the INTERNAL form
    // (x 2^261), so mul(mul(hi, lo), v) is w_2n^k * v in the reference form again; comb_c = 16: x 2^-257.
    __device__ static __forceinline__ Fe run(const PassArgs& A, const F29& v, uint64_t t) {
        typedef Field29<P> F;
        const F29 e = F::unpack(A.comb_e[t]);
        F29 x;
        if (t == 0) {
            x = F::sub(e, v);
        } else {
            const uint64_t k = ((uint64_t)1 << A.log_n) - t;
            const F29 f = F::mul(F::unpack(A.comb_hi[k >> A.comb_hc]), F::unpack(A.comb_lo[k & (((uint64_t)1 << A.comb_hc) - 1)]));
            x = F::add(e, F::mul(f, v));
        }
        return F::pack(F::canonical(F::mul(x, F::unpack(A.comb_c))));
    }
};

// A small-twiddle entry of the radix-2^29 path: the nine limbs as they are used, 48-byte stride (three 16-byte loads)
struct alignas(16) TwLimbs { uint32_t v[12]; };
template <class F> struct SmallTw;
template <class P> struct SmallTw<Field<P>> {
    __device__ static __forceinline__ Fe get(const PassArgs& A, uint32_t i) { return A.tw_small[i]; }
};
template <class P> struct SmallTw<Field29<P>> {
    __device__ static __forceinline__ F29 get(const PassArgs& A, uint32_t i) {
        const TwLimbs t = A.tw_small29[i];
        return F29{{t.v[0], t.v[1], t.v[2], t.v[3], t.v[4], t.v[5], t.v[6], t.v[7], t.v[8]}};
    }
};

// LDS tile storage: 16-byte planes so that consecutive lanes read consecutive 16-byte slots.
// 4x64 field: two planes (32 B/element).  Radix-2^29 field: limbs 0-3, limbs 4-7 and limb 8 in
// three planes (36 B/element), i.e. elements stay unpacked between butterflies.
template <class F> struct LdsTile;
template <class P> struct LdsTile<Field<P>> {
    static constexpr uint32_t kBytes = 32;
    Q128 *plo, *phi;
    __device__ __forceinline__ LdsTile(unsigned char* base, uint32_t LT) : plo((Q128*)base), phi((Q128*)base + LT) {}
    __device__ __forceinline__ Fe get(uint32_t i) const { Q128 a = plo[i], b = phi[i]; return Fe{{a.a, a.b, b.a, b.b}}; }
    __device__ __forceinline__ void put(uint32_t i, const Fe& v) { plo[i] = Q128{v.l[0], v.l[1]}; phi[i] = Q128{v.l[2], v.l[3]}; }
};
struct alignas(16) U4 { uint32_t x, y, z, w; };
template <class P> struct LdsTile<Field29<P>> {
    static constexpr uint32_t kBytes = 36;
    U4 *p0, *p1; uint32_t* p2;
    __device__ __forceinline__ LdsTile(unsigned char* base, uint32_t LT) : p0((U4*)base), p1((U4*)base + LT), p2((uint32_t*)((U4*)base + 2 * LT)) {}
    __device__ __forceinline__ F29 get(uint32_t i) const {
        U4 a = p0[i], b = p1[i];
        return F29{{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, p2[i]}};
    }
    __device__ __forceinline__ void put(uint32_t i, const F29& v) {
        p0[i] = U4{v.v[0], v.v[1], v.v[2], v.v[3]};
        p1[i] = U4{v.v[4], v.v[5], v.v[6], v.v[7]};
        p2[i] = v.v[8];
    }
};

// The pass's own small twiddles w_L^j, j < L/2, staged in LDS once per workgroup (round 6; VERDICT r5 "what's weak" 1: every butterfly
// pulled its twiddle from global memory as three 16-byte loads -- ~500 cycles of latency inside the butterfly loop against ~64 from
// LDS).  Same three planes as the tile; lanes of a wavefront read few distinct entries (LDS broadcasts equal addresses).
template <class F> struct LdsTw;
template <class P> struct LdsTw<Field<P>> {
    static constexpr uint32_t kBytes = 0;
    __device__ __forceinline__ LdsTw(unsigned char*, uint32_t) {}
    __device__ __forceinline__ void fill(const PassArgs&, uint32_t, uint32_t, uint32_t, uint32_t) {}
    __device__ __forceinline__ Fe get(uint32_t) const { return Fe{}; }
};
template <class P> struct LdsTw<Field29<P>> {
    static constexpr uint32_t kBytes = 36;
    U4 *p0, *p1; uint32_t* p2;
    __device__ __forceinline__ LdsTw(unsigned char* base, uint32_t half) : p0((U4*)base), p1((U4*)base + half), p2((uint32_t*)((U4*)base + 2 * half)) {}
    __device__ __forceinline__ void fill(const PassArgs& A, uint32_t half, uint32_t shift, uint32_t tid, uint32_t nthr) {
        for (uint32_t j = tid; j < half; j += nthr) {
            const TwLimbs t = A.tw_small29[j << shift];
            p0[j] = U4{t.v[0], t.v[1], t.v[2], t.v[3]};
            p1[j] = U4{t.v[4], t.v[5], t.v[6], t.v[7]};
            p2[j] = t.v[8];
        }
    }
    __device__ __forceinline__ F29 get(uint32_t j) const {
        const U4 a = p0[j], b = p1[j];
        return F29{{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, p2[j]}};
    }
};

// (Round 6, measured and dropped: the same kernel on the inlined-product field -- 33 product bodies, 8 898 VALU instructions, a code
//  object beyond the instruction cache -- ran the 2^20 passes in 73.8 us per launch against 74.0 with the product as a call:
//  profiles/r06_ntt_experiments.txt.)

// F = Field<FrParams> (saturated 4x64, values canonical, tables in the reference Montgomery form) or
// Field29<Fr29Params> (values in [0,2p), internal domain R' = 2^261; the host scales every table by
// 2^5 so that table products land in the internal domain, and the conversions from / to the
// reference format are folded into the first load and the last store).
template <class F, int MODE>
__global__ __launch_bounds__(512) void ntt_pass_kernel(PassArgs A) {
    constexpr bool POST = MODE == NTT_POST;
    typedef typename F::El El;
    WS_DYN_SMEM(unsigned char, sm);
    const uint32_t log_L = A.log_L, log_T = A.log_T;
    const uint32_t L = 1u << log_L, T = 1u << log_T, LT = L << log_T;
    LdsTile<F> tile(sm, LT);
    const uint32_t tid = threadIdx.x, nthr = blockDim.x;
    const bool lds_tw = A.lds_tw != 0;
    const uint32_t tw_shift = A.log_lmax - log_L;               // table index of w_L^j = j << tw_shift
    LdsTw<F> ltw(sm + (size_t)LT * LdsTile<F>::kBytes, L >> 1);
    if (lds_tw) ltw.fill(A, L >> 1, tw_shift, tid, nthr);       // (visible after the load phase's barrier)
    const uint32_t w = blockIdx.x;
    // batched form: blockIdx.y selects one of several independent transforms stored back to back
    const Fe* __restrict__ src = A.in + ((uint64_t)blockIdx.y << A.log_n);
    Fe* __restrict__ dst = A.out + ((uint64_t)blockIdx.y << A.log_n);

    // ---- tile coordinates ----
    uint64_t base = 0;       // non-last: hi*L*S + lo0
    uint32_t lo0 = 0;        // non-last: first lower-digit value of the tile
    uint32_t a0 = 0, mid = 0;
    if (!A.is_last) {
        const uint32_t tiles_per_block = 1u << (A.log_S - log_T);
        const uint32_t hi = w >> (A.log_S - log_T);
        lo0 = (w & (tiles_per_block - 1)) << log_T;
        base = ((uint64_t)hi << (log_L + A.log_S)) + lo0;
    } else {
        const uint32_t log_groups = A.log_L0 - log_T;     // tiles along digit 0
        a0 = (w & ((1u << log_groups) - 1)) << log_T;
        mid = w >> log_groups;
    }

    // ---- load tile into LDS (element (j,t) at j*T + t), optional coset pre-scale ----
    // Round 6: a lane has FOUR elements' loads in flight before it touches the first (the per-element loop waited out one HBM round
    // trip per element, and a 2^20 pass is exactly one resident set of workgroups that all sit in their load phase at once).
    for (uint32_t idx0 = tid; idx0 < LT; idx0 += 4 * nthr) {
        Fe raw[4], raw2[4];
        uint32_t slot[4];
        uint64_t gs[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t idx = idx0 + u * nthr;
            if (idx >= LT) break;
            uint32_t j, t;
            uint64_t g;
            if (!A.is_last) {
                t = idx & (T - 1); j = idx >> log_T;
                g = base + ((uint64_t)j << A.log_S) + t;
            } else {
                j = idx & (L - 1); t = idx >> log_L;
                g = ((uint64_t)(a0 + t) << A.log_S0) + ((uint64_t)mid << log_L) + j;
            }
            const Fe* from = &src[g];
            if (MODE == NTT_GATHER && A.first) {
                const uint64_t vec = blockIdx.y >> A.post_lr2, c2 = blockIdx.y & ((1u << A.post_lr2) - 1);
                const uint64_t q = g >> A.post_lr1, r = g & (((uint64_t)1 << A.post_lr1) - 1);
                from = &A.gather_in[((((q * A.post_k + vec) << A.post_lr1) + r) << A.post_lr2) + c2];
            }
            raw[u] = *from;
            if (A.in2) raw2[u] = A.in2[g];
            slot[u] = (j << log_T) + t;
            gs[u] = g;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (idx0 + u * nthr >= LT) break;
            const uint64_t g = gs[u];
            El v = F::unpack(raw[u]);
            if (A.in2) {        // pass 0 of a product transform: v = in[g] * in2[g] (Montgomery product of the reference format)
                v = F::mul(v, F::unpack(raw2[u]));
                if (F::kInternalDomain && A.fold_in != 2) v = F::mul(v, F::unpack(A.k271));
            } else if (A.prescale) {   // only ever set for pass 0, where storage index == input index
                const uint32_t e = A.prescale == 2 ? ((uint32_t)g << A.pre_shift) + A.pre_row0 + (blockIdx.y & A.pre_row_mask) : (uint32_t)g;
                El f = F::mul(F::unpack(A.cs_hi[e >> A.hc]), F::unpack(A.cs_lo[e & ((1u << A.hc) - 1)]));
                v = F::mul(v, f);          // (radix-2^29: cs_lo carries the extra 2^5 => also converts the domain)
            } else if (F::kInternalDomain && A.first && !A.fold_in) {
                v = F::to_internal(raw[u]);
            }
            tile.put(slot[u], v);
        }
    }
    __syncthreads();

    // ---- length-L DIF transform of every column t (Gentleman-Sande; output bit-reversed) ----
    // Two stages per LDS round trip: a lane takes the four elements j0, j0+h/2, j0+h, j0+3h/2 of one column
    // through stage hs (span h) and stage hs-1 (span h/2) in registers -- half the LDS traffic and half the
    // barriers of a stage-by-stage loop, same products.
    int hs = (int)log_L - 1;
    for (; hs >= 1; hs -= 2) {
        const uint32_t qmask = (1u << (hs - 1)) - 1;              // bits below the two stage bits
        const uint32_t sh_hi = A.log_lmax - 1 - hs;               // table stride of stage hs (w_{2h})
        for (uint32_t idx = tid; idx < (LT >> 2); idx += nthr) {
            const uint32_t t = idx & (T - 1), g = idx >> log_T;
            const uint32_t jl = g & qmask;
            const uint32_t j0 = ((g >> (hs - 1)) << (hs + 1)) | jl;
            const uint32_t q = 1u << (hs - 1 + log_T);            // h/2 in LDS slots
            const uint32_t i0 = (j0 << log_T) + t;
            El x0 = tile.get(i0), x1 = tile.get(i0 + q), x2 = tile.get(i0 + 2 * q), x3 = tile.get(i0 + 3 * q);
            // stage hs: (j0, j0+h) with w_{2h}^{jl}, (j0+h/2, j0+3h/2) with w_{2h}^{jl+h/2}
            // (radix-2^29 field: tile values live in [0, 4p); sums are not corrected, only y0 -- and y1 of the last pair
            // of stages -- are folded back from [0, 16p); differences go into products uncorrected: field29.h add_nr)
            // w_{2h}^{jl}, w_{2h}^{jl + h/2}, w_h^{jl}: out of LDS (index j of w_L^j) or the global table (index j << tw_shift)
            const uint32_t ia = jl << (sh_hi - tw_shift), ib = (jl + (1u << (hs - 1))) << (sh_hi - tw_shift);
            const El wa = lds_tw ? ltw.get(ia) : SmallTw<F>::get(A, ia << tw_shift);
            const El wb = lds_tw ? ltw.get(ib) : SmallTw<F>::get(A, ib << tw_shift);
            El s0 = F::add_nr(x0, x2), d0 = F::mul(F::sub_weak4(x0, x2), wa);
            El s1 = F::add_nr(x1, x3), d1 = F::mul(F::sub_weak4(x1, x3), wb);
            // stage hs-1: (j0, j0+h/2) and (j0+h, j0+3h/2), both with w_{h}^{jl}
            El y0 = F::fold16(F::add_nr(s0, s1)), y2 = F::add_nr(d0, d1), y1, y3;
            if (hs > 1) {
                const uint32_t ic = jl << (sh_hi + 1 - tw_shift);
                const El w = lds_tw ? ltw.get(ic) : SmallTw<F>::get(A, ic << tw_shift);
                y1 = F::mul(F::sub_weak8(s0, s1), w);
                y3 = F::mul(F::sub_weak(d0, d1), w);
            } else {
                y1 = F::fold16(F::sub_weak8(s0, s1));
                y3 = F::sub_weak(d0, d1);
            }
            tile.put(i0, y0);
            tile.put(i0 + q, y1);
            tile.put(i0 + 2 * q, y2);
            tile.put(i0 + 3 * q, y3);
        }
        __syncthreads();
    }
    if (hs == 0) {                                                // odd log_L: one plain stage of span 1, no twiddle
        for (uint32_t idx = tid; idx < (LT >> 1); idx += nthr) {
            const uint32_t t = idx & (T - 1), bb = idx >> log_T;
            const uint32_t i0 = ((bb << 1) << log_T) + t, i1 = i0 + T;
            El u = tile.get(i0), v = tile.get(i1);
            tile.put(i0, F::fold8(F::add_nr(u, v)));
            tile.put(i1, F::fold8(F::sub_weak4(u, v)));
        }
        __syncthreads();
    }

    // ---- write out ----
    if (!A.is_last) {
        const uint32_t log_c = A.log_n - log_L - A.log_S;   // c = N / (L*S)
        for (uint32_t idx0 = tid; idx0 < LT; idx0 += 4 * nthr) {
            Fe twf[4];
            if (A.tw_full) {        // the four table entries' loads in flight before the first product (round 6)
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t idx = idx0 + u * nthr;
                    if (idx >= LT) break;
                    twf[u] = A.tw_full[((uint64_t)(idx >> log_T) << A.log_S) + lo0 + (idx & (T - 1))];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t idx = idx0 + u * nthr;
                if (idx >= LT) break;
                const uint32_t t = idx & (T - 1), kk = idx >> log_T;
                const uint32_t r = __brev(kk) >> (32 - log_L);
                El v = tile.get((r << log_T) + t);
                if (A.tw_full) {
                    v = F::mul(v, F::unpack(twf[u]));
                } else if (A.apply_twiddle) {
                    const uint32_t e = (kk * (lo0 + t)) << log_c;
                    El f = F::mul(F::unpack(A.tw_hi[e >> A.h]), F::unpack(A.tw_lo[e & ((1u << A.h) - 1)]));
                    v = F::mul(v, f);
                }
                dst[base + ((uint64_t)kk << A.log_S) + t] = F::pack(v);
            }
        }
    } else {
        // digit-reverse the middle digits k_1..k_{np-2} of this tile
        uint64_t revmid = 0;
        {
            uint32_t rem = mid, shift = 0;
            for (uint32_t d = 0; d + 1 < A.np; d++) shift += A.kdig[d];   // = log2(L_0..L_{np-2})
            for (int d = (int)A.np - 2; d >= 1; d--) {
                shift -= A.kdig[d];
                const uint32_t kd = rem & ((1u << A.kdig[d]) - 1);
                rem >>= A.kdig[d];
                revmid += (uint64_t)kd << shift;
            }
        }
        const uint32_t log_rest = A.log_n - log_L;   // multiplier of the last digit
        for (uint32_t idx = tid; idx < LT; idx += nthr) {
            const uint32_t t = idx & (T - 1), kk = idx >> log_T;
            const uint32_t r = (log_L == 0) ? 0 : (__brev(kk) >> (32 - log_L));
            El v = tile.get((r << log_T) + t);
            Fe o;
            const uint64_t at = ((uint64_t)kk << log_rest) + revmid + a0 + t;
            if (POST) {
                const uint32_t r = blockIdx.y & ((1u << A.post_lr1) - 1), vec = blockIdx.y >> A.post_lr1;
                const uint64_t e = (A.post_row0 + r) * at;
                const El f = F::mul(F::unpack(A.post_hi[e >> A.post_h]), F::unpack(A.post_lo[e & (((uint64_t)1 << A.post_h) - 1)]));
                v = A.out_plain ? F::fold4to2(v) : F::mul(v, F::unpack(A.out_scale));      // the value in the reference form
                const uint64_t q = at >> A.post_lr2, c2 = at & (((uint64_t)1 << A.post_lr2) - 1);
                A.post_out[((((q * A.post_k + vec) << A.post_lr1) + r) << A.post_lr2) + c2] = F::pack(F::canonical(F::mul(v, f)));
                continue;
            }
            if (F::kInternalDomain && A.out_plain) {
                v = F::canonical(F::fold4to2(v));   // already in the reference domain (folded upstream)
            } else if (F::kInternalDomain) {
                // out_scale = 1 or 1/n in the REFERENCE Montgomery form: as an internal-domain operand it is
                // (2^-5) or (2^-5 / n), so this one product also converts back; then canonicalise
                v = F::canonical(F::mul(v, F::unpack(A.out_scale)));
            } else {
                if (A.scale) v = F::mul(v, F::unpack(A.out_scale));
            }
            o = A.comb_e ? CombineEpilogue<F>::run(A, v, at) : F::pack(v);
            dst[at] = o;
        }
    }
}

// full inter-digit twiddle table of one pass: out[kk*S + i] = w_N^(c*kk*i) * factor   (internal domain)
template <class F>
__global__ __launch_bounds__(256) void ntt_build_twiddles(const Fe* __restrict__ tw_lo, const Fe* __restrict__ tw_hi, uint32_t h,
                                                            uint32_t log_c, uint32_t log_S, uint64_t count, Fe factor,
                                                            Fe* __restrict__ out) {
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count) return;
    const uint32_t kk = (uint32_t)(idx >> log_S), i = (uint32_t)(idx & (((uint64_t)1 << log_S) - 1));
    const uint32_t e = (kk * i) << log_c;
    typename F::El f = F::mul(F::unpack(tw_hi[e >> h]), F::unpack(tw_lo[e & ((1u << h) - 1)]));
    f = F::mul(f, F::unpack(factor));
    out[idx] = F::pack(F::canonical(f));
}

// ---------------------------------------------------------------------------
// host side: plans (twiddle tables) and pass scheduling
// ---------------------------------------------------------------------------
static const int LOG_LMAX = 10;

struct NttPlan {
    int bits = 0;
    int np = 1;
    int k[4] = {0, 0, 0, 0};
    int h = 0, hc = 0;
    bool field29 = false;                     // table format: internal domain of Field29 (entries x 2^5)
    DevBuf tw_small[2], tw_lo[2], tw_hi[2];   // [0] forward root, [1] inverse root
    DevBuf tw_small29[2];                     // tw_small as limbs (TwLimbs), radix-2^29 path
    DevBuf cs_lo, cs_hi;                      // coset w_{2n}^i (forward root), format of the NTT kernel
    DevBuf cs_lo_ref, cs_hi_ref;              // the same in the reference Montgomery form (calch.hip)
    DevBuf tw_lo_ref[2], tw_hi_ref[2];        // w_n^e two-level in the reference form (dist_scale: four-step inter-digit twiddle)
    DevBuf tw_full[2][4][3];                  // [dir][pass][fold_in]: full per-pass twiddles (radix-2^29, np >= 2); fold_in 2 = x 2^10
    // The full tables are generated lazily, by whichever caller needs one first.  Several lanes (host threads, streams)
    // share one plan, so a table is built under tw_mu on the context's utility queue and the builder WAITS for the
    // kernel before publishing the flag: a consumer on any stream then finds a finished table (no cross-stream event).
    std::mutex tw_mu;
    bool tw_ready[2][4][3] = {};
    DevBuf cs_lo_int;                         // coset factors' low table in the internal form (x 2^5 once): CALC_H epilogue
    Fe n_inv;                                 // reference Montgomery form of 1/n
};

// w_{2^28} = 5^((r-1)/2^28) (src/build_fft.js:29-47), plain form
static Fe root_2_28_mont() {
    Fe plain = {{0x9bd61b6e725b19f0ull, 0x402d111e41112ed4ull, 0x00e0a7eb8ef62abcull, 0x2a3c09f0a58a7e85ull}};
    return Fr::to_mont(plain);
}
static Fe root_of_unity(int s) {   // w_{2^s}, Montgomery
    Fe w = root_2_28_mont();
    for (int i = 28; i > s; i--) w = Fr::sqr(w);
    return w;
}
static void powers(const Fe& base, size_t count, std::vector<Fe>& out) {
    out.resize(count);
    Fe acc = Fr::one();
    for (size_t i = 0; i < count; i++) { out[i] = acc; acc = Fr::mul(acc, base); }
}
// x -> x * 2^5 : reference Montgomery form (R = 2^256) -> internal form of Field29 (R' = 2^261)
static void scale32(std::vector<Fe>& v) {
    const Fe m32 = Fr::to_mont(Fe{{32, 0, 0, 0}});
    for (auto& x : v) x = Fr::mul(x, m32);
}
static int upload(DevBuf& b, const std::vector<Fe>& v, hipStream_t s) {
    WS_HIP_CHECK(b.alloc(v.size() * sizeof(Fe)));
    WS_HIP_CHECK(hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(Fe), hipMemcpyHostToDevice, s));
    WS_HIP_CHECK(hipStreamSynchronize(s));   // v is a host temporary
    return WS_OK;
}

static int build_plan(int bits, NttPlan& P, hipStream_t s) {
    P.bits = bits;
    P.np = bits <= LOG_LMAX ? 1 : (bits <= 16 ? 2 : (bits <= 24 ? 3 : 4));
    // (2^20 in TWO passes of 2^10 was measured in round 5: 0.164 ms per transform against 0.157 in three, whole proofs 9.98 against
    //  9.70 ms -- profiles/r05_schedule_experiments.txt: ten butterfly stages per tile cost a pass more LDS round trips than they save)
    int basek = bits / P.np, rem = bits % P.np;
    for (int d = 0; d < P.np; d++) P.k[d] = basek + (d < rem ? 1 : 0);
    P.h = (bits + 1) / 2;
    P.hc = (bits + 1) / 2;
    P.field29 = true;       // (the saturated 4 x 64 transform kernels were an A/B path of rounds 1-3; the default build no longer carries them)
    const bool f29 = P.field29;
    std::vector<Fe> tmp;
    for (int dir = 0; dir < 2; dir++) {
        Fe wl = root_of_unity(LOG_LMAX), wn = root_of_unity(bits);
        if (dir) { wl = Fr::inv(wl); wn = Fr::inv(wn); }
        powers(wl, (size_t)1 << (LOG_LMAX - 1), tmp);
        if (f29) scale32(tmp);
        int rc = upload(P.tw_small[dir], tmp, s); if (rc) return rc;
        if (f29) {
            std::vector<TwLimbs> limbs(tmp.size());
            for (size_t i = 0; i < tmp.size(); i++) {
                const F29 u = Fr29::unpack(tmp[i]);
                for (int k = 0; k < 12; k++) limbs[i].v[k] = k < 9 ? u.v[k] : 0;
            }
            WS_HIP_CHECK(P.tw_small29[dir].alloc(limbs.size() * sizeof(TwLimbs)));
            WS_HIP_CHECK(hipMemcpyAsync(P.tw_small29[dir].p, limbs.data(), limbs.size() * sizeof(TwLimbs), hipMemcpyHostToDevice, s));
            WS_HIP_CHECK(hipStreamSynchronize(s));
        }
        powers(wn, (size_t)1 << P.h, tmp);
        rc = upload(P.tw_lo_ref[dir], tmp, s); if (rc) return rc;
        if (f29) scale32(tmp);
        rc = upload(P.tw_lo[dir], tmp, s); if (rc) return rc;
        powers(Fr::pow_u64(wn, (uint64_t)1 << P.h), (size_t)1 << (bits - P.h), tmp);
        rc = upload(P.tw_hi_ref[dir], tmp, s); if (rc) return rc;
        if (f29) scale32(tmp);
        rc = upload(P.tw_hi[dir], tmp, s); if (rc) return rc;
    }
    if (bits < 28) {
        Fe g = root_of_unity(bits + 1);
        powers(g, (size_t)1 << P.hc, tmp);
        int rc = upload(P.cs_lo_ref, tmp, s); if (rc) return rc;
        if (f29) { scale32(tmp); rc = upload(P.cs_lo_int, tmp, s); if (rc) return rc; scale32(tmp); }   // cs_lo carries the 2^5 of the entry AND the 2^5 that converts the loaded value
        rc = upload(P.cs_lo, tmp, s); if (rc) return rc;
        powers(Fr::pow_u64(g, (uint64_t)1 << P.hc), (size_t)1 << (bits - P.hc), tmp);
        rc = upload(P.cs_hi_ref, tmp, s); if (rc) return rc;
        if (f29) scale32(tmp);
        rc = upload(P.cs_hi, tmp, s); if (rc) return rc;
    }
    // n^-1 = (2^-1)^bits  (INV2 table of build_fft.js:59-72)
    Fe two = Fr::add(Fr::one(), Fr::one());
    Fe half = Fr::inv(two), ninv = Fr::one();
    for (int i = 0; i < bits; i++) ninv = Fr::mul(ninv, half);
    P.n_inv = ninv;
    return WS_OK;
}

static int get_plan(Context* C, int bits, std::shared_ptr<NttPlan>& P, hipStream_t s) {
    std::lock_guard<std::mutex> lk(C->mu);
    auto it = C->ntt_plans.find(bits);
    if (it == C->ntt_plans.end()) {
        P = std::make_shared<NttPlan>();
        int rc = build_plan(bits, *P, s);
        if (rc) return rc;
        C->ntt_plans[bits] = P;
    } else {
        P = it->second;
    }
    return WS_OK;
}

// two-level table of the coset factors w_{2n}^i, i < n (n = 2^bits): value = hi[i >> hc] * lo[i & mask]
int ntt_coset_tables(int bits, const Fe** lo, const Fe** hi, int* hc, Fe* n_inv, hipStream_t s) {
    Context* C = ctx();
    if (!C) return WS_ERR_NOINIT;
    if (bits < 1 || bits >= 28) return WS_ERR_SIZE;
    std::shared_ptr<NttPlan> P;
    int rc = get_plan(C, bits, P, s);
    if (rc) return rc;
    *lo = P->cs_lo_ref.as<Fe>(); *hi = P->cs_hi_ref.as<Fe>(); *hc = P->hc; *n_inv = P->n_inv;
    return WS_OK;
}

// the coset factors w_2n^i of a length-n transform in the format ntt_pass_kernel's pre-scale reads (NttRowCoset: lo, hi, hc)
int ntt_coset_tables_kernel_format(int bits, const Fe** lo, const Fe** hi, uint32_t* hc, hipStream_t s) {
    Context* C = ctx();
    if (!C) return WS_ERR_NOINIT;
    if (bits < 1 || bits >= 28) return WS_ERR_SIZE;
    std::shared_ptr<NttPlan> P;
    int rc = get_plan(C, bits, P, s);
    if (rc) return rc;
    *lo = P->cs_lo.as<Fe>(); *hi = P->cs_hi.as<Fe>(); *hc = (uint32_t)P->hc;
    return WS_OK;
}

// w_n^(+-e), e < n = 2^bits, two-level in the reference Montgomery form: value = hi[e >> h] * lo[e & mask]
int ntt_twiddle_tables(int bits, int inverse, const Fe** lo, const Fe** hi, int* h, hipStream_t s, bool internal) {
    Context* C = ctx();
    if (!C) return WS_ERR_NOINIT;
    if (bits < 1 || bits > 28) return WS_ERR_SIZE;
    std::shared_ptr<NttPlan> P;
    int rc = get_plan(C, bits, P, s);
    if (rc) return rc;
    if (internal) {
        if (!P->field29) { set_last_error("ntt: the internal twiddle tables exist on the radix-2^29 field only"); return WS_ERR_ARG; }
        *lo = P->tw_lo[inverse ? 1 : 0].as<Fe>(); *hi = P->tw_hi[inverse ? 1 : 0].as<Fe>(); *h = P->h;
        return WS_OK;
    }
    *lo = P->tw_lo_ref[inverse ? 1 : 0].as<Fe>(); *hi = P->tw_hi_ref[inverse ? 1 : 0].as<Fe>(); *h = P->h;
    return WS_OK;
}

int ntt_dev(Lane& L, Fe* d_data, uint64_t n, int odd, int inverse, hipStream_t s, uint64_t count) {
    return ntt_run(L, d_data, nullptr, d_data, nullptr, n, odd, inverse, s, count);
}

// src (x in2) -> dst; combine_e != nullptr: the last pass stores CALC_H's h instead of the transform (inverse only)
int ntt_run(Lane& L, const Fe* d_src, const Fe* d_in2, Fe* d_data, const Fe* combine_e, uint64_t n, int odd, int inverse,
            hipStream_t s, uint64_t count, const NttRowCoset* rc_pre, const NttRowPost* rc_post, const NttRowGather* rc_gather) {
    Context* C = ctx();
    if (!C) return WS_ERR_NOINIT;
    if (!d_data || !d_src) return WS_ERR_ARG;
    if ((d_in2 && odd) || (combine_e && (!inverse || count != 1)) || (rc_pre && (odd || d_in2)) || (rc_post && combine_e) ||
        (rc_gather && (rc_post || rc_pre || odd || d_in2))) return WS_ERR_ARG;
    if (!s) s = L.stream;
    // src/build_fft.js:92-157: n must be a power of two <= 2^28 (the reference traps otherwise)
    if (n == 0 || (n & (n - 1)) || n > ((uint64_t)1 << 28)) return WS_ERR_SIZE;
    int bits = 0;
    while (((uint64_t)1 << bits) < n) bits++;
    if (odd && bits >= 28) return WS_ERR_SIZE;   // needs w_{2^29}, which does not exist
    if (count == 0) return WS_OK;
    if (count > 65535 || n * count > ((uint64_t)1 << 30)) return WS_ERR_SIZE;
    if (bits == 0) {
        // fft(n=1) is the identity; ifft(n=1) never terminates in the reference
        // (src/build_fft.js:575-583) -> reported as a size error here
        return inverse ? WS_ERR_SIZE : WS_OK;
    }
    ScratchGuard scratch_turn(L.ntt_chain, s);   // the ping-pong buffer is shared by every transform on this lane
    std::shared_ptr<NttPlan> P;
    {
        int rc = get_plan(C, bits, P, s);
        if (rc) return rc;
        if (P->np > 1) WS_HIP_CHECK(L.ntt_scratch.reserve(n * count * sizeof(Fe)));
    }
    Fe* scratch = L.ntt_scratch.as<Fe>();
    // inverse (reference semantics): raw forward transform, then y[i] = raw[(n-i) mod n]/n
    //   == DFT with the inverse root, scaled by 1/n; an `odd` input keeps the FORWARD coset
    //   factors w_2n^i because rawfft(odd) multiplies x[i] by w_2n^i before the index flip.
    const int dir = inverse ? 1 : 0;
    int sum_after = bits;
    for (int p = 0; p < P->np; p++) {
        PassArgs A;
        const bool last = (p == P->np - 1);
        sum_after -= P->k[p];
        A.in = (p == 0) ? d_src : scratch;
        A.out = last ? d_data : scratch;
        A.in2 = (p == 0) ? d_in2 : nullptr;
        A.comb_e = last ? combine_e : nullptr;
        A.comb_lo = P->field29 ? P->cs_lo_int.as<Fe>() : P->cs_lo_ref.as<Fe>();
        A.comb_hi = P->field29 ? P->cs_hi.as<Fe>() : P->cs_hi_ref.as<Fe>();
        A.comb_hc = (uint32_t)P->hc;
        // 2^271 mod r (raw): as an internal-domain operand it multiplies by 2^10
        A.k271 = Fr::to_mont(Fe{{(uint64_t)1 << 15, 0, 0, 0}});
        // epilogue constant: internal path raw 16 (mul: x 16 x 2^-261 = halve and leave the Montgomery form); 4x64 path 1/2
        A.comb_c = P->field29 ? Fe{{16, 0, 0, 0}} : Fr::inv(Fr::add(Fr::one(), Fr::one()));
        A.log_n = bits; A.log_L = P->k[p]; A.log_S = sum_after;
        A.is_last = last ? 1 : 0;
        A.log_L0 = P->k[0]; A.log_S0 = bits - P->k[0];
        A.np = P->np;
        for (int d = 0; d < 4; d++) A.kdig[d] = P->k[d];
        A.tw_small = P->tw_small[dir].as<Fe>(); A.log_lmax = LOG_LMAX;
        A.tw_small29 = P->tw_small29[dir].as<TwLimbs>();
        A.tw_lo = P->tw_lo[dir].as<Fe>(); A.tw_hi = P->tw_hi[dir].as<Fe>(); A.h = P->h;
        A.apply_twiddle = last ? 0 : 1;
        A.cs_lo = P->cs_lo.as<Fe>(); A.cs_hi = P->cs_hi.as<Fe>(); A.hc = P->hc;
        A.prescale = (odd && p == 0) ? 1 : 0;
        A.pre_shift = A.pre_row0 = A.pre_row_mask = 0;
        if (rc_pre && p == 0) {
            A.cs_lo = rc_pre->lo; A.cs_hi = rc_pre->hi; A.hc = rc_pre->hc;
            A.prescale = 2; A.pre_shift = rc_pre->shift; A.pre_row0 = rc_pre->row0; A.pre_row_mask = rc_pre->row_mask;
        }
        A.post_out = nullptr; A.post_lo = A.post_hi = nullptr; A.post_h = A.post_lr1 = A.post_lr2 = A.post_k = 0; A.post_row0 = 0;
        if (rc_post && last) {
            A.post_out = rc_post->out; A.post_lo = rc_post->lo; A.post_hi = rc_post->hi; A.post_h = rc_post->h;
            A.post_lr1 = rc_post->lr1; A.post_lr2 = rc_post->lr2; A.post_k = rc_post->k; A.post_row0 = rc_post->row0;
        }
        A.gather_in = nullptr;
        if (rc_gather && p == 0) { A.gather_in = rc_gather->in; A.post_lr1 = rc_gather->lr1; A.post_lr2 = rc_gather->lr2; A.post_k = rc_gather->k; }
        A.scale = (inverse && last) ? 1 : 0;
        A.first = (p == 0) ? 1 : 0;
        A.out_scale = A.scale ? P->n_inv : Fr::one();
        A.tw_full = nullptr; A.fold_in = 0; A.out_plain = 0;
        if (P->field29 && P->np >= 2) {
            const int fold_in = (p == 0 && !odd && !rc_pre) ? (d_in2 ? 2 : 1) : 0;   // odd: the coset product already converts; 2: product on load
            const bool fold_out = (p == P->np - 2);
            A.fold_in = (uint32_t)fold_in;
            A.out_plain = last ? 1 : 0;
            if (!last) {
                DevBuf& tb = P->tw_full[dir][p][fold_in];
                const uint64_t count = (uint64_t)1 << (A.log_L + A.log_S);
                {
                    std::lock_guard<std::mutex> tw_lk(P->tw_mu);
                    if (!P->tw_ready[dir][p][fold_in]) {
                        // factor (plain field element): 2^5 (in; 2^10 after a product on load), 2^-5 (out), times 1/n on the inverse's out pass
                        Fe f = Fr::one();                                           // Montgomery-R form of 1
                        const Fe m32 = Fr::to_mont(Fe{{32, 0, 0, 0}});
                        for (int k = 0; k < fold_in; k++) f = Fr::mul(f, m32);
                        if (fold_out) { f = Fr::mul(f, Fr::inv(m32)); if (inverse) f = Fr::mul(f, P->n_inv); }
                        f = Fr::mul(f, m32);                                        // -> internal form (x 2^5)
                        WS_HIP_CHECK(tb.alloc(count * sizeof(Fe)));
                        hipLaunchKernelGGL(ntt_build_twiddles<Fr29>, dim3(ceil_div_u64(count, 256)), dim3(256), 0, C->stream,
                                           P->tw_lo[dir].as<Fe>(), P->tw_hi[dir].as<Fe>(), (uint32_t)P->h,
                                           (uint32_t)(bits - A.log_L - A.log_S), A.log_S, count, f, tb.as<Fe>());
                        WS_HIP_CHECK(hipGetLastError());
                        WS_HIP_CHECK(hipStreamSynchronize(C->stream));
                        P->tw_ready[dir][p][fold_in] = true;
                    }
                }
                A.tw_full = tb.as<Fe>();
            }
        }
        // tile: 2^tile_log elements of LDS
        static const int tile_log = 10;   // 1024-element tiles (36 KiB): four workgroups per CU hide each other's load/store phases (2^22 pair 1.34 -> 1.30 ms vs 2048)
        int log_T = tile_log - (int)A.log_L;
        if (log_T > 5) log_T = 5;
        if (log_T < 0) log_T = 0;
        if (P->np == 1) {
            log_T = 0; A.log_S0 = 0; A.log_L0 = 0;
        } else if (last) {
            if (log_T > (int)A.log_L0) log_T = A.log_L0;
        } else {
            if (log_T > (int)A.log_S) log_T = A.log_S;
        }
        A.log_T = log_T;
        const uint32_t grid = (uint32_t)(n >> (A.log_L + log_T));
        const size_t elems = (size_t)1 << (A.log_L + log_T);
        C->timer.begin(last ? "ntt_pass_last" : "ntt_pass", s);
        if (P->field29) {
            // round 6: the pass's L/2 small twiddles behind the tile (<= 128 entries = 4.5 KiB: four workgroups per CU still fit)
            A.lds_tw = (A.log_L >= 2 && A.log_L <= 8 && elems % 4 == 0 && !tuning_get("NTT_GLOBAL_TW", 0)) ? 1 : 0;
            const size_t smem = elems * LdsTile<Fr29>::kBytes + (A.lds_tw ? ((size_t)1 << (A.log_L - 1)) * LdsTw<Fr29>::kBytes : 0);
            if (!C->ntt_attr_set) {   // (per device; two lanes may both set it once: same value)  2048-element tiles would need 72 KiB of dynamic LDS (> the 64 KiB default cap)
                WS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ntt_pass_kernel<Fr29, NTT_PLAIN>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 2048 * 36));
                WS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ntt_pass_kernel<Fr29, NTT_POST>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 2048 * 36));
                WS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ntt_pass_kernel<Fr29, NTT_GATHER>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 2048 * 36));
                C->ntt_attr_set = true;
            }
            const dim3 g3(grid, (uint32_t)count), b3(512 >> (11 - tile_log));
            if (A.post_out) hipLaunchKernelGGL((ntt_pass_kernel<Fr29, NTT_POST>), g3, b3, smem, s, A);
            else if (A.gather_in) hipLaunchKernelGGL((ntt_pass_kernel<Fr29, NTT_GATHER>), g3, b3, smem, s, A);
            else hipLaunchKernelGGL((ntt_pass_kernel<Fr29, NTT_PLAIN>), g3, b3, smem, s, A);
        }
        C->timer.end(s);
        WS_HIP_CHECK(hipGetLastError());
    }
    return WS_OK;
}

WS_DEFINE_WARM(ntt)

}  // namespace wsnark
