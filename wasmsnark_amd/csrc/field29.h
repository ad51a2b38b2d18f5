// field29.h -- carry-free radix-2^29 Montgomery arithmetic for the heavy device kernels.
//
// Why: on gfx950 a saturated 8x32-bit multiplier spends more instructions on carries, 64-bit adds
// and register moves than on products (584 VALU instructions per product, 96 G modmul/s measured),
// and every carry chain through VCC pays wait states.  With 9 limbs of 29 bits every column of the
// schoolbook product (<= 9 a*b + 9 m*p terms of < 2^58) fits one 64-bit accumulator, so a whole
// Montgomery product is a chain of v_mad_u64_u32 (162) plus 9 v_mul_lo, 16 shifts and 17 masks:
// no carries, no moves -- 205 instructions, 178-183 G modmul/s measured (wsnark_peak_probe), in the form the device runs:
// mad_chain.h, generated from the column structure below (the compiler's own arrangement of these loops is 221 instructions,
// 172 G/s: one 64-bit addition per column more).
//
// Semantics.  Same field, same results: this is an internal representation of the heavy kernels
// (MSM curve arithmetic, NTT butterflies).  It replaces the same reference functions as field.h
// (/root/reference src/build_f1m.js:67-113, 235-436).
//   * Montgomery radix here is R' = 2^261 (9 x 29 bits).  External data stays in the reference's
//     format (R = 2^256, canonical, 32-byte LE): to_internal() / from_internal() convert with one
//     product each (x * 2^5 and x * 2^-5), and from_internal() canonicalises to [0,p).
//   * Inside, values are kept in [0, 2p) ("lazy mod 2p") with limbs v[0..7] < 2^29: add/sub
//     correct by 2p once, so every operand bound is uniform and the generic curve formulas of
//     curve.h work unchanged.  A value is zero iff it is 0 or p.
//   * Packed form (global memory / LDS between kernels): 32 bytes, any value < 2^256.
#pragma once
#include "field.h"

namespace wsnark {

struct F29 {
    uint32_t v[9];
};

struct Fq29Params : FqParams {
    // 2^261 mod p (one in the internal domain) and 2^266 mod p (the to_internal multiplier)
    static constexpr uint64_t ONE0 = 0x4e8384eb157ccc21ull, ONE1 = 0xfb90a6020ce148c3ull, ONE2 = 0x5301fa84819caa36ull, ONE3 = 0x0dc83629563d4475ull;
    static constexpr uint64_t CIN0 = 0xb34bb09513349ca1ull, CIN1 = 0x1e880124f028f972ull, CIN2 = 0xe56cdd25a6092b95ull, CIN3 = 0x05800320dce9ed32ull;
};
struct Fr29Params : FrParams {
    static constexpr uint64_t ONE0 = 0x2fd4e1568fffff57ull, ONE1 = 0x75bba827a494b01aull, ONE2 = 0x5301fa84819caa80ull, ONE3 = 0x0dc83629563d4475ull;
    static constexpr uint64_t CIN0 = 0x97aa889e8fffead7ull, CIN1 = 0x4da1da684b110e2aull, CIN2 = 0xe56cdd25a60934c8ull, CIN3 = 0x05800320dce9ed32ull;
};

#if defined(__HIP_DEVICE_COMPILE__)
#define WS_NOINLINE_DEV __device__ __attribute__((noinline))
#else
#define WS_NOINLINE_DEV inline
#endif
// operands travel as 18 / 9 scalar arguments: the AMDGPU calling convention keeps scalars in VGPRs,
// whereas a 36-byte aggregate is partly passed through scratch memory
#define WS_L9(p) uint32_t p##0, uint32_t p##1, uint32_t p##2, uint32_t p##3, uint32_t p##4, uint32_t p##5, uint32_t p##6, uint32_t p##7, uint32_t p##8
#define WS_A9(x) (x).v[0], (x).v[1], (x).v[2], (x).v[3], (x).v[4], (x).v[5], (x).v[6], (x).v[7], (x).v[8]
template <class P> WS_NOINLINE_DEV F29 mont_mul29(WS_L9(a), WS_L9(b));
template <class P> WS_NOINLINE_DEV F29 mont_sqr29(WS_L9(a));
template <class P> WS_NOINLINE_DEV F29 mont_mul2add29(WS_L9(a), WS_L9(b), WS_L9(c), WS_L9(d));
template <class P> WS_HD F29 mont_mul29_body(const F29& a, const F29& b);
template <class P> WS_HD F29 mont_sqr29_body(const F29& a);
// On the device every product below runs as mad_chain.h's form of it (generated: the same columns with their multiply-adds as
// explicit v_mad_u64_u32 chains); the C bodies in this file are the host's, the emulator's, and the specification of the generated ones.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(WSNARK_EMUL) && !defined(WS_NO_MAD_CHAIN)
#define WS_USE_MAD_CHAIN 1
template <class P> __device__ __forceinline__ F29 mont_mul29_chain(const F29& a, const F29& b);
template <class P> __device__ __forceinline__ F29 mont_sqr29_chain(const F29& a);
template <class P> __device__ __forceinline__ F29 mont_mul2add29_chain(const F29& a, const F29& b, const F29& c, const F29& d);
template <class P> __device__ __forceinline__ F29 mont_mul4add29_chain(const F29& a, const F29& b, const F29& c, const F29& d, const F29& e, const F29& f,
                                                                       const F29& g, const F29& h);
#endif

#define WS_M29 0x1FFFFFFFu
#ifndef WS_F29_MULSUB_INLINE
#define WS_F29_MULSUB_INLINE 0
#endif

// limb i (29 bits; limb 8 takes the rest) of the 256-bit integer (w3:w2:w1:w0)
constexpr uint32_t ws_limb29(uint64_t w0, uint64_t w1, uint64_t w2, uint64_t w3, int i) {
    const int o = 29 * i, k = o >> 6, s = o & 63;
    const uint64_t lo = k == 0 ? w0 : k == 1 ? w1 : k == 2 ? w2 : w3;
    const uint64_t hi = k == 0 ? w1 : k == 1 ? w2 : k == 2 ? w3 : 0;
    const uint64_t x = (lo >> s) | (s ? (hi << (64 - s)) : 0);
    return i == 8 ? (uint32_t)x : (uint32_t)(x & WS_M29);
}

template <class P>
struct Field29 {
    typedef F29 El;
    typedef Fe Packed;
    static constexpr bool kInternalDomain = true;
    static constexpr bool kHasMul2Add = true;

    // ---- constants as compile-time limbs ----
    WS_HD static constexpr uint32_t p_limb(int i) { return ws_limb29(P::P0, P::P1, P::P2, P::P3, i); }
    WS_HD static constexpr uint32_t p2_limb(int i) {   // 2p
        return ws_limb29(P::P0 << 1, (P::P1 << 1) | (P::P0 >> 63), (P::P2 << 1) | (P::P1 >> 63), (P::P3 << 1) | (P::P2 >> 63), i);
    }
    static constexpr uint32_t NP29 = (uint32_t)(P::NP & WS_M29);
    // limb i of k*p for a small k (the top limb takes the rest: 10p < 2^257 still fits)
    WS_HD static constexpr uint32_t kp_limb(uint32_t k, int i) {
        uint64_t carry = 0, t = 0;
        for (int j = 0; j <= i; j++) {
            t = (uint64_t)p_limb(j) * k + carry;
            carry = t >> 29;
            if (j < 8) t &= WS_M29;
        }
        return (uint32_t)t;
    }

    WS_HD static F29 from_words(uint64_t w0, uint64_t w1, uint64_t w2, uint64_t w3) {
        F29 r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.v[i] = ws_limb29(w0, w1, w2, w3, i);
        return r;
    }
    WS_HD static F29 zero() { return F29{{0, 0, 0, 0, 0, 0, 0, 0, 0}}; }
    WS_HD static F29 one() { return from_words(P::ONE0, P::ONE1, P::ONE2, P::ONE3); }   // 1 in the internal domain

    // ---- pack / unpack (32-byte <-> 9 x 29) ----
    WS_HD static F29 unpack(const Fe& x) { return from_words(x.l[0], x.l[1], x.l[2], x.l[3]); }
    WS_HD static Fe pack(const F29& a) {   // value must be < 2^256 (v[8] < 2^24)
        uint32_t w[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int o = 32 * j, i0 = o / 29, s = o - 29 * i0;   // word j starts inside limb i0 at bit s
            uint32_t x = a.v[i0] >> s;
            x |= a.v[i0 + 1] << (29 - s);
            w[j] = x;
        }
        Fe r;
#pragma unroll
        for (int k = 0; k < 4; k++) r.l[k] = (uint64_t)w[2 * k] | ((uint64_t)w[2 * k + 1] << 32);
        return r;
    }

    // ---- predicates ----
    WS_HD static bool is_zero(const F29& a) {   // value in [0,2p): zero iff 0 or p
        uint32_t o = 0, d = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) { o |= a.v[i]; d |= a.v[i] ^ p_limb(i); }
        return o == 0 || d == 0;
    }
    WS_HD static bool eq(const F29& a, const F29& b) { return is_zero(sub(a, b)); }

    // ---- add / sub / neg, all results in [0, 2p) with tight limbs ----
    // r = s - 2p if s >= 2p else s, for s < 4p given with tight limbs
    WS_HD static F29 cond_sub_2p(const F29& s) {
        F29 d;
        int32_t bw = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int32_t t = (int32_t)s.v[i] - (int32_t)p2_limb(i) + bw;
            d.v[i] = (uint32_t)t & WS_M29;
            bw = t >> 29;
        }
        const int32_t t8 = (int32_t)s.v[8] - (int32_t)p2_limb(8) + bw;
        d.v[8] = (uint32_t)t8;
        const bool neg = t8 < 0;
        F29 r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.v[i] = neg ? s.v[i] : d.v[i];
        return r;
    }
    WS_HD static F29 add(const F29& a, const F29& b) {
        F29 s;
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t t = a.v[i] + b.v[i] + c;
            s.v[i] = t & WS_M29;
            c = t >> 29;
        }
        s.v[8] = a.v[8] + b.v[8] + c;
        return cond_sub_2p(s);
    }
    WS_HD static F29 dbl(const F29& a) { return add(a, a); }
    // a + b WITHOUT carry propagation or reduction: limbs < 2^30, value < 4p.  Only valid as a direct
    // operand of mul/sqr (column sums stay < 9*2^60 + 9*2^58 < 2^64; output < p(1 + 16p/2^261) < 2p).
    WS_HD static F29 add_lazy(const F29& a, const F29& b) {
        F29 s;
#pragma unroll
        for (int i = 0; i < 9; i++) s.v[i] = a.v[i] + b.v[i];
        return s;
    }
    WS_HD static F29 sub(const F29& a, const F29& b) {
        // d = a - b + 2p in (0, 4p), then one conditional subtraction of 2p
        F29 d;
        int32_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int32_t t = (int32_t)a.v[i] - (int32_t)b.v[i] + (int32_t)p2_limb(i) + c;
            d.v[i] = (uint32_t)t & WS_M29;
            c = t >> 29;
        }
        d.v[8] = (uint32_t)((int32_t)a.v[8] - (int32_t)b.v[8] + (int32_t)p2_limb(8) + c);
        return cond_sub_2p(d);
    }
    // a - b + 2p in (0, 4p) WITHOUT the final correction: tight limbs.  Valid only as an operand of
    // mul / sqr / is_zero_weak (16 p^2 / 2^261 < 0.1 p keeps products below 2p).
    WS_HD static F29 sub_weak(const F29& a, const F29& b) {
        F29 d;
        int32_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int32_t t = (int32_t)a.v[i] - (int32_t)b.v[i] + (int32_t)p2_limb(i) + c;
            d.v[i] = (uint32_t)t & WS_M29;
            c = t >> 29;
        }
        d.v[8] = (uint32_t)((int32_t)a.v[8] - (int32_t)b.v[8] + (int32_t)p2_limb(8) + c);
        return d;
    }
    // a - b + 4p for a, b < 4p (uncorrected differences themselves): in (0, 8p), tight limbs.  Operand of a product only.
    WS_HD static F29 sub_weak4(const F29& a, const F29& b) {
        F29 d;
        int32_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int32_t t = (int32_t)a.v[i] - (int32_t)b.v[i] + (int32_t)kp_limb(4, i) + c;
            d.v[i] = (uint32_t)t & WS_M29;
            c = t >> 29;
        }
        d.v[8] = (uint32_t)((int32_t)a.v[8] - (int32_t)b.v[8] + (int32_t)kp_limb(4, 8) + c);
        return d;
    }
    // 4p - c for c in [0, 4p): in (0, 4p]; only ever an operand of a product
    WS_HD static F29 neg_weak4(const F29& c) {
        F29 n;
        int32_t cy = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int32_t t = (int32_t)kp_limb(4, i) - (int32_t)c.v[i] + cy;
            n.v[i] = (uint32_t)t & WS_M29;
            cy = t >> 29;
        }
        n.v[8] = (uint32_t)((int32_t)kp_limb(4, 8) - (int32_t)c.v[8] + cy);
        return n;
    }
    // zero test for a sub_weak result (value in (0, 4p)): zero iff p, 2p or 3p
    WS_HD static bool is_zero_weak(const F29& a) {
        // cheap filter on the lowest limb first (a multiple of p matches one of three constants)
        const uint32_t l0 = a.v[0];
        if (l0 != p_limb(0) && l0 != p2_limb(0) && l0 != ((p_limb(0) + p2_limb(0)) & WS_M29)) return false;
        F29 t = cond_sub_2p(a);        // now in [0, 2p)
        return is_zero(t);
    }
    // ---- "wide" values: in (0, 8p) or (0, 10p), tight limbs (the top one holds up to 25 bits).  They save the correction
    // passes of the strict forms and are valid ONLY as operands of mul / sqr / mulsub2's first pair / sub_wide / is_zero_wide,
    // and as the stored x of a lazily accumulated point (Curve::madd_wide); narrow() brings one back to [0, 2p).
    // (a * b + m * p) / 2^261 with a, b < 10p is < (100 * 2^-7.4 + 1) p < 1.6p: products of wide operands stay below 2p. ----
    // rr - ppp - 2q + 6p for strict rr, ppp, q: the X3 of an addition in ONE carry pass; in (0, 8p)
    WS_HD static F29 x3_wide(const F29& rr, const F29& ppp, const F29& q) {
        F29 d;
        int32_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int32_t t = (int32_t)rr.v[i] - (int32_t)ppp.v[i] - 2 * (int32_t)q.v[i] + (int32_t)kp_limb(6, i) + c;   // > -2^31
            d.v[i] = (uint32_t)t & WS_M29;
            c = t >> 29;
        }
        d.v[8] = (uint32_t)((int32_t)rr.v[8] - (int32_t)ppp.v[8] - 2 * (int32_t)q.v[8] + (int32_t)kp_limb(6, 8) + c);
        return d;
    }
    // a - b + 8p for strict a and wide b (< 8p): in (0, 10p)
    WS_HD static F29 sub_wide(const F29& a, const F29& b) {
        F29 d;
        int32_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int32_t t = (int32_t)a.v[i] - (int32_t)b.v[i] + (int32_t)kp_limb(8, i) + c;
            d.v[i] = (uint32_t)t & WS_M29;
            c = t >> 29;
        }
        d.v[8] = (uint32_t)((int32_t)a.v[8] - (int32_t)b.v[8] + (int32_t)kp_limb(8, 8) + c);
        return d;
    }
    // zero test for a sub_wide result (value in (0, 10p)): zero iff k*p, k = 1..9
    WS_HD static bool is_zero_wide(const F29& a) {
        const uint32_t l0 = a.v[0];
        bool hit = false;
#pragma unroll
        for (uint32_t k = 1; k <= 9; k++) hit = hit || l0 == kp_limb(k, 0);
        if (!hit) return false;                 // (the low limbs of p, 2p, ..., 9p differ: p is odd)
#pragma unroll
        for (uint32_t k = 1; k <= 9; k++) {
            uint32_t d = 0;
#pragma unroll
            for (int i = 0; i < 9; i++) d |= a.v[i] ^ kp_limb(k, i);
            if (d == 0) return true;
        }
        return false;
    }
    // Cheap NECESSARY condition for "a is a multiple k p, 1 <= k <= kmax" (a difference that vanishes mod p): the lowest limb of k p is
    // k p_0 mod 2^29 and p_0 is odd, so l_0 p_0^-1 mod 2^29 must be k (or 0).  Three instructions where the nine-way comparison of
    // is_zero_wide compiles to a branch tree that every wavefront walks; a hit (2^-26 by chance) sends the lane to the exact test.
    WS_HD static constexpr uint32_t p0_inv29() {
        uint32_t x = 1;                                    // Newton: x <- x (2 - p_0 x), doubling the correct low bits
        for (int i = 0; i < 6; i++) x = x * (2u - p_limb(0) * x);
        return x & WS_M29;
    }
    WS_HD static bool maybe_kp(const F29& a, uint32_t kmax) { return ((a.v[0] * p0_inv29()) & WS_M29) <= kmax; }   // (k = 0 included: strict forms)
    WS_HD static bool maybe_zero_wide(const F29& a) { return maybe_kp(a, 9); }      // sub_wide results: (0, 10p)
    WS_HD static bool maybe_zero_weak(const F29& a) { return maybe_kp(a, 3); }      // sub_weak results: (0, 4p)
    // is_zero on the STORED form (any representative below 2^256 of a value in [0, 2p)): all words zero -- what every writer of a
    // point table stores for infinity -- or p itself (a non-canonical input), kept behind a real branch
    WS_HD static bool packed_is_zero(const Fe& x) {
        const uint32_t lo = (uint32_t)x.l[0];
        if (((uint32_t)(x.l[0] >> 32) | lo | (uint32_t)x.l[1] | (uint32_t)(x.l[1] >> 32) | (uint32_t)x.l[2] | (uint32_t)(x.l[2] >> 32) |
             (uint32_t)x.l[3] | (uint32_t)(x.l[3] >> 32)) == 0)
            return true;
        if (__builtin_expect(lo == (uint32_t)P::P0, 0)) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(WSNARK_EMUL)
            asm volatile("");           // (not to be folded into the common path as a chain of selects)
#endif
            return x.l[0] == P::P0 && x.l[1] == P::P1 && x.l[2] == P::P2 && x.l[3] == P::P3;
        }
        return false;
    }
    // pin an element's limbs in registers HERE (the accumulation loop unpacks a prefetched point before the registers it arrived in are
    // loaded again; without this the compiler sinks the unpacking below the loads and copies the sixteen words instead)
    WS_HD static void keep(F29& a) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(WSNARK_EMUL)
        asm volatile("" : "+v"(a.v[0]), "+v"(a.v[1]), "+v"(a.v[2]), "+v"(a.v[3]), "+v"(a.v[4]), "+v"(a.v[5]), "+v"(a.v[6]), "+v"(a.v[7]), "+v"(a.v[8]));
#endif
    }
    // r = s - k*p if s >= k*p else s (tight limbs in and out)
    WS_HD static F29 cond_sub_kp(const F29& s, uint32_t k) {
        F29 d;
        int32_t bw = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int32_t t = (int32_t)s.v[i] - (int32_t)kp_limb(k, i) + bw;
            d.v[i] = (uint32_t)t & WS_M29;
            bw = t >> 29;
        }
        const int32_t t8 = (int32_t)s.v[8] - (int32_t)kp_limb(k, 8) + bw;
        d.v[8] = (uint32_t)t8;
        const bool neg = t8 < 0;
        F29 r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.v[i] = neg ? s.v[i] : d.v[i];
        return r;
    }
    // wide (< 8p) -> [0, 2p)
    WS_HD static F29 narrow(const F29& a) { return cond_sub_kp(cond_sub_kp(a, 4), 2); }

    // ---- lazily reduced sums for the NTT butterflies: LDS-resident values live in [0, 4p) (tight limbs), sums are
    // formed WITHOUT a correction pass and folded back only where the bound would pass 16p; differences go straight
    // into a product ((a * b + m * p) / 2^261 < 1.2p for a < 16p, b < 2p).  Half the correction passes of add / sub. ----
    // a + b, carries propagated, no reduction: the bound just adds up
    WS_HD static F29 add_nr(const F29& a, const F29& b) {
        F29 s;
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t t = a.v[i] + b.v[i] + c;
            s.v[i] = t & WS_M29;
            c = t >> 29;
        }
        s.v[8] = a.v[8] + b.v[8] + c;
        return s;
    }
    // a - b + 8p for a, b < 8p: in (0, 16p), tight limbs
    WS_HD static F29 sub_weak8(const F29& a, const F29& b) {
        F29 d;
        int32_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int32_t t = (int32_t)a.v[i] - (int32_t)b.v[i] + (int32_t)kp_limb(8, i) + c;
            d.v[i] = (uint32_t)t & WS_M29;
            c = t >> 29;
        }
        d.v[8] = (uint32_t)((int32_t)a.v[8] - (int32_t)b.v[8] + (int32_t)kp_limb(8, 8) + c);
        return d;
    }
    WS_HD static F29 fold8(const F29& a) { return cond_sub_kp(a, 4); }                        // [0, 8p)  -> [0, 4p)
    WS_HD static F29 fold16(const F29& a) { return cond_sub_kp(cond_sub_kp(a, 8), 4); }       // [0, 16p) -> [0, 4p)
    WS_HD static F29 fold4to2(const F29& a) { return cond_sub_2p(a); }                        // [0, 4p)  -> [0, 2p)

    WS_HD static F29 neg(const F29& a) {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) o |= a.v[i];
        if (o == 0) return a;                 // 2p - 0 would leave [0, 2p)
        F29 d;
        int32_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int32_t t = (int32_t)p2_limb(i) - (int32_t)a.v[i] + c;
            d.v[i] = (uint32_t)t & WS_M29;
            c = t >> 29;
        }
        d.v[8] = (uint32_t)((int32_t)p2_limb(8) - (int32_t)a.v[8] + c);
        return d;
    }
    WS_HD static F29 cneg(const F29& a, bool s) { return s ? neg(a) : a; }

    // ---- Montgomery product a*b*2^-261 mod p; inputs < 2p (limbs < 2^29), output < 2p ----
    WS_HD static F29 mul(const F29& a, const F29& b) { return mont_mul29<P>(WS_A9(a), WS_A9(b)); }
    WS_HD static F29 sqr(const F29& a) { return mont_sqr29<P>(WS_A9(a)); }
    WS_HD static F29 mul_inl(const F29& a, const F29& b) { return mont_mul29_body<P>(a, b); }
    // (a*b + c*d) * 2^-261 mod p with ONE Montgomery reduction (the quadratic-extension product needs two
    // of these instead of three products and five additions).  Inputs < 2p, output < 2p:
    // (8p^2 + 2^261 p)/2^261 < 1.05p; columns <= 18 products + 9 reduction terms < 2^63.
    WS_HD static F29 mul2add(const F29& a, const F29& b, const F29& c, const F29& d) {
        return mont_mul2add29<P>(WS_A9(a), WS_A9(b), WS_A9(c), WS_A9(d));
    }

    // a*b - c*d with one reduction: a*b + (2p - c)*d.  a, b may be sub_weak results (< 4p), c, d < 2p:
    // (16 p^2 + 4 p^2 + 2^261 p) / 2^261 < 1.12 p.
    // 2p - c for c in [0, 2p): in (0, 2p], tight limbs; only ever an operand of a product
    WS_HD static F29 neg_weak(const F29& c) {
        F29 n;
        int32_t cy = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int32_t t = (int32_t)p2_limb(i) - (int32_t)c.v[i] + cy;
            n.v[i] = (uint32_t)t & WS_M29;
            cy = t >> 29;
        }
        n.v[8] = (uint32_t)((int32_t)p2_limb(8) - (int32_t)c.v[8] + cy);
        return n;
    }
    WS_HD static F29 mulsub2(const F29& a, const F29& b, const F29& c, const F29& d) {
        const F29 n = neg_weak(c);
#if WS_F29_MULSUB_INLINE
        return mul2add_inl(a, b, n, d);
#else
        return mont_mul2add29<P>(WS_A9(a), WS_A9(b), WS_A9(n), WS_A9(d));
#endif
    }
    // mul2add as an inlinable body (the quadratic extension's products: the call passes 36 operands, four of
    // them through scratch memory)
    WS_HD static F29 mul2add_inl(const F29& a, const F29& b, const F29& c, const F29& d) {
#ifdef WS_USE_MAD_CHAIN
        return mont_mul2add29_chain<P>(a, b, c, d);
#endif
        uint32_t m[9];
        uint64_t acc = 0;
        F29 r;
#pragma unroll
        for (int k = 0; k < 9; k++) {
#pragma unroll
            for (int i = 0; i <= k; i++) {
                acc += (uint64_t)a.v[i] * b.v[k - i];
                acc += (uint64_t)c.v[i] * d.v[k - i];
            }
#pragma unroll
            for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * p_limb(k - i);
            m[k] = ((uint32_t)acc * NP29) & WS_M29;
            acc += (uint64_t)m[k] * p_limb(0);
            acc >>= 29;
        }
#pragma unroll
        for (int k = 9; k < 17; k++) {
#pragma unroll
            for (int i = k - 8; i <= 8; i++) {
                acc += (uint64_t)a.v[i] * b.v[k - i];
                acc += (uint64_t)c.v[i] * d.v[k - i];
            }
#pragma unroll
            for (int i = k - 8; i <= 8; i++) acc += (uint64_t)m[i] * p_limb(k - i);
            r.v[k - 9] = (uint32_t)acc & WS_M29;
            acc >>= 29;
        }
        r.v[8] = (uint32_t)acc;
        return r;
    }
    // (a*b + c*d + e*f + g*h) * 2^-261 mod p with ONE reduction: one component of a*b - c*d in the quadratic
    // extension.  Operands < 2p with tight limbs: every column holds <= 36 products + 9 reduction terms
    // < 45 * 2^58 < 2^64; (16 p^2 + 2^261 p) / 2^261 < 1.1 p.  Inlined (eight operands do not fit the registers
    // of the calling convention).
    WS_HD static F29 mul4add(const F29& a, const F29& b, const F29& c, const F29& d, const F29& e, const F29& f,
                             const F29& g, const F29& h) {
#ifdef WS_USE_MAD_CHAIN
        return mont_mul4add29_chain<P>(a, b, c, d, e, f, g, h);
#endif
        uint32_t m[9];
        uint64_t acc = 0;
        F29 r;
#pragma unroll
        for (int k = 0; k < 9; k++) {
#pragma unroll
            for (int i = 0; i <= k; i++) {
                acc += (uint64_t)a.v[i] * b.v[k - i];
                acc += (uint64_t)c.v[i] * d.v[k - i];
                acc += (uint64_t)e.v[i] * f.v[k - i];
                acc += (uint64_t)g.v[i] * h.v[k - i];
            }
#pragma unroll
            for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * p_limb(k - i);
            m[k] = ((uint32_t)acc * NP29) & WS_M29;
            acc += (uint64_t)m[k] * p_limb(0);
            acc >>= 29;
        }
#pragma unroll
        for (int k = 9; k < 17; k++) {
#pragma unroll
            for (int i = k - 8; i <= 8; i++) {
                acc += (uint64_t)a.v[i] * b.v[k - i];
                acc += (uint64_t)c.v[i] * d.v[k - i];
                acc += (uint64_t)e.v[i] * f.v[k - i];
                acc += (uint64_t)g.v[i] * h.v[k - i];
            }
#pragma unroll
            for (int i = k - 8; i <= 8; i++) acc += (uint64_t)m[i] * p_limb(k - i);
            r.v[k - 9] = (uint32_t)acc & WS_M29;
            acc >>= 29;
        }
        r.v[8] = (uint32_t)acc;
        return r;
    }

    // canonical representative in [0, p) of a value in [0, 2p)
    // a^(p-2), square-and-multiply over the 254 bits of p - 2.  Key-load time only (msm_table_kernel); ~380 products.
    WS_HD static F29 inv(const F29& a) {
        const uint64_t e[4] = {P::P0 - 2, P::P1, P::P2, P::P3};
        F29 acc = one(), base = a;
        for (int i = 0; i < 254; i++) {
            if ((e[i >> 6] >> (i & 63)) & 1) acc = mul(acc, base);
            base = sqr(base);
        }
        return acc;
    }
    WS_HD static F29 canonical(const F29& a) {
        F29 d;
        int32_t bw = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int32_t t = (int32_t)a.v[i] - (int32_t)p_limb(i) + bw;
            d.v[i] = (uint32_t)t & WS_M29;
            bw = t >> 29;
        }
        const int32_t t8 = (int32_t)a.v[8] - (int32_t)p_limb(8) + bw;
        d.v[8] = (uint32_t)t8;
        const bool neg = t8 < 0;
        F29 r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.v[i] = neg ? a.v[i] : d.v[i];
        return r;
    }
    // reference format (canonical, Montgomery R = 2^256) -> internal (Montgomery R' = 2^261): x * 2^5
    WS_HD static F29 to_internal(const Fe& x) { return mul(unpack(x), from_words(P::CIN0, P::CIN1, P::CIN2, P::CIN3)); }
    // internal -> reference format: x * 2^-5, canonical
    WS_HD static Fe from_internal(const F29& a) { return pack(canonical(mul(a, from_words(P::R0, P::R1, P::R2_, P::R3)))); }
    // the multiplier that turns a reference-format table entry t into the internal-domain entry of
    // t * 2^k is itself a host-side job (see ntt.hip); only the device ops live here.
};

// ---- the multipliers: real functions (not inlined) taking operands by value in VGPRs ----

template <class P>
WS_NOINLINE_DEV F29 mont_mul29(WS_L9(a), WS_L9(b)) {
    const F29 a = {{a0, a1, a2, a3, a4, a5, a6, a7, a8}}, b = {{b0, b1, b2, b3, b4, b5, b6, b7, b8}};
    return mont_mul29_body<P>(a, b);
}
// the same product as an inlinable body: latency-bound kernels (the MSM reduction tail runs ~1 wavefront
// per SIMD) let the compiler interleave the independent products of one group addition
template <class P>
WS_HD F29 mont_mul29_body(const F29& a, const F29& b) {
#ifdef WS_USE_MAD_CHAIN
    return mont_mul29_chain<P>(a, b);
#endif
    typedef Field29<P> F;
    uint32_t m[9];
    uint64_t acc = 0;
    F29 r;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * F::p_limb(k - i);
        m[k] = ((uint32_t)acc * F::NP29) & WS_M29;
        acc += (uint64_t)m[k] * F::p_limb(0);
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i <= 8; i++) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
        for (int i = k - 8; i <= 8; i++) acc += (uint64_t)m[i] * F::p_limb(k - i);
        r.v[k - 9] = (uint32_t)acc & WS_M29;
        acc >>= 29;
    }
    r.v[8] = (uint32_t)acc;
    return r;
}

template <class P>
WS_NOINLINE_DEV F29 mont_mul2add29(WS_L9(a), WS_L9(b), WS_L9(c), WS_L9(d)) {
    typedef Field29<P> F;
    const F29 a = {{a0, a1, a2, a3, a4, a5, a6, a7, a8}}, b = {{b0, b1, b2, b3, b4, b5, b6, b7, b8}};
    const F29 c = {{c0, c1, c2, c3, c4, c5, c6, c7, c8}}, d = {{d0, d1, d2, d3, d4, d5, d6, d7, d8}};
#ifdef WS_USE_MAD_CHAIN
    return mont_mul2add29_chain<P>(a, b, c, d);
#endif
    uint32_t m[9];
    uint64_t acc = 0;
    F29 r;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)c.v[i] * d.v[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * F::p_limb(k - i);
        m[k] = ((uint32_t)acc * F::NP29) & WS_M29;
        acc += (uint64_t)m[k] * F::p_limb(0);
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i <= 8; i++) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
        for (int i = k - 8; i <= 8; i++) acc += (uint64_t)c.v[i] * d.v[k - i];
#pragma unroll
        for (int i = k - 8; i <= 8; i++) acc += (uint64_t)m[i] * F::p_limb(k - i);
        r.v[k - 9] = (uint32_t)acc & WS_M29;
        acc >>= 29;
    }
    r.v[8] = (uint32_t)acc;
    return r;
}

// squaring: off-diagonal products once, doubled (45 instead of 81 a*a products).  Operands up to 10p ("wide"): the doubled limbs
// stay below 2^30 (top one 2^26), a column holds at most 4 doubled products + a square + 9 reduction terms < 2^63.
template <class P>
WS_HD F29 mont_sqr29_body(const F29& a) {
#ifdef WS_USE_MAD_CHAIN
    return mont_sqr29_chain<P>(a);
#endif
    typedef Field29<P> F;
    uint32_t m[9], ad[9];
#pragma unroll
    for (int i = 0; i < 9; i++) ad[i] = a.v[i] << 1;
    uint64_t acc = 0;
    F29 r;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; 2 * i < k; i++) acc += (uint64_t)ad[i] * a.v[k - i];
        if ((k & 1) == 0) acc += (uint64_t)a.v[k / 2] * a.v[k / 2];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * F::p_limb(k - i);
        m[k] = ((uint32_t)acc * F::NP29) & WS_M29;
        acc += (uint64_t)m[k] * F::p_limb(0);
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; 2 * i < k; i++) acc += (uint64_t)ad[i] * a.v[k - i];
        if ((k & 1) == 0) acc += (uint64_t)a.v[k / 2] * a.v[k / 2];
#pragma unroll
        for (int i = k - 8; i <= 8; i++) acc += (uint64_t)m[i] * F::p_limb(k - i);
        r.v[k - 9] = (uint32_t)acc & WS_M29;
        acc >>= 29;
    }
    r.v[8] = (uint32_t)acc;
    return r;
}
template <class P>
WS_NOINLINE_DEV F29 mont_sqr29(WS_L9(a)) {
    const F29 a = {{a0, a1, a2, a3, a4, a5, a6, a7, a8}};
    return mont_sqr29_body<P>(a);
}

// variant whose products are inlined (see mont_mul29_body)
template <class P>
struct Field29I : Field29<P> {
    WS_HD static F29 mul(const F29& a, const F29& b) { return mont_mul29_body<P>(a, b); }
#ifndef WS_SQR_INL_DEDICATED
#define WS_SQR_INL_DEDICATED 1
#endif
#if WS_SQR_INL_DEDICATED
    WS_HD static F29 sqr(const F29& a) { return mont_sqr29_body<P>(a); }
#else
    WS_HD static F29 sqr(const F29& a) { return mont_mul29_body<P>(a, a); }
#endif
    WS_HD static F29 mulsub2(const F29& a, const F29& b, const F29& c, const F29& d) {
        return Field29<P>::mul2add_inl(a, b, Field29<P>::neg_weak(c), d);   // fused, like Field29's, but inlined
    }
};

typedef Field29<Fq29Params> Fq29;
typedef Field29I<Fq29Params> Fq29I;
typedef Field29<Fr29Params> Fr29;

}  // namespace wsnark
#include "mad_chain.h"
