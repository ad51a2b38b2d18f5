// field.h -- 256-bit Montgomery prime-field arithmetic for BN128 (Fq, Fr), R = 2^256.
//
// One element = 4 x u64 limbs, little-endian, exactly the 32-byte layout of the
// reference (8 x u32 LE; /root/reference tools/buildpkey.js:57-77).  Every
// operation returns the canonical representative in [0,p) like the reference's
// f1m_* (src/build_f1m.js:67-113, 235-436), so intermediate arrays are
// bit-comparable with the reference.
//
// Replaces: f1m_mul/frm_mul, f1m_square,
// f1m_add/sub/neg, to/fromMontgomery, inverse.
//
// Shared by device kernels (one element per lane, limbs in VGPRs; the multiplier
// works on 32-bit half-limbs so each partial product is one v_mad_u64_u32) and by
// the host-side proof assembly (64-bit limbs with unsigned __int128).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#if defined(__HIP_DEVICE_COMPILE__)
#define WS_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define WS_HD __host__ __device__ inline
#endif
#else
#define WS_HD inline
#endif

#if defined(__HIP_DEVICE_COMPILE__) || defined(WSNARK_FORCE_32BIT_MUL)
#define WS_MUL32 1
#else
#define WS_MUL32 0
#endif

namespace wsnark {

struct alignas(16) Fe {
    uint64_t l[4];
};

// ---- parameters (src/bn128/build_bn128.js:19-20; constants SURVEY.md section 8) ----
struct FqParams {
    static constexpr uint64_t P0 = 0x3c208c16d87cfd47ull, P1 = 0x97816a916871ca8dull,
                              P2 = 0xb85045b68181585dull, P3 = 0x30644e72e131a029ull;
    static constexpr uint64_t NP = 0x87d20782e4866389ull;  // -p^-1 mod 2^64
    static constexpr uint64_t R0 = 0xd35d438dc58f0d9dull, R1 = 0x0a78eb28f5c70b3dull,   // 2^256 mod p
                              R2_ = 0x666ea36f7879462cull, R3 = 0x0e0a77c19a07df2full;
    static constexpr uint64_t RR0 = 0xf32cfc5b538afa89ull, RR1 = 0xb5e71911d44501fbull,  // 2^512 mod p
                              RR2 = 0x47ab1eff0a417ff6ull, RR3 = 0x06d89f71cab8351full;
};
struct FrParams {
    static constexpr uint64_t P0 = 0x43e1f593f0000001ull, P1 = 0x2833e84879b97091ull,
                              P2 = 0xb85045b68181585dull, P3 = 0x30644e72e131a029ull;
    static constexpr uint64_t NP = 0xc2e1f593efffffffull;
    static constexpr uint64_t R0 = 0xac96341c4ffffffbull, R1 = 0x36fc76959f60cd29ull,
                              R2_ = 0x666ea36f7879462eull, R3 = 0x0e0a77c19a07df2full;
    static constexpr uint64_t RR0 = 0x1bb8e645ae216da7ull, RR1 = 0x53fe3ab1e35c59e3ull,
                              RR2 = 0x8c49833d53bb8085ull, RR3 = 0x0216d0b17f4e44a5ull;
};

WS_HD uint64_t ws_addc(uint64_t a, uint64_t b, uint64_t cin, uint64_t* cout) {
    unsigned __int128 s = (unsigned __int128)a + b + cin;
    *cout = (uint64_t)(s >> 64);
    return (uint64_t)s;
}
WS_HD uint64_t ws_subb(uint64_t a, uint64_t b, uint64_t bin, uint64_t* bout) {
    unsigned __int128 d = (unsigned __int128)a - b - bin;
    *bout = (uint64_t)(d >> 64) & 1;
    return (uint64_t)d;
}


// 8 x 32-bit CIOS Montgomery product: every partial product is one 32x32+64 multiply-add
// (v_mad_u64_u32 on gfx950).  Deliberately a real (non-inlined) device function taking its
// operands by value in VGPRs: the curve formulas call it 10-40 times per group operation, and
// one ~5 KiB body keeps the kernels inside the instruction cache (and the build in minutes).
#if WS_MUL32
template <class P>
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __attribute__((noinline))
#else
inline
#endif
Fe mont_mul32(Fe a, Fe b) {
    const uint32_t p[8] = {(uint32_t)P::P0, (uint32_t)(P::P0 >> 32), (uint32_t)P::P1, (uint32_t)(P::P1 >> 32),
                           (uint32_t)P::P2, (uint32_t)(P::P2 >> 32), (uint32_t)P::P3, (uint32_t)(P::P3 >> 32)};
    const uint32_t np32 = (uint32_t)P::NP;
    uint32_t x[8], y[8], t[8];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        x[2 * i] = (uint32_t)a.l[i]; x[2 * i + 1] = (uint32_t)(a.l[i] >> 32);
        y[2 * i] = (uint32_t)b.l[i]; y[2 * i + 1] = (uint32_t)(b.l[i] >> 32);
    }
#pragma unroll
    for (int j = 0; j < 8; j++) t[j] = 0;
    uint32_t t8 = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            c = (uint64_t)x[j] * y[i] + t[j] + c;
            t[j] = (uint32_t)c;
            c >>= 32;
        }
        uint32_t top = t8 + (uint32_t)c;   // p < 2^254: never overflows
        uint32_t m = t[0] * np32;
        c = ((uint64_t)m * p[0] + t[0]) >> 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            c = (uint64_t)m * p[j] + t[j] + c;
            t[j - 1] = (uint32_t)c;
            c >>= 32;
        }
        c += top;
        t[7] = (uint32_t)c;
        t8 = (uint32_t)(c >> 32);
    }
    Fe r;
#pragma unroll
    for (int i = 0; i < 4; i++) r.l[i] = (uint64_t)t[2 * i] | ((uint64_t)t[2 * i + 1] << 32);
    // final conditional subtraction -> canonical representative
    uint64_t bw = 0;
    Fe s;
    s.l[0] = ws_subb(r.l[0], P::P0, bw, &bw);
    s.l[1] = ws_subb(r.l[1], P::P1, bw, &bw);
    s.l[2] = ws_subb(r.l[2], P::P2, bw, &bw);
    s.l[3] = ws_subb(r.l[3], P::P3, bw, &bw);
    return bw ? r : s;
}
#endif

template <class P>
struct Field {
    typedef Fe El;
    typedef Fe Packed;                             // global-memory form == register form here
    static constexpr bool kInternalDomain = false;
    WS_HD static Fe unpack(const Fe& x) { return x; }
    WS_HD static Fe pack(const Fe& x) { return x; }
    WS_HD static Fe to_internal(const Fe& x) { return x; }
    WS_HD static Fe from_internal(const Fe& x) { return x; }
    WS_HD static Fe canonical(const Fe& x) { return x; }
    WS_HD static Fe zero() { return Fe{{0, 0, 0, 0}}; }
    WS_HD static Fe one() { return Fe{{P::R0, P::R1, P::R2_, P::R3}}; }   // Montgomery 1
    WS_HD static Fe rsq() { return Fe{{P::RR0, P::RR1, P::RR2, P::RR3}}; }
    WS_HD static Fe modulus() { return Fe{{P::P0, P::P1, P::P2, P::P3}}; }

    WS_HD static bool is_zero(const Fe& a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
    WS_HD static bool eq(const Fe& a, const Fe& b) {
        return ((a.l[0] ^ b.l[0]) | (a.l[1] ^ b.l[1]) | (a.l[2] ^ b.l[2]) | (a.l[3] ^ b.l[3])) == 0;
    }

    // t - p if t >= p (t < 2p assumed)
    WS_HD static Fe reduce_once(const Fe& t) {
        uint64_t bw = 0;
        Fe s;
        s.l[0] = ws_subb(t.l[0], P::P0, bw, &bw);
        s.l[1] = ws_subb(t.l[1], P::P1, bw, &bw);
        s.l[2] = ws_subb(t.l[2], P::P2, bw, &bw);
        s.l[3] = ws_subb(t.l[3], P::P3, bw, &bw);
        Fe r;
        r.l[0] = bw ? t.l[0] : s.l[0];
        r.l[1] = bw ? t.l[1] : s.l[1];
        r.l[2] = bw ? t.l[2] : s.l[2];
        r.l[3] = bw ? t.l[3] : s.l[3];
        return r;
    }

    // build_f1m.js:67-84 (p < 2^254, so a+b never carries out of 256 bits)
    WS_HD static Fe add(const Fe& a, const Fe& b) {
        uint64_t c = 0;
        Fe t;
        t.l[0] = ws_addc(a.l[0], b.l[0], c, &c);
        t.l[1] = ws_addc(a.l[1], b.l[1], c, &c);
        t.l[2] = ws_addc(a.l[2], b.l[2], c, &c);
        t.l[3] = ws_addc(a.l[3], b.l[3], c, &c);
        return reduce_once(t);
    }
    WS_HD static Fe dbl(const Fe& a) { return add(a, a); }
    WS_HD static Fe add_lazy(const Fe& a, const Fe& b) { return add(a, b); }   // (no lazy form for the saturated limbs)
    static constexpr bool kHasMul2Add = false;
    WS_HD static Fe sub_weak(const Fe& a, const Fe& b) { return sub(a, b); }
    WS_HD static bool is_zero_weak(const Fe& a) { return is_zero(a); }
    WS_HD static Fe mulsub2(const Fe& a, const Fe& b, const Fe& c, const Fe& d) { return sub(mul(a, b), mul(c, d)); }
    WS_HD static Fe mul_inl(const Fe& a, const Fe& b) { return mul(a, b); }
    // (the "wide" forms of the radix-2^29 field, field29.h: strict here)
    WS_HD static Fe x3_wide(const Fe& rr, const Fe& ppp, const Fe& q) { return sub(sub(rr, ppp), dbl(q)); }
    WS_HD static Fe sub_wide(const Fe& a, const Fe& b) { return sub(a, b); }
    WS_HD static bool is_zero_wide(const Fe& a) { return is_zero(a); }
    WS_HD static Fe narrow(const Fe& a) { return a; }
    // the accumulation loop's cheap tests (curve.h: madd_fast; field29.h has the forms that matter): "may be zero" must hold for every
    // zero, "packed_is_zero" is is_zero on the stored form
    WS_HD static bool maybe_zero_wide(const Fe& a) { return is_zero(a); }
    WS_HD static bool maybe_zero_weak(const Fe& a) { return is_zero(a); }
    WS_HD static bool packed_is_zero(const Fe& a) { return is_zero(a); }
    WS_HD static void keep(Fe&) {}
    WS_HD static Fe sub_weak4(const Fe& a, const Fe& b) { return sub(a, b); }
    // (the lazily reduced butterfly sums of field29.h: strict here)
    WS_HD static Fe add_nr(const Fe& a, const Fe& b) { return add(a, b); }
    WS_HD static Fe sub_weak8(const Fe& a, const Fe& b) { return sub(a, b); }
    WS_HD static Fe fold8(const Fe& a) { return a; }
    WS_HD static Fe fold16(const Fe& a) { return a; }
    WS_HD static Fe fold4to2(const Fe& a) { return a; }
    WS_HD static Fe neg_weak(const Fe& a) { return neg(a); }
    WS_HD static Fe neg_weak4(const Fe& a) { return neg(a); }

    // build_f1m.js:86-100
    WS_HD static Fe sub(const Fe& a, const Fe& b) {
        uint64_t bw = 0;
        Fe t;
        t.l[0] = ws_subb(a.l[0], b.l[0], bw, &bw);
        t.l[1] = ws_subb(a.l[1], b.l[1], bw, &bw);
        t.l[2] = ws_subb(a.l[2], b.l[2], bw, &bw);
        t.l[3] = ws_subb(a.l[3], b.l[3], bw, &bw);
        uint64_t m = (uint64_t)0 - bw;   // all ones if borrowed
        uint64_t c = 0;
        Fe r;
        r.l[0] = ws_addc(t.l[0], P::P0 & m, c, &c);
        r.l[1] = ws_addc(t.l[1], P::P1 & m, c, &c);
        r.l[2] = ws_addc(t.l[2], P::P2 & m, c, &c);
        r.l[3] = ws_addc(t.l[3], P::P3 & m, c, &c);
        return r;
    }
    // build_f1m.js:102-113
    WS_HD static Fe neg(const Fe& a) {
        uint64_t bw = 0;
        Fe t;
        t.l[0] = ws_subb(P::P0, a.l[0], bw, &bw);
        t.l[1] = ws_subb(P::P1, a.l[1], bw, &bw);
        t.l[2] = ws_subb(P::P2, a.l[2], bw, &bw);
        t.l[3] = ws_subb(P::P3, a.l[3], bw, &bw);
        uint64_t nz = is_zero(a) ? 0 : ~(uint64_t)0;
        t.l[0] &= nz; t.l[1] &= nz; t.l[2] &= nz; t.l[3] &= nz;
        return t;
    }
    // conditional negate (sign != 0 -> -a)
    WS_HD static Fe cneg(const Fe& a, bool sign) { return sign ? neg(a) : a; }

    // Montgomery product a*b*2^-256 mod p, canonical (build_f1m.js:235-436).
    WS_HD static Fe mul(const Fe& a, const Fe& b) {
#if WS_MUL32
        return mont_mul32<P>(a, b);
#else
        const uint64_t p[4] = {P::P0, P::P1, P::P2, P::P3};
        uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
        for (int i = 0; i < 4; i++) {
            unsigned __int128 c;
            c = (unsigned __int128)a.l[0] * b.l[i] + t0; t0 = (uint64_t)c; c >>= 64;
            c += (unsigned __int128)a.l[1] * b.l[i] + t1; t1 = (uint64_t)c; c >>= 64;
            c += (unsigned __int128)a.l[2] * b.l[i] + t2; t2 = (uint64_t)c; c >>= 64;
            c += (unsigned __int128)a.l[3] * b.l[i] + t3; t3 = (uint64_t)c; c >>= 64;
            uint64_t t5 = (uint64_t)c + t4;
            uint64_t m = t0 * P::NP;
            c = (unsigned __int128)m * p[0] + t0; c >>= 64;
            c += (unsigned __int128)m * p[1] + t1; t0 = (uint64_t)c; c >>= 64;
            c += (unsigned __int128)m * p[2] + t2; t1 = (uint64_t)c; c >>= 64;
            c += (unsigned __int128)m * p[3] + t3; t2 = (uint64_t)c; c >>= 64;
            c += t5; t3 = (uint64_t)c; t4 = (uint64_t)(c >> 64);
        }
        return reduce_once(Fe{{t0, t1, t2, t3}});
#endif
    }
    WS_HD static Fe sqr(const Fe& a) { return mul(a, a); }   // build_f1m.js:439-736

    WS_HD static Fe to_mont(const Fe& a) { return mul(a, rsq()); }              // build_f1m.js:749-759
    WS_HD static Fe from_mont(const Fe& a) { return mul(a, Fe{{1, 0, 0, 0}}); } // build_f1m.js:761-770

    // a^(p-2): unique inverse, equal to the reference's ext-Euclid result (build_f1m.js:772-782)
    WS_HD static Fe inv(const Fe& a) {
        const uint64_t e[4] = {P::P0 - 2, P::P1, P::P2, P::P3};
        Fe acc = one(), base = a;
        for (int i = 0; i < 256; i++) {
            if ((e[i >> 6] >> (i & 63)) & 1) acc = mul(acc, base);
            base = sqr(base);
        }
        return acc;
    }
    WS_HD static Fe pow_u64(const Fe& a, uint64_t e) {
        Fe acc = one(), base = a;
        while (e) {
            if (e & 1) acc = mul(acc, base);
            base = sqr(base);
            e >>= 1;
        }
        return acc;
    }
    // reduce an arbitrary 256-bit integer mod p (plain domain): 2^256 / p < 6
    WS_HD static Fe reduce_full(const Fe& a) {
        Fe t = a;
        for (int k = 0; k < 6; k++) {
            uint64_t bw = 0;
            Fe s;
            s.l[0] = ws_subb(t.l[0], P::P0, bw, &bw);
            s.l[1] = ws_subb(t.l[1], P::P1, bw, &bw);
            s.l[2] = ws_subb(t.l[2], P::P2, bw, &bw);
            s.l[3] = ws_subb(t.l[3], P::P3, bw, &bw);
            if (!bw) t = s;
        }
        return t;
    }
};

typedef Field<FqParams> Fq;
typedef Field<FrParams> Fr;

}  // namespace wsnark
