// fp2.h -- quadratic extension Fq2 = Fq[u]/(u^2 + 1) for BN128 G2 coordinates, generic over the
// base-field implementation (Field<FqParams> on the host, Field29<Fq29Params> in the heavy kernels).
// Replaces SURVEY.md section 8a row a6: f2m_mul/square/add/sub/neg/inverse/isZero/eq
// (/root/reference src/build_f2m.js:127-163, 186-227, 30-108, 353-383; the
// non-residue map is f1m_neg, src/bn128/build_bn128.js:40).
// Layout: (c0, c1) = 64 bytes, each Montgomery LE, as in the reference.
#pragma once
#include <type_traits>

#include "field.h"
#include "rt.h"     // threadIdx / the emulator's pair exchange, for the lane-paired variant below

#ifndef WS_FP2_INLINE
#define WS_FP2_INLINE 1   // products of the quadratic extension as inlined bodies (G2 accumulate 4.15 -> 3.93 ms)
#endif

#ifndef WS_FP2_WEAK
#define WS_FP2_WEAK 1   // uncorrected differences as operands of the extension's products + one-pass X3 (0: round-1 strict forms, for A/B builds)
#endif

#ifndef WS_FP2_SQR_INLINE
#define WS_FP2_SQR_INLINE 1   // (G2 accumulate 3.76 -> 3.65 ms)
#endif

namespace wsnark {

template <class E>
struct alignas(16) Fe2T {
    E c0, c1;
};
typedef Fe2T<Fe> Fe2;

template <class B>
struct Fp2T {
    typedef typename B::El BE;
    typedef Fe2T<BE> El;
    typedef Fe2T<typename B::Packed> Packed;
    static constexpr bool kInternalDomain = B::kInternalDomain;
    WS_HD static El unpack(const Packed& x) { return El{B::unpack(x.c0), B::unpack(x.c1)}; }
    WS_HD static Packed pack(const El& x) { return Packed{B::pack(x.c0), B::pack(x.c1)}; }
    WS_HD static El to_internal(const Packed& x) { return El{B::to_internal(x.c0), B::to_internal(x.c1)}; }
    WS_HD static Packed from_internal(const El& x) { return Packed{B::from_internal(x.c0), B::from_internal(x.c1)}; }

    WS_HD static El zero() { return El{B::zero(), B::zero()}; }
    WS_HD static El one() { return El{B::one(), B::zero()}; }
    WS_HD static bool is_zero(const El& a) { return B::is_zero(a.c0) && B::is_zero(a.c1); }
    WS_HD static bool eq(const El& a, const El& b) { return B::eq(a.c0, b.c0) && B::eq(a.c1, b.c1); }
    WS_HD static El add(const El& a, const El& b) { return El{B::add(a.c0, b.c0), B::add(a.c1, b.c1)}; }
    WS_HD static El dbl(const El& a) { return El{B::dbl(a.c0), B::dbl(a.c1)}; }
    WS_HD static El sub(const El& a, const El& b) { return El{B::sub(a.c0, b.c0), B::sub(a.c1, b.c1)}; }
    // Uncorrected differences (components in (0, 4p), base field's sub_weak) are allowed where the curve formulas put them:
    // as the FIRST operand of mul, as the operand of sqr, and as the first two operands of mulsub2.  The second operand of
    // mul and the last two of mulsub2 stay strict (< 2p): their components are negated with the 2p form.
    // (For the saturated base field every weak form is the strict one.)
#if WS_FP2_WEAK
    WS_HD static El sub_weak(const El& a, const El& b) { return El{B::sub_weak(a.c0, b.c0), B::sub_weak(a.c1, b.c1)}; }
    WS_HD static bool is_zero_weak(const El& a) { return B::is_zero_weak(a.c0) && B::is_zero_weak(a.c1); }
    // X of an addition: one carry pass per component, then narrowed (the extension keeps its stored coordinates strict)
    WS_HD static El x3_wide(const El& rr, const El& ppp, const El& q) {
        return El{B::narrow(B::x3_wide(rr.c0, ppp.c0, q.c0)), B::narrow(B::x3_wide(rr.c1, ppp.c1, q.c1))};
    }
#else
    WS_HD static El sub_weak(const El& a, const El& b) { return sub(a, b); }
    WS_HD static bool is_zero_weak(const El& a) { return is_zero(a); }
    WS_HD static El x3_wide(const El& rr, const El& ppp, const El& q) { return sub(sub(rr, ppp), dbl(q)); }
#endif
    WS_HD static El sub_wide(const El& a, const El& b) { return sub_weak(a, b); }
    WS_HD static bool is_zero_wide(const El& a) { return is_zero_weak(a); }
    // (necessary conditions / the stored form's test, as in the base field: one component decides the cheap test)
    WS_HD static bool maybe_zero_weak(const El& a) { return B::maybe_zero_weak(a.c0); }
    WS_HD static bool maybe_zero_wide(const El& a) { return B::maybe_zero_weak(a.c0); }
    WS_HD static bool packed_is_zero(const Packed& a) { return B::packed_is_zero(a.c0) && B::packed_is_zero(a.c1); }
    WS_HD static void keep(El& a) { B::keep(a.c0); B::keep(a.c1); }
    WS_HD static El narrow(const El& a) { return a; }
    WS_HD static El neg(const El& a) { return El{B::neg(a.c0), B::neg(a.c1)}; }
    WS_HD static El cneg(const El& a, bool s) { return s ? neg(a) : a; }
    // (a0 + a1 u)(b0 + b1 u) = (a0 b0 - a1 b1) + (a0 b1 + a1 b0) u.  The reference uses Karatsuba with 3
    // base-field products (build_f2m.js:127-163); with a fused double product (one Montgomery reduction per
    // component) the radix-2^29 field does it in two calls and one negation: ~30 % fewer instructions.
    template <class BB = B>
    WS_HD static typename std::enable_if<BB::kHasMul2Add, El>::type mul(const El& a, const El& b) {
        const BE nb1 = B::neg_weak(b.c1);     // a: components < 4p allowed; b: strict.  (4p*2p)*2 = 16 p^2: < 1.1p after reduction
#if WS_FP2_INLINE
        return El{B::mul2add_inl(a.c0, b.c0, a.c1, nb1), B::mul2add_inl(a.c0, b.c1, a.c1, b.c0)};
#else
        return El{B::mul2add(a.c0, b.c0, a.c1, nb1), B::mul2add(a.c0, b.c1, a.c1, b.c0)};
#endif
    }
    template <class BB = B>
    WS_HD static typename std::enable_if<!BB::kHasMul2Add, El>::type mul(const El& a, const El& b) {
        BE A = B::mul(a.c0, b.c0);
        BE Bv = B::mul(a.c1, b.c1);
        BE C = B::mul(B::add_lazy(a.c0, a.c1), B::add_lazy(b.c0, b.c1));   // sums feed a product only
        return El{B::sub(A, Bv), B::sub(C, B::add(A, Bv))};
    }
    // a*b - c*d: with the fused four-product reduction two Montgomery passes instead of four products + a correction
    template <class BB = B>
    WS_HD static typename std::enable_if<BB::kHasMul2Add, El>::type mulsub2(const El& a, const El& b, const El& c, const El& d) {
        // a, b: components < 4p allowed (b.c1 is negated with the 4p form); c, d strict.  32 p^2 + 8 p^2 = 40 p^2: < 1.3p
        const BE nb1 = B::neg_weak4(b.c1), nc0 = B::neg_weak(c.c0), nc1 = B::neg_weak(c.c1);
        return El{B::mul4add(a.c0, b.c0, a.c1, nb1, nc0, d.c0, c.c1, d.c1),
                  B::mul4add(a.c0, b.c1, a.c1, b.c0, nc0, d.c1, nc1, d.c0)};
    }
    template <class BB = B>
    WS_HD static typename std::enable_if<!BB::kHasMul2Add, El>::type mulsub2(const El& a, const El& b, const El& c, const El& d) {
        return sub(mul(a, b), mul(c, d));
    }
    // complex squaring, 2 base-field products (build_f2m.js:186-227)
    WS_HD static El sqr(const El& a) {
#if WS_FP2_SQR_INLINE
        BE AB = B::mul_inl(a.c0, a.c1);      // a: components < 4p allowed: (a0 + a1) < 8p, (a0 - a1 + 4p) < 8p, 64 p^2: < 1.4p
        BE t = B::mul_inl(B::add_lazy(a.c0, a.c1), B::sub_weak4(a.c0, a.c1));
#else
        BE AB = B::mul(a.c0, a.c1);
        BE t = B::mul(B::add_lazy(a.c0, a.c1), B::sub_weak4(a.c0, a.c1));
#endif
        return El{t, B::dbl(AB)};
    }
    // inverse via the norm (build_f2m.js:353-383)
    WS_HD static El inv(const El& a) {
        BE t = B::inv(B::add(B::sqr(a.c0), B::sqr(a.c1)));
        return El{B::mul(a.c0, t), B::neg(B::mul(a.c1, t))};
    }
    WS_HD static El from_mont(const El& a) { return El{B::from_mont(a.c0), B::from_mont(a.c1)}; }
};

typedef Fp2T<Fq> Fq2;

// ---------------------------------------------------------------------------------------------------------------------
// The same extension with its two components on TWO ADJACENT LANES (2k holds c0, 2k + 1 holds c1): El is ONE base-field
// element per lane, so a G2 point is 36 VGPRs per lane instead of 72.  For the LATENCY-bound kernels of an MSM's reduction
// tail (msm.hip: chains of ~20-30 dependent additions on a few hundred wavefronts, where a G2 addition on a lone wavefront
// took 23 us at 256 VGPRs + ~100 AGPRs of spill): every extension product is still two fused base products, but each lane
// computes ONE of them -- half the dependent instruction chain per addition -- after fetching the partner's operands with a DPP
// quad_perm swap (9 v_mov_dpp per operand against 243 multiply-adds per fused product).  Not for throughput kernels: the
// instruction count per addition is the same plus the swaps.
//   * Both lanes of a pair must execute the same control flow: every predicate (is_zero, is_zero_weak, eq) is the AND over the
//     pair, so the curve formulas' branches are pair-uniform by construction.
//   * Packed = the base field's 32 bytes: in a stored Fp2 element (c0 | c1) lane p owns bytes [32 p, 32 p + 32).
//   * Same operand bounds as Fp2T above (first operand of mul / operand of sqr / first two of mulsub2 may be uncorrected
//     differences), same results bit for bit (tests: selftest impl 4).
// ---------------------------------------------------------------------------------------------------------------------
#if defined(WSNARK_EMUL)
#define WS_PAIR_SWAP_U32(v) (::hip_emul::pair_exchange(v))
#define WS_PAIR2_SWAP_U32(v) (::hip_emul::pair_exchange_dist((v), 2))
#define WS_PAIR_HI() ((threadIdx.x & 1u) != 0)
#define WS_PAIR2_HI() ((threadIdx.x & 2u) != 0)
#elif defined(__HIP_DEVICE_COMPILE__)
#define WS_PAIR_SWAP_U32(v) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true))
#define WS_PAIR2_SWAP_U32(v) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), 0x4E /* quad_perm [2,3,0,1] */, 0xF, 0xF, true))
#define WS_PAIR_HI() ((threadIdx.x & 1u) != 0)
#define WS_PAIR2_HI() ((threadIdx.x & 2u) != 0)
#else
#define WS_PAIR_SWAP_U32(v) (v)        // (host pass of hipcc: these functions are device-only)
#define WS_PAIR2_SWAP_U32(v) (v)
#define WS_PAIR_HI() (false)
#define WS_PAIR2_HI() (false)
#endif

template <class B>
struct Fp2PairT {
    typedef typename B::El El;
    typedef typename B::Packed Packed;
    static constexpr bool kInternalDomain = B::kInternalDomain;
    static constexpr bool kPaired = true;
    WS_HD static bool hi() { return WS_PAIR_HI(); }
    WS_HD static El swap(const El& a) {                 // the partner lane's component
        El r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.v[i] = WS_PAIR_SWAP_U32(a.v[i]);
        return r;
    }
    WS_HD static El sel(bool c, const El& x, const El& y) {
        El r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.v[i] = c ? x.v[i] : y.v[i];
        return r;
    }
    WS_HD static bool both(bool mine) {                 // (the exchange is unconditional: both lanes of the pair must execute it)
        const uint32_t other = WS_PAIR_SWAP_U32(mine ? 1u : 0u);
        return mine && other != 0;
    }

    WS_HD static El unpack(const Packed& x) { return B::unpack(x); }
    WS_HD static Packed pack(const El& x) { return B::pack(x); }
    WS_HD static El to_internal(const Packed& x) { return B::to_internal(x); }
    WS_HD static Packed from_internal(const El& x) { return B::from_internal(x); }
    WS_HD static El zero() { return B::zero(); }
    WS_HD static El one() { return hi() ? B::zero() : B::one(); }
    WS_HD static bool is_zero(const El& a) { return both(B::is_zero(a)); }
    WS_HD static bool is_zero_weak(const El& a) { return both(B::is_zero_weak(a)); }
    WS_HD static bool is_zero_wide(const El& a) { return is_zero_weak(a); }
    WS_HD static bool eq(const El& a, const El& b) { return is_zero(B::sub(a, b)); }
    WS_HD static El add(const El& a, const El& b) { return B::add(a, b); }
    WS_HD static El dbl(const El& a) { return B::dbl(a); }
    WS_HD static El sub(const El& a, const El& b) { return B::sub(a, b); }
    WS_HD static El sub_weak(const El& a, const El& b) { return B::sub_weak(a, b); }
    WS_HD static El sub_wide(const El& a, const El& b) { return B::sub_weak(a, b); }
    WS_HD static El narrow(const El& a) { return a; }
    WS_HD static El x3_wide(const El& rr, const El& ppp, const El& q) { return B::narrow(B::x3_wide(rr, ppp, q)); }
    WS_HD static El neg(const El& a) { return B::neg(a); }
    WS_HD static El cneg(const El& a, bool s) { return s ? B::neg(a) : a; }
    // (a0 + a1 u)(b0 + b1 u): lane 0 forms a0 b0 + a1 (2p - b1), lane 1 forms a0 b1 + a1 b0 -- one fused double product each
    WS_HD static El mul(const El& a, const El& b) {
        const bool h = hi();
        const El oa = swap(a), ob = swap(b);
        return B::mul2add_inl(sel(h, oa, a), b, sel(h, a, oa), sel(h, ob, B::neg_weak(ob)));
    }
    // complex squaring: lane 0 forms (a0 + a1)(a0 - a1 + 4p), lane 1 forms 2 (a0 a1)
    WS_HD static El sqr(const El& a) {
        const bool h = hi();
        const El oa = swap(a);
        const El t = B::mul_inl(sel(h, oa, B::add_lazy(a, oa)), sel(h, a, B::sub_weak4(a, oa)));
        return h ? B::dbl(t) : t;
    }
    // a b - c d: one fused four-product reduction per lane (operands as in Fp2T::mulsub2)
    WS_HD static El mulsub2(const El& a, const El& b, const El& c, const El& d) {
        const bool h = hi();
        const El oa = swap(a), ob = swap(b), oc = swap(c), od = swap(d);
        return B::mul4add(sel(h, oa, a), b, sel(h, a, oa), sel(h, ob, B::neg_weak4(ob)), B::neg_weak(sel(h, oc, c)), d,
                          sel(h, B::neg_weak(c), oc), od);
    }
};


}  // namespace wsnark
