// curve.h -- BN128 group arithmetic (y^2 = x^3 + b, a = 0), generic over the
// coordinate field F (Fq for G1, Fq2 for G2).
//
// Replaces SURVEY.md section 8a rows a7-a10 (/root/reference
// src/build_curve_jacobian_a0.js:174-235 double, :280-385 add, :33-172/:387-457
// zero/neg/affine, src/build_timesscalar.js:20-80 timesScalar).  The reference
// works in Jacobian (x,y,z) with full 11M+5S additions; this build accumulates in
// XYZZ coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2) with MIXED additions of the
// key's affine points (8M+2S) -- a build-side optimisation: results are only ever
// compared after affine normalisation (src/bn128.js:706-712), where the group
// element's coordinates are unique.
//
// Conventions kept from the reference: an affine input with x == 0 is the point
// at infinity (src/build_multiexp.js:335-349); infinity in XYZZ is ZZ == 0; the
// canonical affine/Jacobian infinity written out is (0, 1, 0) (:87-113).
#pragma once
#include "field.h"
#include "fp2.h"
#include "field29.h"

namespace wsnark {

template <class F>
struct Affine {
    typename F::El x, y;
};
template <class F>
struct XYZZ {
    typename F::El x, y, zz, zzz;
};
template <class F>
struct Jac {
    typename F::El x, y, z;
};
// global-memory forms (32-byte coordinates; identical to the above for the 4x64 fields)
template <class F>
struct AffineP {
    typename F::Packed x, y;
};
template <class F>
struct XYZZP {
    typename F::Packed x, y, zz, zzz;
};

template <class F>
struct Curve {
    typedef F Field;
    typedef typename F::El El;
    typedef Affine<F> Aff;
    typedef XYZZ<F> Pt;
    typedef AffineP<F> AffP;
    typedef XYZZP<F> PtP;

    // packed <-> register forms (no domain change)
    WS_HD static Aff unpack_aff(const AffP& a) { return Aff{F::unpack(a.x), F::unpack(a.y)}; }
    WS_HD static AffP pack_aff(const Aff& a) { return AffP{F::pack(a.x), F::pack(a.y)}; }
    WS_HD static Pt unpack_pt(const PtP& p) { return Pt{F::unpack(p.x), F::unpack(p.y), F::unpack(p.zz), F::unpack(p.zzz)}; }
    WS_HD static PtP pack_pt(const Pt& p) { return PtP{F::pack(p.x), F::pack(p.y), F::pack(p.zz), F::pack(p.zzz)}; }
    // reference format (canonical Montgomery R = 2^256) <-> the field's internal domain
    WS_HD static Aff aff_to_internal(const AffP& a) { return Aff{F::to_internal(a.x), F::to_internal(a.y)}; }
    WS_HD static PtP pt_from_internal(const Pt& p) {
        return PtP{F::from_internal(p.x), F::from_internal(p.y), F::from_internal(p.zz), F::from_internal(p.zzz)};
    }

    WS_HD static Pt infinity() { return Pt{F::zero(), F::one(), F::zero(), F::zero()}; }
    WS_HD static bool is_inf(const Pt& p) { return F::is_zero(p.zz); }
    WS_HD static bool aff_is_inf(const Aff& a) { return F::is_zero(a.x); }
    WS_HD static Pt from_affine(const Aff& a) {
        if (aff_is_inf(a)) return infinity();
        return Pt{a.x, a.y, F::one(), F::one()};
    }
    WS_HD static Pt neg(const Pt& p) { return Pt{p.x, F::neg(p.y), p.zz, p.zzz}; }

    // dbl-2008-s-1 (a = 0).  ZZ == 0 or Y == 0 yield ZZ3 == 0 (infinity) by themselves.
    WS_HD static Pt dbl(const Pt& p) {
        El U = F::dbl(p.y);
        El V = F::sqr(U);
        El W = F::mul(U, V);
        El S = F::mul(p.x, V);
        El X2 = F::sqr(p.x);
        El M = F::add(F::dbl(X2), X2);
        El X3 = F::sub(F::sqr(M), F::dbl(S));
        El Y3 = F::mulsub2(M, F::sub_weak(S, X3), W, p.y);
        return Pt{X3, Y3, F::mul(V, p.zz), F::mul(W, p.zzz)};
    }
    // doubling of an affine point (mdbl-2008-s-1)
    WS_HD static Pt dbl_affine(const El& x, const El& y) {
        El U = F::dbl(y);
        El V = F::sqr(U);
        El W = F::mul(U, V);
        El S = F::mul(x, V);
        El X2 = F::sqr(x);
        El M = F::add(F::dbl(X2), X2);
        El X3 = F::sub(F::sqr(M), F::dbl(S));
        El Y3 = F::mulsub2(M, F::sub_weak(S, X3), W, y);
        return Pt{X3, Y3, V, W};
    }

    // acc += (+/-) affine point  (madd-2008-s, 8M+2S) with all corner cases:
    // affine infinity (x == 0) -> no-op; acc infinity -> copy; equal -> double;
    // opposite -> infinity.  (reference branches: build_curve_jacobian_a0.js:322-356)
    WS_HD static void madd(Pt& acc, const Aff& a, bool negate) {
        if (aff_is_inf(a)) return;
        El y2 = F::cneg(a.y, negate);
        if (is_inf(acc)) {
            acc = Pt{a.x, y2, F::one(), F::one()};
            return;
        }
        El U2 = F::mul(a.x, acc.zz);
        El S2 = F::mul(y2, acc.zzz);
        // P, R and (Q - X3) only feed products (and the zero tests): the uncorrected "weak" difference
        // saves the second carry pass on the radix-2^29 field (plain sub elsewhere)
        El P = F::sub_weak(U2, acc.x);
        El R = F::sub_weak(S2, acc.y);
        if (F::is_zero_weak(P)) {
            if (F::is_zero_weak(R)) acc = dbl_affine(a.x, y2);
            else acc = infinity();
            return;
        }
        El PP = F::sqr(P);
        El PPP = F::mul(P, PP);
        El Q = F::mul(acc.x, PP);
        El X3 = F::sub(F::sub(F::sqr(R), PPP), F::dbl(Q));
        El Y3 = F::mulsub2(R, F::sub_weak(Q, X3), acc.y, PPP);
        acc.x = X3;
        acc.y = Y3;
        acc.zz = F::mul(acc.zz, PP);
        acc.zzz = F::mul(acc.zzz, PPP);
    }

    // The same addition for the accumulation loop, with acc.x kept WIDE between additions (field29.h: X3 in one carry
    // pass instead of two corrected subtractions and a corrected doubling, ~110 of the ~2900 instructions of an addition;
    // for fields without wide forms this is madd).  acc.y, acc.zz, acc.zzz stay strict.  Call narrow_x() before the
    // point leaves the loop (packing needs x < 2^256).
    WS_HD static void madd_wide(Pt& acc, const Aff& a, bool negate) {
        if (aff_is_inf(a)) return;
        El y2 = F::cneg(a.y, negate);
        if (is_inf(acc)) {
            acc = Pt{a.x, y2, F::one(), F::one()};
            return;
        }
        El U2 = F::mul(a.x, acc.zz);
        El S2 = F::mul(y2, acc.zzz);
        El P = F::sub_wide(U2, acc.x);
        El R = F::sub_weak(S2, acc.y);
        if (F::is_zero_wide(P)) {
            if (F::is_zero_weak(R)) acc = dbl_affine(a.x, y2);
            else acc = infinity();
            return;
        }
        El PP = F::sqr(P);
        El PPP = F::mul(P, PP);
        El Q = F::mul(acc.x, PP);
        El X3 = F::x3_wide(F::sqr(R), PPP, Q);
        El Y3 = F::mulsub2(R, F::sub_wide(Q, X3), acc.y, PPP);
        acc.x = X3;
        acc.y = Y3;
        acc.zz = F::mul(acc.zz, PP);
        acc.zzz = F::mul(acc.zzz, PPP);
    }
    WS_HD static void narrow_x(Pt& acc) { acc.x = F::narrow(acc.x); }
    // The accumulation loop's COMMON case as one straight path (round 6): a finite acc (x wide, as madd_wide keeps it) plus a finite
    // affine point that is neither acc nor -acc.  Returns false WITHOUT touching acc when the pair may be one of the reference's corner
    // cases (build_curve_jacobian_a0.js:322-356: the cheap necessary test of field29.h on x2 zz1 - x1), and the caller runs the pair
    // through madd_wide.  Why a second form: with madd_wide's early returns inlined in a loop the compiler merges five definitions of
    // the accumulator at the loop's end -- 70 register moves, 36 constant loads, two nine-limb zero tests and a branch tree per
    // addition, ~6 % of the loop's issue cycles (ISA of msm_accumulate, round 6) -- for paths a proof never takes.
    WS_HD static bool madd_fast(Pt& acc, const Aff& a, bool negate) {
        El y2 = F::cneg(a.y, negate);
        El U2 = F::mul(a.x, acc.zz);
        El S2 = F::mul(y2, acc.zzz);
        El P = F::sub_wide(U2, acc.x);
        if (F::maybe_zero_wide(P)) return false;
        El R = F::sub_weak(S2, acc.y);
        El PP = F::sqr(P);
        El PPP = F::mul(P, PP);
        El Q = F::mul(acc.x, PP);
        El X3 = F::x3_wide(F::sqr(R), PPP, Q);
        El Y3 = F::mulsub2(R, F::sub_wide(Q, X3), acc.y, PPP);
        acc.x = X3;
        acc.y = Y3;
        acc.zz = F::mul(acc.zz, PP);
        acc.zzz = F::mul(acc.zzz, PPP);
        return true;
    }
    // The SECOND entry of a task meets a sum that is still an affine point (zz = zzz = 1): mmadd-2008-s, the same formulas without the
    // four products by zz / zzz (4M + 2S; ZZ3 = PP, ZZZ3 = PPP).  `p` is that point (x, y strict); same contract as madd_fast.
    WS_HD static bool mmadd_fast(Pt& acc, const Aff& p, const Aff& a, bool negate) {
        El y2 = F::cneg(a.y, negate);
        El P = F::sub_wide(a.x, p.x);
        if (F::maybe_zero_wide(P)) return false;
        El R = F::sub_weak(y2, p.y);
        El PP = F::sqr(P);
        El PPP = F::mul(P, PP);
        El Q = F::mul(p.x, PP);
        El X3 = F::x3_wide(F::sqr(R), PPP, Q);
        El Y3 = F::mulsub2(R, F::sub_wide(Q, X3), p.y, PPP);
        acc = Pt{X3, Y3, PP, PPP};
        return true;
    }

    // full addition (add-2008-s, 12M+2S) with all corner cases
    WS_HD static Pt add(const Pt& a, const Pt& b) {
        if (is_inf(a)) return b;
        if (is_inf(b)) return a;
        El U1 = F::mul(a.x, b.zz);
        El U2 = F::mul(b.x, a.zz);
        El S1 = F::mul(a.y, b.zzz);
        El S2 = F::mul(b.y, a.zzz);
        El P = F::sub_weak(U2, U1);
        El R = F::sub_weak(S2, S1);
        if (F::is_zero_weak(P)) {
            if (F::is_zero_weak(R)) return dbl(a);
            return infinity();
        }
        El PP = F::sqr(P);
        El PPP = F::mul(P, PP);
        El Q = F::mul(U1, PP);
        El X3 = F::sub(F::sub(F::sqr(R), PPP), F::dbl(Q));
        El Y3 = F::mulsub2(R, F::sub_weak(Q, X3), S1, PPP);
        return Pt{X3, Y3, F::mul(F::mul(a.zz, b.zz), PP), F::mul(F::mul(a.zzz, b.zzz), PPP)};
    }

    // k * p for a small unsigned k (MSB-first double-and-add)
    WS_HD static Pt mul_small(const Pt& p, uint32_t k) {
        Pt r = infinity();
        for (int i = 31; i >= 0; i--) {
            r = dbl(r);
            if ((k >> i) & 1) r = add(r, p);
        }
        return r;
    }

    // MSB-first double-and-add over `nbytes` little-endian scalar bytes
    // (the evaluation order of build_timesscalar.js:20-80; any order gives the same group element)
    WS_HD static Pt mul_bytes(const Pt& p, const uint8_t* scalar, int nbytes) {
        Pt r = infinity();
        for (int i = nbytes * 8 - 1; i >= 0; i--) {
            r = dbl(r);
            if ((scalar[i >> 3] >> (i & 7)) & 1) r = add(r, p);
        }
        return r;
    }

    // affine normalisation into the reference's Jacobian-Montgomery triple:
    // infinity -> (0, 1, 0), else (x, y, 1)   (build_curve_jacobian_a0.js:421-457)
    WS_HD static Jac<F> to_affine_jac(const Pt& p) {
        if (is_inf(p)) return Jac<F>{F::zero(), F::one(), F::zero()};
        El inv = F::inv(F::mul(p.zz, p.zzz));
        El izz = F::mul(inv, p.zzz);    // 1/ZZ
        El izzz = F::mul(inv, p.zz);    // 1/ZZZ
        return Jac<F>{F::mul(p.x, izz), F::mul(p.y, izzz), F::one()};
    }
    // Jacobian (x,y,z) -> XYZZ (X, Y, z^2, z^3)
    WS_HD static Pt from_jac(const Jac<F>& j) {
        if (F::is_zero(j.z)) return infinity();
        El zz = F::sqr(j.z);
        return Pt{j.x, j.y, zz, F::mul(zz, j.z)};
    }
};

typedef Curve<Fq> G1;
typedef Curve<Fq2> G2;
// device-side curves over the carry-free radix-2^29 field (heavy kernels only)
// G1 kernels use the variant with inlined products (WS_G1_INLINE=0: products as calls of the noinline functions).
// Measured on MI355X: accumulate 1.31 -> 1.25-1.30 ms, and the fused Y3 in the inlined tails 0.34 -> 0.32 ms.
#ifndef WS_G1_INLINE
#define WS_G1_INLINE 1
#endif
#if WS_G1_INLINE
typedef Curve<Fq29I> G1R29;
#else
typedef Curve<Fq29> G1R29;
#endif
typedef Curve<Fq29I> G1R29I;   // inlined products: reduction-tail kernels
typedef Curve<Fp2T<Fq29>> G2R29;
typedef Curve<Fp2PairT<Fq29>> G2P29;   // G2 with the extension's components on two adjacent lanes: reduction-tail kernels (fp2.h)

// How the lanes of a kernel map to STORED points: one lane per point, or -- the paired extension field -- two lanes per point,
// lane p of the pair owning component p of every coordinate (bytes [32 p, 32 p + 32) of each 64-byte Fp2 element).
template <class C>
struct PointIO {
    static constexpr uint32_t LPP = 1;                 // lanes per point
    typedef typename C::PtP Stored;
    WS_HD static typename C::Pt load(const Stored* a, uint64_t i) { return C::unpack_pt(a[i]); }
    WS_HD static void store(Stored* a, uint64_t i, const typename C::Pt& p) { a[i] = C::pack_pt(p); }
    WS_HD static void store_ref(Stored* a, uint64_t i, const typename C::Pt& p) { a[i] = C::pt_from_internal(p); }   // reference format
};
template <class B>
struct PointIO<Curve<Fp2PairT<B>>> {
    typedef Curve<Fp2PairT<B>> C;
    typedef Fp2PairT<B> F;
    static constexpr uint32_t LPP = 2;
    typedef XYZZP<Fp2T<B>> Stored;                     // the full 256-byte point, as the one-lane kernels store it
    WS_HD static typename C::Pt load(const Stored* a, uint64_t i) {
        const typename B::Packed* f = reinterpret_cast<const typename B::Packed*>(a + i) + (F::hi() ? 1 : 0);
        return typename C::Pt{F::unpack(f[0]), F::unpack(f[2]), F::unpack(f[4]), F::unpack(f[6])};
    }
    WS_HD static void store(Stored* a, uint64_t i, const typename C::Pt& p) {
        typename B::Packed* f = reinterpret_cast<typename B::Packed*>(a + i) + (F::hi() ? 1 : 0);
        f[0] = F::pack(p.x); f[2] = F::pack(p.y); f[4] = F::pack(p.zz); f[6] = F::pack(p.zzz);
    }
    WS_HD static void store_ref(Stored* a, uint64_t i, const typename C::Pt& p) {
        typename B::Packed* f = reinterpret_cast<typename B::Packed*>(a + i) + (F::hi() ? 1 : 0);
        f[0] = F::from_internal(p.x); f[2] = F::from_internal(p.y); f[4] = F::from_internal(p.zz); f[6] = F::from_internal(p.zzz);
    }
};

}  // namespace wsnark
