// fp2.h -- Fq2 = Fq[u]/(u^2 + 1) for BN128 G2 coordinates.
// Replaces SURVEY.md section 8a row a6: f2m_mul/square/add/sub/neg/inverse/isZero/eq
// (/root/reference src/build_f2m.js:127-163, 186-227, 30-108, 353-383; the
// non-residue map is f1m_neg, src/bn128/build_bn128.js:40).
// Layout: (c0, c1) = 64 bytes, each Montgomery LE, as in the reference.
#pragma once
#include "field.h"

namespace wsnark {

struct alignas(16) Fe2 {
    Fe c0, c1;
};

struct Fq2 {
    typedef Fe2 El;
    WS_HD static Fe2 zero() { return Fe2{Fq::zero(), Fq::zero()}; }
    WS_HD static Fe2 one() { return Fe2{Fq::one(), Fq::zero()}; }
    WS_HD static bool is_zero(const Fe2& a) { return Fq::is_zero(a.c0) && Fq::is_zero(a.c1); }
    WS_HD static bool eq(const Fe2& a, const Fe2& b) { return Fq::eq(a.c0, b.c0) && Fq::eq(a.c1, b.c1); }
    WS_HD static Fe2 add(const Fe2& a, const Fe2& b) { return Fe2{Fq::add(a.c0, b.c0), Fq::add(a.c1, b.c1)}; }
    WS_HD static Fe2 dbl(const Fe2& a) { return Fe2{Fq::dbl(a.c0), Fq::dbl(a.c1)}; }
    WS_HD static Fe2 sub(const Fe2& a, const Fe2& b) { return Fe2{Fq::sub(a.c0, b.c0), Fq::sub(a.c1, b.c1)}; }
    WS_HD static Fe2 neg(const Fe2& a) { return Fe2{Fq::neg(a.c0), Fq::neg(a.c1)}; }
    WS_HD static Fe2 cneg(const Fe2& a, bool s) { return s ? neg(a) : a; }
    // Karatsuba, 3 base-field products (build_f2m.js:127-163)
    WS_HD static Fe2 mul(const Fe2& a, const Fe2& b) {
        Fe A = Fq::mul(a.c0, b.c0);
        Fe B = Fq::mul(a.c1, b.c1);
        Fe C = Fq::mul(Fq::add(a.c0, a.c1), Fq::add(b.c0, b.c1));
        return Fe2{Fq::sub(A, B), Fq::sub(C, Fq::add(A, B))};
    }
    // complex squaring, 2 base-field products (build_f2m.js:186-227)
    WS_HD static Fe2 sqr(const Fe2& a) {
        Fe AB = Fq::mul(a.c0, a.c1);
        Fe t = Fq::mul(Fq::add(a.c0, a.c1), Fq::sub(a.c0, a.c1));
        return Fe2{t, Fq::dbl(AB)};
    }
    // inverse via the norm (build_f2m.js:353-383)
    WS_HD static Fe2 inv(const Fe2& a) {
        Fe t = Fq::inv(Fq::add(Fq::sqr(a.c0), Fq::sqr(a.c1)));
        return Fe2{Fq::mul(a.c0, t), Fq::neg(Fq::mul(a.c1, t))};
    }
    WS_HD static Fe2 from_mont(const Fe2& a) { return Fe2{Fq::from_mont(a.c0), Fq::from_mont(a.c1)}; }
};

}  // namespace wsnark
