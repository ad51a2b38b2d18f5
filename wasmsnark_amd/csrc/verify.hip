// verify.hip -- native Groth16 verification on the host.
//
// Replaces /root/reference src/bn128.js:722-791 (Bn128.groth16Verify) and the pairing it calls,
// bn128_pairingEq4 (src/bn128/build_bn128.js:265-1374: prepareG1/prepareG2, Miller loop, final exponentiation over the
// Fp2 / Fp6 / Fp12 tower of build_f2m.js / build_f3m.js).  Same check, same verdicts:
//     e(A, B) * e(-IC(inputs), gamma2) * e(-C, delta2) * e(-alfa1, beta2) == 1,      false if an input is >= r (:772)
// The verdict of a pairing-product equation does not depend on WHICH bilinear non-degenerate pairing on G1 x G2 is
// used, so this build does not restate the reference's optimal-ate code with its Frobenius tables.  It evaluates the
// plain ate pairing  a(Q, P) = f_{T,Q}(P)^((p^12-1)/r),  T = t - 1 = p - r  (127 bits),  with
//   * Fp12 = Fp2[w]/(w^6 - xi), xi = 9 + u (one flat degree-6 extension; w^2 = v, v^3 = xi is the reference's tower),
//   * the untwist (x', y') -> (x' w^2, y' w^3) of the D-type twist  y^2 = x^3 + 3/xi  (build_bn128.js:79-90),
//   * affine Miller steps: the line through the untwisted points evaluated at P = (xP, yP) is
//         l = yP - lambda' xP * w + (lambda' xT' - yT') * w^3                  (three non-zero coefficients)
//     where lambda' is the slope on the twist; vertical lines lie in a proper subfield and vanish in the final
//     exponentiation,
//   * ONE final exponentiation of the product of the four Miller values by plain square-and-multiply with the
//     2790-bit exponent (p^12 - 1)/r: no Frobenius constants anywhere.
// Host arithmetic (field.h / fp2.h, the same headers the proof assembly uses); ~30 ms per verification on one core.
// "Same verdict under any pairing" holds for points of G1 x G2 only: for a B on the twist but outside the order-r
// subgroup, or for off-curve coordinates, neither this map nor the reference's is bilinear and the two could disagree
// (ADVICE r2).  The reference makes no such checks (its verdict on malformed points is an accident of its Miller loop);
// here every proof and key point must satisfy its curve equation and every G2 point must be killed by r, otherwise the
// proof is INVALID (*valid = 0) before any pairing is evaluated.  Like the reference (setG1Affine / setG2Affine force
// z = 1, src/bn128.js:741-760) the z coordinates of the proof are ignored: (x, y) is the point, so the (0, 1, 0) a
// prover prints for infinity is the off-curve point (0, 1) and makes the proof invalid.
#include <string.h>

#include "../../include/wsnark.h"
#include "internal.h"

namespace wsnark {
namespace {

typedef Fe2 F2;   // Fq2 element, Montgomery form

struct F12 {
    F2 c[6];      // sum c[i] w^i,  w^6 = xi
};

inline F2 f2_mul_xi(const F2& a) {          // (a0 + a1 u)(9 + u) = (9 a0 - a1) + (9 a1 + a0) u
    Fe a0_2 = Fq::dbl(a.c0), a0_4 = Fq::dbl(a0_2), a0_8 = Fq::dbl(a0_4), a0_9 = Fq::add(a0_8, a.c0);
    Fe a1_2 = Fq::dbl(a.c1), a1_4 = Fq::dbl(a1_2), a1_8 = Fq::dbl(a1_4), a1_9 = Fq::add(a1_8, a.c1);
    return F2{Fq::sub(a0_9, a.c1), Fq::add(a1_9, a.c0)};
}
inline F12 f12_one() {
    F12 r;
    for (auto& x : r.c) x = Fq2::zero();
    r.c[0] = Fq2::one();
    return r;
}
inline bool f12_is_one(const F12& a) {
    if (!Fq2::eq(a.c[0], Fq2::one())) return false;
    for (int i = 1; i < 6; i++) if (!Fq2::is_zero(a.c[i])) return false;
    return true;
}
F12 f12_mul(const F12& a, const F12& b) {
    F2 lo[6], hi[5];
    for (auto& x : lo) x = Fq2::zero();
    for (auto& x : hi) x = Fq2::zero();
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) {
            const F2 t = Fq2::mul(a.c[i], b.c[j]);
            if (i + j < 6) lo[i + j] = Fq2::add(lo[i + j], t);
            else hi[i + j - 6] = Fq2::add(hi[i + j - 6], t);
        }
    F12 r;
    for (int k = 0; k < 6; k++) r.c[k] = k < 5 ? Fq2::add(lo[k], f2_mul_xi(hi[k])) : lo[k];
    return r;
}
// a * (l0 + l1 w + l3 w^3): the line's three non-zero coefficients
F12 f12_mul_line(const F12& a, const F2& l0, const F2& l1, const F2& l3) {
    F2 lo[6], hi[5];
    for (auto& x : lo) x = Fq2::zero();
    for (auto& x : hi) x = Fq2::zero();
    const F2* L[3] = {&l0, &l1, &l3};
    const int deg[3] = {0, 1, 3};
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 3; j++) {
            const F2 t = Fq2::mul(a.c[i], *L[j]);
            const int k = i + deg[j];
            if (k < 6) lo[k] = Fq2::add(lo[k], t);
            else hi[k - 6] = Fq2::add(hi[k - 6], t);
        }
    F12 r;
    for (int k = 0; k < 6; k++) r.c[k] = k < 5 ? Fq2::add(lo[k], f2_mul_xi(hi[k])) : lo[k];
    return r;
}

// (p^12 - 1) / r, little-endian 64-bit words (2790 bits)
const uint64_t kFinalExp[44] = {
    0x86964b64ca86f120ull, 0x40a4efb7e54523a4ull, 0x837fa97896e84abbull, 0x361102b6b9b2b918ull,
    0xc0de81def35692daull, 0xbe04c7e8a6c3c760ull, 0xd766f9c9d570bb7full, 0xc230974d83561841ull,
    0x5bba1668c3be69a3ull, 0x7f3811c410526294ull, 0x29baee7ddadda71cull, 0xbf813b8d145da900ull,
    0x641bbadf423f9a2cull, 0xa80bb4ea44eacc5eull, 0xcd65664814fde37cull, 0x4a0364b9580291d2ull,
    0xee93dfb10826f0ddull, 0x6b42db8dc5514724ull, 0xbb10cf430b0f3785ull, 0x40494e406f804216ull,
    0x55cfe107acf3aafbull, 0x2088ec80e0ebae87ull, 0x846a3ed011a337a0ull, 0x48a45a4a1e3a5195ull,
    0xe5664568dfc50e16ull, 0xab6a41294c0cc4ebull, 0x82d0d602d268c7daull, 0x6668449aed3cc48aull,
    0x5062cd0fb2015dfcull, 0x7f2940a8b1ddb3d1ull, 0x77f5b63a2a226448ull, 0xfef0781361e443aeull,
    0xf977870e88d5c6c8ull, 0x790364a61f676baaull, 0x5887e72eceaddea3ull, 0x1377e563a09a1b70ull,
    0x0c54efee1bd8c3b2ull, 0x3ec3d15ad524d8f7ull, 0xdaf15466b2383a5dull, 0xe1e30a73bb94fec0ull,
    0x6a1c71015f3f7be2ull, 0x842d43bf6369b1ffull, 0x20fddadf107d20bcull, 0x0000002f4b6dc970ull,
};
F12 final_exponentiation(const F12& f) {
    F12 acc = f12_one();
    bool started = false;
    for (int i = 44 * 64 - 1; i >= 0; i--) {
        if (started) acc = f12_mul(acc, acc);
        if ((kFinalExp[i >> 6] >> (i & 63)) & 1) { acc = started ? f12_mul(acc, f) : f; started = true; }
    }
    return acc;
}

struct G1A { Fe x, y; bool inf; };      // affine, Montgomery
struct G2A { F2 x, y; bool inf; };

// f_{T,Q}(P) for T = p - r, affine steps on the twist.  Returns false if a step degenerates (Q not of order r).
bool miller_ate(const G2A& Q, const G1A& P, F12* out) {
    *out = f12_one();
    if (Q.inf || P.inf) return true;                          // e(O, .) = e(., O) = 1
    static const uint64_t T[2] = {0xf83e9682e87cfd46ull, 0x6f4d8248eeb859fbull};   // p - r
    const F2 xP = F2{P.x, Fq::zero()}, yP = F2{P.y, Fq::zero()};
    F2 tx = Q.x, ty = Q.y;
    F12 f = f12_one();
    for (int i = 125; i >= 0; i--) {                          // bit 126 is the leading one
        // doubling step: lambda = 3 x^2 / (2 y)
        if (Fq2::is_zero(ty)) return false;
        const F2 x2 = Fq2::sqr(tx);
        const F2 lam = Fq2::mul(Fq2::add(Fq2::dbl(x2), x2), Fq2::inv(Fq2::dbl(ty)));
        f = f12_mul(f, f);
        f = f12_mul_line(f, yP, Fq2::neg(Fq2::mul(lam, xP)), Fq2::sub(Fq2::mul(lam, tx), ty));
        const F2 nx = Fq2::sub(Fq2::sqr(lam), Fq2::dbl(tx));
        ty = Fq2::sub(Fq2::mul(lam, Fq2::sub(tx, nx)), ty);
        tx = nx;
        if ((T[i >> 6] >> (i & 63)) & 1) {
            // addition step with Q: lambda = (yT - yQ) / (xT - xQ)
            const F2 dx = Fq2::sub(tx, Q.x);
            if (Fq2::is_zero(dx)) return false;
            const F2 l2 = Fq2::mul(Fq2::sub(ty, Q.y), Fq2::inv(dx));
            f = f12_mul_line(f, yP, Fq2::neg(Fq2::mul(l2, xP)), Fq2::sub(Fq2::mul(l2, Q.x), Q.y));
            const F2 ax = Fq2::sub(Fq2::sub(Fq2::sqr(l2), tx), Q.x);
            ty = Fq2::sub(Fq2::mul(l2, Fq2::sub(Q.x, ax)), Q.y);
            tx = ax;
        }
    }
    *out = f;
    return true;
}

// plain 32-byte LE integers -> Montgomery; false if a coordinate is >= q
bool load_fq(const uint8_t* p, Fe* out) {
    Fe v;
    memcpy(&v, p, 32);
    const Fe red = Fq::reduce_full(v);
    if (!Fq::eq(red, v)) return false;
    *out = Fq::to_mont(v);
    return true;
}
// (x, y[, z]) plain coordinates; the z of a proof element must be a reduced field element but is otherwise ignored
// (the reference forces z = 1).  Key points (no z): (0, 0) stands for infinity, as in proving keys.
bool load_g1(const uint8_t* p, bool has_z, G1A* out) {
    Fe z = Fq::one();
    if (!load_fq(p, &out->x) || !load_fq(p + 32, &out->y) || (has_z && !load_fq(p + 64, &z))) return false;
    out->inf = !has_z && Fq::is_zero(out->x) && Fq::is_zero(out->y);
    return true;
}
bool load_g2(const uint8_t* p, bool has_z, G2A* out) {
    F2 z = Fq2::one();
    if (!load_fq(p, &out->x.c0) || !load_fq(p + 32, &out->x.c1) || !load_fq(p + 64, &out->y.c0) || !load_fq(p + 96, &out->y.c1)) return false;
    if (has_z && (!load_fq(p + 128, &z.c0) || !load_fq(p + 160, &z.c1))) return false;
    out->inf = !has_z && Fq2::is_zero(out->x) && Fq2::is_zero(out->y);
    return true;
}
// y^2 == x^3 + 3 (G1 has cofactor 1: on the curve is in the group)
bool g1_ok(const G1A& P) {
    if (P.inf) return true;
    const Fe three = Fq::to_mont(Fe{{3, 0, 0, 0}});
    return Fq::eq(Fq::sqr(P.y), Fq::add(Fq::mul(Fq::sqr(P.x), P.x), three));
}
// on the twist y^2 == x^3 + 3/(9 + u) (src/bn128/build_bn128.js:79-90) AND in the order-r subgroup: [r] Q == O
bool g2_ok(const G2A& Q) {
    if (Q.inf) return true;
    static const F2 b2 = Fq2::mul(F2{Fq::to_mont(Fe{{3, 0, 0, 0}}), Fq::zero()}, Fq2::inv(F2{Fq::to_mont(Fe{{9, 0, 0, 0}}), Fq::one()}));
    if (!Fq2::eq(Fq2::sqr(Q.y), Fq2::add(Fq2::mul(Fq2::sqr(Q.x), Q.x), b2))) return false;
    const Fe r = Fr::modulus();
    const G2::Pt rq = G2::mul_bytes(G2::Pt{Q.x, Q.y, Fq2::one(), Fq2::one()}, reinterpret_cast<const uint8_t*>(&r), 32);
    return G2::is_inf(rq);
}

}  // namespace

// vk: alfa1 (64 B) | beta2 (128 B) | gamma2 (128 B) | delta2 (128 B) | IC[0 .. n_inputs] (64 B each); all affine, PLAIN LE.
int groth16_verify(const uint8_t* vk, size_t vk_len, const uint8_t* inputs, uint64_t n_inputs, const uint8_t* proof384, int* valid) {
    *valid = 0;
    // (compared without the multiplication: (n_inputs + 1) * 64 wraps for n_inputs near 2^58)
    if (vk_len < 512 || n_inputs > (vk_len - 448) / 64 - 1) { set_last_error("verification key has fewer IC points than inputs + 1"); return WS_ERR_SIZE; }
    G1A alfa1, A, C;
    G2A beta2, gamma2, delta2, B;
    if (!load_g1(vk, false, &alfa1) || !load_g2(vk + 64, false, &beta2) || !load_g2(vk + 192, false, &gamma2) ||
        !load_g2(vk + 320, false, &delta2) || !load_g1(proof384, true, &A) || !load_g2(proof384 + 96, true, &B) ||
        !load_g1(proof384 + 288, true, &C)) {
        set_last_error("verify: a coordinate is not a reduced field element");
        return WS_ERR_FORMAT;
    }
    // malformed points: invalid, before anything is paired (see the header)
    if (!g1_ok(alfa1) || !g1_ok(A) || !g1_ok(C) || !g2_ok(beta2) || !g2_ok(gamma2) || !g2_ok(delta2) || !g2_ok(B)) return WS_OK;
    // IC(inputs) = IC[0] + sum input_i * IC[i+1]   (src/bn128.js:765-777; an input >= r makes the proof invalid, :772)
    G1::Pt acc = G1::infinity();
    for (uint64_t i = 0; i <= n_inputs; i++) {
        G1A ic;
        if (!load_g1(vk + 448 + i * 64, false, &ic)) { set_last_error("verify: IC coordinate not reduced"); return WS_ERR_FORMAT; }
        if (!g1_ok(ic)) return WS_OK;
        G1::Pt pt = ic.inf ? G1::infinity() : G1::Pt{ic.x, ic.y, Fq::one(), Fq::one()};
        if (i > 0) {
            Fe s;
            memcpy(&s, inputs + (i - 1) * 32, 32);
            if (!Fr::eq(Fr::reduce_full(s), s)) return WS_OK;                       // >= r: invalid, not an error
            pt = G1::mul_bytes(pt, reinterpret_cast<const uint8_t*>(&s), 32);
        }
        acc = G1::add(acc, pt);
    }
    const Jac<Fq> icj = G1::to_affine_jac(acc);
    G1A IC{icj.x, icj.y, Fq::is_zero(icj.z)};
    auto neg = [](G1A p) { p.y = Fq::neg(p.y); return p; };
    F12 f = f12_one(), m;
    const G1A Ps[4] = {A, neg(IC), neg(C), neg(alfa1)};
    const G2A Qs[4] = {B, gamma2, delta2, beta2};
    for (int k = 0; k < 4; k++) {
        if (!miller_ate(Qs[k], Ps[k], &m)) return WS_OK;      // a G2 point outside the order-r subgroup: invalid
        f = f12_mul(f, m);
    }
    *valid = f12_is_one(final_exponentiation(f)) ? 1 : 0;
    return WS_OK;
}

}  // namespace wsnark
