// curve_pair.h -- G1 with ONE XYZZ point on TWO adjacent lanes (round 6), for the reduction-tail kernels of msm.hip.
//
// The tail of a sum (msm_chunks / msm_tree / msm_rows) is chains of dependent FULL additions (add-2008-s: 12M + 2S = 14 products)
// on one or two wavefronts per SIMD: what it costs is the length of the chain times the time of ONE addition.  The fourteen products
// come in seven independent pairs, so two lanes that hold half a point each run an addition in SEVEN product steps:
//
//      lane 2k ("lo") holds (X, ZZ)            lane 2k + 1 ("hi") holds (Y, ZZZ)
//      1   U1 = X1 ZZ2                          S1 = Y1 ZZZ2
//      2   U2 = X2 ZZ1                          S2 = Y2 ZZZ1                    P = U2 - U1 | R = S2 - S1   (uncorrected differences)
//      3   PP = P^2                             RR = R^2
//      4   PPP = P PP                           ZZZ12 = ZZZ1 ZZZ2               hi <- PPP (swap)
//      5   Q = U1 PP                            ZZZ3 = ZZZ12 PPP                lo <- RR  (swap);  lo: X3 = RR - PPP - 2 Q,  w = Q - X3
//      6   ZZ12 = ZZ1 ZZ2                       T = S1 PPP                      hi <- w   (swap)
//      7   ZZ3 = ZZ12 PP                        Y3 = R w - T
//
// Both lanes of a pair sit in one wavefront and execute the SAME instructions: every step is one product whose operands are picked
// by the lane's parity (a bit-select per limb), the three swaps are DPP quad_perm moves (9 v_mov_dpp each), and every branch
// condition (infinity operands, P == Q, P == -Q: the reference's cases, /root/reference src/build_curve_jacobian_a0.js:322-356) is
// made pair-uniform by exchanging the flag.  Same buffers as the one-lane kernels (128-byte XYZZ points: lo reads / writes bytes
// [0, 32) and [64, 96), hi [32, 64) and [96, 128)), same group elements; the representation (X, Y, ZZ, ZZZ) it leaves is the one
// add-2008-s leaves.  The doubling case -- the two operands are the same point: never on random data, but the parity tests plant it --
// gathers the whole point on both lanes and runs the one-lane doubling redundantly.
#pragma once
#include "curve.h"
#include "fp2.h"      // WS_PAIR_SWAP_U32 / WS_PAIR_HI

namespace wsnark {

// DIST = 1: the two halves on lanes 2k, 2k + 1 (G1: B = the base field).  DIST = 2: on lanes 4k + {0, 1} and 4k + {2, 3}, for a base
// "field" that is itself spread over lane pairs -- B = Fp2PairT<Fq29>: a G2 point on FOUR lanes, lane (h2, h1) holding component h1 of
// (X, ZZ) (h2 = 0) or of (Y, ZZZ) (h2 = 1); every step is then one extension product = one fused double product per lane.
template <class B, int DIST = 1>
struct CurvePairG1 {
    typedef B Field;
    typedef typename B::El El;
    typedef Curve<B> Full;                          // the one-lane curve: layouts and the rare doubling
    struct Pt { El a, b; };                         // lo: (X, ZZ)   hi: (Y, ZZZ)
    typedef typename Full::PtP PtP;

    WS_HD static bool hi() { return DIST == 1 ? WS_PAIR_HI() : WS_PAIR2_HI(); }
    WS_HD static uint32_t swap_u32(uint32_t v) { return DIST == 1 ? WS_PAIR_SWAP_U32(v) : WS_PAIR2_SWAP_U32(v); }
    WS_HD static El swap(const El& x) {
        El r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.v[i] = swap_u32(x.v[i]);
        return r;
    }
    // the lane's own choice between two values: x on lo, y on hi -- by arithmetic on a 0 / ~0 mask (a run of nine v_cndmask_b32 on one
    // standing mask issues at ~20 cycles each on gfx950: profiles/r06_product_variants.txt)
    WS_HD static El pick(uint32_t hi_mask, const El& x, const El& y) {
        El r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.v[i] = x.v[i] ^ ((x.v[i] ^ y.v[i]) & hi_mask);
        return r;
    }
    WS_HD static bool from_lo(bool mine) {          // the LO lane's flag, on both lanes (unconditional exchange)
        const uint32_t other = swap_u32(mine ? 1u : 0u);
        return hi() ? other != 0 : mine;
    }
    WS_HD static bool from_hi(bool mine) {
        const uint32_t other = swap_u32(mine ? 1u : 0u);
        return hi() ? mine : other != 0;
    }

    WS_HD static Pt infinity() { return hi() ? Pt{B::one(), B::zero()} : Pt{B::zero(), B::zero()}; }      // (0, 1, 0, 0)
    WS_HD static bool is_inf(const Pt& p) { return from_lo(B::is_zero(p.b)); }                           // ZZ == 0
    // this lane's half of a whole point, and the whole point from the pair's halves
    WS_HD static Pt split(const typename Full::Pt& p) { return hi() ? Pt{p.y, p.zzz} : Pt{p.x, p.zz}; }
    WS_HD static typename Full::Pt join(const Pt& p) {
        const El oa = swap(p.a), ob = swap(p.b);
        return hi() ? typename Full::Pt{oa, p.a, ob, p.b} : typename Full::Pt{p.a, oa, p.b, ob};
    }
    WS_HD static Pt dbl(const Pt& p) { return split(Full::dbl(join(p))); }

    WS_HD static Pt add(const Pt& p, const Pt& q) {
        if (is_inf(p)) return q;
        if (is_inf(q)) return p;
        const uint32_t m = hi() ? 0xFFFFFFFFu : 0u;
        const El t1 = B::mul(p.a, q.b);                                  // U1 | S1
        const El t2 = B::mul(q.a, p.b);                                  // U2 | S2
        const El d = B::sub_weak(t2, t1);                                // P  | R
        const bool dz = B::is_zero_weak(d);
        if (from_lo(dz)) {                                               // P == 0: the same x
            if (from_hi(dz)) return dbl(p);                              // ... and the same y: the doubling
            return infinity();                                           // P == -Q
        }
        const El sq = B::sqr(d);                                         // PP | RR
        const El t4 = B::mul(pick(m, d, p.b), pick(m, sq, q.b));         // PPP | ZZZ1 ZZZ2
        const El o4 = swap(t4);                                          // (hi: PPP)
        const El t5 = B::mul(pick(m, t1, t4), pick(m, sq, o4));          // Q | ZZZ3
        const El osq = swap(sq);                                         // (lo: RR)
        const El x3 = B::sub(B::sub(osq, t4), B::dbl(t5));               // lo: X3 = RR - PPP - 2 Q   (hi: unused)
        const El w = B::sub(t5, x3);                                     // lo: Q - X3   (strict: it is a SECOND operand below, and the lane-paired
                                                                         //  extension field negates its second operand's components)
        const El t6 = B::mul(pick(m, p.b, t1), pick(m, q.b, o4));        // ZZ1 ZZ2 | S1 PPP
        const El ow = swap(w);                                           // (hi: Q - X3)
        const El t7 = B::mul(pick(m, t6, d), pick(m, sq, ow));           // ZZ3 | R (Q - X3)
        const El y3 = B::sub(t7, t6);                                    // hi: Y3
        return Pt{pick(m, x3, y3), pick(m, t7, t5)};
    }
};

typedef CurvePairG1<Fq29I> G1P29;
typedef CurvePairG1<Fp2PairT<Fq29>, 2> G2Q29;       // G2 on four lanes

template <class B>
struct PointIO<CurvePairG1<B, 1>> {
    typedef CurvePairG1<B, 1> C;
    static constexpr uint32_t LPP = 2;
    typedef XYZZP<B> Stored;                         // the full 128-byte point, as the one-lane kernels store it
    WS_HD static typename C::Pt load(const Stored* a, uint64_t i) {
        const typename B::Packed* f = reinterpret_cast<const typename B::Packed*>(a + i) + (C::hi() ? 1 : 0);
        return typename C::Pt{B::unpack(f[0]), B::unpack(f[2])};
    }
    WS_HD static void store(Stored* a, uint64_t i, const typename C::Pt& p) {
        typename B::Packed* f = reinterpret_cast<typename B::Packed*>(a + i) + (C::hi() ? 1 : 0);
        f[0] = B::pack(p.a); f[2] = B::pack(p.b);
    }
    WS_HD static void store_ref(Stored* a, uint64_t i, const typename C::Pt& p) {
        typename B::Packed* f = reinterpret_cast<typename B::Packed*>(a + i) + (C::hi() ? 1 : 0);
        f[0] = B::from_internal(p.a); f[2] = B::from_internal(p.b);
    }
};

// four lanes per 256-byte G2 point: lane (h2, h1) reads / writes component h1 of coordinates (x, zz) or (y, zzz)
template <class BB>
struct PointIO<CurvePairG1<Fp2PairT<BB>, 2>> {
    typedef Fp2PairT<BB> F;
    typedef CurvePairG1<F, 2> C;
    static constexpr uint32_t LPP = 4;
    typedef XYZZP<Fp2T<BB>> Stored;
    WS_HD static typename C::Pt load(const Stored* a, uint64_t i) {
        const typename BB::Packed* f = reinterpret_cast<const typename BB::Packed*>(a + i) + (F::hi() ? 1 : 0) + (C::hi() ? 2 : 0);
        return typename C::Pt{F::unpack(f[0]), F::unpack(f[4])};
    }
    WS_HD static void store(Stored* a, uint64_t i, const typename C::Pt& p) {
        typename BB::Packed* f = reinterpret_cast<typename BB::Packed*>(a + i) + (F::hi() ? 1 : 0) + (C::hi() ? 2 : 0);
        f[0] = F::pack(p.a); f[4] = F::pack(p.b);
    }
    WS_HD static void store_ref(Stored* a, uint64_t i, const typename C::Pt& p) {
        typename BB::Packed* f = reinterpret_cast<typename BB::Packed*>(a + i) + (F::hi() ? 1 : 0) + (C::hi() ? 2 : 0);
        f[0] = F::from_internal(p.a); f[4] = F::from_internal(p.b);
    }
};

}  // namespace wsnark
