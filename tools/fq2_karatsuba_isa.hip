// fq2_karatsuba_isa.hip -- VERDICT r4 item 1(d): the Fq2 product as three schoolbook products with ONE lazy reduction per component
// (Karatsuba on the unreduced 64-bit columns) against the shipped form (fp2.h: two fused double products, 4 schoolbook products,
// one reduction per component), on the radix-2^29 field.  Not part of the product.
//   host check:  g++ -O2 -std=c++17 -x c++ -DWSNARK_EMUL -Itests/emul -Iwasmsnark_amd/csrc tools/fq2_karatsuba_isa.hip -o /tmp/k && /tmp/k
//                (the two forms agree mod p on random and on extreme operands)
//   ISA counts:  tools/fq2_karatsuba_isa.sh   ->  profiles/r05_fq2_karatsuba_isa.txt
// The point of the exercise: on gfx950 v_mad_u64_u32 issues at the full wave rate (the product chain reaches 0.98 of the issue
// slots, bench.py: int_alu_peaks_this_run), so a schoolbook product (81 multiply-adds) is no dearer per instruction than the
// 64-bit column subtractions Karatsuba pays for the product it saves: 243 + 8 x 17 + 18 against 324.
#include <stdint.h>
#include <stdio.h>

#include "rt.h"
#include "curve.h"

namespace wsnark {

// (a0 + a1 u)(b0 + b1 u), u^2 = -1, operands strict (< 2p, tight limbs).  Columns are SIGNED 64-bit sums:
//   c0_k = t0_k - t1_k,  c1_k = t2_k - t0_k - t1_k,   t0 = a0 b0, t1 = a1 b1, t2 = (a0 + a1)(b0 + b1)  (limbs of the sums < 2^30:
//   nine products < 2^60 per column + nine reduction terms < 2^58 stay below 2^64 / 2).
// Each component: the Montgomery reduction of mul2add_inl on a signed accumulator; the result lies in (-1.1p, 2.2p) and is
// brought to [0, 2p) by one conditional addition of 2p and the usual conditional subtraction.
template <class P>
WS_HD void fq2_mul_karatsuba(const F29& a0, const F29& a1, const F29& b0, const F29& b1, F29* c0, F29* c1) {
    typedef Field29<P> B;
    uint32_t sa[9], sb[9];
#pragma unroll
    for (int i = 0; i < 9; i++) { sa[i] = a0.v[i] + a1.v[i]; sb[i] = b0.v[i] + b1.v[i]; }
    uint32_t m0[9], m1[9];
    int64_t acc0 = 0, acc1 = 0;
    int32_t r0[9], r1[9];
#pragma unroll
    for (int k = 0; k < 17; k++) {
        int64_t t0 = 0, t1 = 0, t2 = 0;
        const int lo = k < 9 ? 0 : k - 8, hi = k < 9 ? k : 8;
#pragma unroll
        for (int i = lo; i <= hi; i++) {
            t0 += (int64_t)((uint64_t)a0.v[i] * b0.v[k - i]);
            t1 += (int64_t)((uint64_t)a1.v[i] * b1.v[k - i]);
            t2 += (int64_t)((uint64_t)sa[i] * sb[k - i]);
        }
        acc0 += t0 - t1;
        acc1 += t2 - t0 - t1;
        if (k < 9) {
#pragma unroll
            for (int i = 0; i < k; i++) { acc0 += (int64_t)((uint64_t)m0[i] * B::p_limb(k - i)); acc1 += (int64_t)((uint64_t)m1[i] * B::p_limb(k - i)); }
            m0[k] = ((uint32_t)acc0 * B::NP29) & WS_M29;
            m1[k] = ((uint32_t)acc1 * B::NP29) & WS_M29;
            acc0 += (int64_t)((uint64_t)m0[k] * B::p_limb(0));
            acc1 += (int64_t)((uint64_t)m1[k] * B::p_limb(0));
            acc0 >>= 29; acc1 >>= 29;
        } else {
#pragma unroll
            for (int i = k - 8; i <= 8; i++) { acc0 += (int64_t)((uint64_t)m0[i] * B::p_limb(k - i)); acc1 += (int64_t)((uint64_t)m1[i] * B::p_limb(k - i)); }
            r0[k - 9] = (int32_t)((uint32_t)acc0 & WS_M29);
            r1[k - 9] = (int32_t)((uint32_t)acc1 & WS_M29);
            acc0 >>= 29; acc1 >>= 29;
        }
    }
    r0[8] = (int32_t)acc0;      // (the top limb carries the sign)
    r1[8] = (int32_t)acc1;
    // negative -> + 2p; then into [0, 2p)
    auto fix = [](const int32_t* r) -> F29 {
        const bool neg = r[8] < 0;
        F29 s;
        int32_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int32_t t = r[i] + (neg ? (int32_t)B::p2_limb(i) : 0) + c;
            s.v[i] = (uint32_t)t & WS_M29;
            c = t >> 29;
        }
        s.v[8] = (uint32_t)(r[8] + (neg ? (int32_t)B::p2_limb(8) : 0) + c);
        return B::cond_sub_2p(B::cond_sub_2p(s));
    };
    *c0 = fix(r0);
    *c1 = fix(r1);
}

#ifndef WSNARK_EMUL
__global__ void k_shipped(const F29* __restrict__ in, F29* __restrict__ out) {
    typedef Fp2T<Fq29> F2;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const F2::El a{in[4 * t], in[4 * t + 1]}, b{in[4 * t + 2], in[4 * t + 3]};
    const F2::El c = F2::mul(a, b);
    out[2 * t] = c.c0; out[2 * t + 1] = c.c1;
}
__global__ void k_karatsuba(const F29* __restrict__ in, F29* __restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    F29 c0, c1;
    fq2_mul_karatsuba<Fq29Params>(in[4 * t], in[4 * t + 1], in[4 * t + 2], in[4 * t + 3], &c0, &c1);
    out[2 * t] = c0; out[2 * t + 1] = c1;
}
#endif

}  // namespace wsnark

#ifdef WSNARK_EMUL
using namespace wsnark;
static uint64_t rng_state = 88172645463325252ull;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
int main() {
    typedef Fq29 B;
    typedef Fp2T<Fq29> F2;
    int bad = 0;
    for (int it = 0; it < 20000; it++) {
        F29 x[4];
        for (auto& e : x) {
            Fe v{{rnd(), rnd(), rnd(), rnd() >> 3}};
            if (it % 7 == 0) v = Fe{{~0ull, ~0ull, ~0ull, 0x30644e72e131a029ull}};      // large: close to p
            if (it % 11 == 0) v = Fe{{it % 22 ? 0ull : 1ull, 0, 0, 0}};
            e = B::to_internal(Fq::reduce_full(v));                                          // strict operand
        }
        const F2::El c = F2::mul(F2::El{x[0], x[1]}, F2::El{x[2], x[3]});
        F29 k0, k1;
        fq2_mul_karatsuba<Fq29Params>(x[0], x[1], x[2], x[3], &k0, &k1);
        const Fe s0 = B::from_internal(c.c0), s1 = B::from_internal(c.c1), t0 = B::from_internal(k0), t1 = B::from_internal(k1);
        if (memcmp(&s0, &t0, 32) || memcmp(&s1, &t1, 32)) bad++;
    }
    printf("fq2 karatsuba (lazy, signed columns) vs the shipped product: %d mismatches in 20000\n", bad);
    return bad != 0;
}
#endif
