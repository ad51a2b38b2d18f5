// product_variants_isa.hip -- VERDICT r5 "next round" 5: the 9-limb radix-2^29 Montgomery product with its a x b part as a Karatsuba split
// into 5 + 4 limbs (3 x (5 x 5) limb products instead of 9 x 9 = 81), against the shipped schoolbook form (field29.h / mad_chain.h).
// Not part of the product.   ISA counts by issue class + costs from the measured class rates:  tools/product_variants_isa.sh
//   host check:  g++ -O2 -std=c++17 -x c++ -DWSNARK_EMUL -Itests/emul -Iwasmsnark_amd/csrc tools/product_variants_isa.hip -o /tmp/pv && /tmp/pv
//
// a = aL + aH X^5, b = bL + bH X^5 (X = 2^29; aH, bH have 4 limbs, padded to 5 with a zero):
//   a b = zL + [ (aL + aH)(bL + bH) - zL - zH ] X^5 + zH X^10,   zL = aL bL (9 columns), zH = aH bH (7 columns)
// The limbs of the sums are < 2^30, a column of the middle product holds <= 5 products < 2^60: below 2^63.  The unreduced 64-bit
// columns of the three products are combined (64-bit additions / subtractions, signed) into the 17 columns of a x b, to which the
// Montgomery reduction of the shipped product (81 m x p multiply-adds, unchanged: p has no structure to split) is then applied.
#include <stdint.h>
#include <stdio.h>

#include "rt.h"
#include "curve.h"

namespace wsnark {

template <class P>
WS_HD F29 mont_mul29_karatsuba(const F29& a, const F29& b) {
    typedef Field29<P> F;
    uint32_t sa[5], sb[5];
#pragma unroll
    for (int i = 0; i < 5; i++) { sa[i] = a.v[i] + (i < 4 ? a.v[5 + i] : 0u); sb[i] = b.v[i] + (i < 4 ? b.v[5 + i] : 0u); }
    int64_t zl[9], zh[9], zm[9];
#pragma unroll
    for (int k = 0; k < 9; k++) {
        int64_t l = 0, h = 0, m = 0;
        const int lo = k < 5 ? 0 : k - 4, hi = k < 5 ? k : 4;
#pragma unroll
        for (int i = lo; i <= hi; i++) {
            l += (int64_t)((uint64_t)a.v[i] * b.v[k - i]);
            if (i < 4 && k - i < 4) h += (int64_t)((uint64_t)a.v[5 + i] * b.v[5 + k - i]);
            m += (int64_t)((uint64_t)sa[i] * sb[k - i]);
        }
        zl[k] = l; zh[k] = h; zm[k] = m;
    }
    int64_t col[19];
#pragma unroll
    for (int k = 0; k < 19; k++) col[k] = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) { col[k] += zl[k]; col[k + 5] += zm[k] - zl[k] - zh[k]; col[k + 10] += zh[k]; }
    uint32_t mm[9];
    int64_t acc = 0;
    F29 r;
#pragma unroll
    for (int k = 0; k < 17; k++) {
        acc += col[k];
        if (k < 9) {
#pragma unroll
            for (int i = 0; i < k; i++) acc += (int64_t)((uint64_t)mm[i] * F::p_limb(k - i));
            mm[k] = ((uint32_t)acc * F::NP29) & WS_M29;
            acc += (int64_t)((uint64_t)mm[k] * F::p_limb(0));
        } else {
#pragma unroll
            for (int i = k - 8; i <= 8; i++) acc += (int64_t)((uint64_t)mm[i] * F::p_limb(k - i));
            r.v[k - 9] = (uint32_t)acc & WS_M29;
        }
        acc >>= 29;
    }
    r.v[8] = (uint32_t)acc;
    return r;
}

#ifndef WSNARK_EMUL
__global__ void k_shipped(const F29* in, F29* out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    out[t] = Field29I<Fq29Params>::mul(in[2 * t], in[2 * t + 1]);
}
__global__ void k_karatsuba(const F29* in, F29* out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    out[t] = mont_mul29_karatsuba<Fq29Params>(in[2 * t], in[2 * t + 1]);
}
#endif

}  // namespace wsnark

#ifdef WSNARK_EMUL
int main() {
    using namespace wsnark;
    typedef Field29<Fq29Params> F;
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    int bad = 0;
    for (int it = 0; it < 20000; it++) {
        F29 a, b;
        for (int i = 0; i < 9; i++) { a.v[i] = (uint32_t)rnd() & WS_M29; b.v[i] = (uint32_t)rnd() & WS_M29; }
        a.v[8] &= 0x3fffff; b.v[8] &= 0x3fffff;                       // < 2^254: below 2p
        if (it < 64) for (int i = 0; i < 8; i++) { a.v[i] = (it & 1) ? WS_M29 : 0; b.v[i] = (it & 2) ? WS_M29 : 0; }
        const F29 x = F::canonical(F::cond_sub_2p(F::mul(a, b))), y = F::canonical(F::cond_sub_2p(mont_mul29_karatsuba<Fq29Params>(a, b)));
        for (int i = 0; i < 9; i++) if (x.v[i] != y.v[i]) { bad++; break; }
    }
    printf("karatsuba 5+4 product vs shipped: %s (20000 operand pairs incl. extreme limbs)\n", bad ? "MISMATCH" : "identical mod p");
    return bad != 0;
}
#endif
