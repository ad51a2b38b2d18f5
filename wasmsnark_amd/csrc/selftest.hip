// selftest.hip -- one lane per vector through the DEVICE field and curve arithmetic (test hooks).
//
// The reference tests its primitives directly: f1m_* on the edge grid {0, 1, 2, q-1, q-2, (q-1)/2 +- k, ...}
// (/root/reference test/f1.js:296-400) and g1m_* / g2m_* on P+P, P-P, infinity and same-point-different-z cases
// (test/bn128.js:84-185).  In this build those primitives are __device__ functions (field29.h lazy [0,2p) radix-2^29
// arithmetic with sub_weak / neg_weak / fused multi-product reductions, fp2.h, curve.h XYZZ formulas) that are
// otherwise only reached through whole MSM / NTT kernels, where a random input hits an edge operand with
// probability ~0.  wsnark_selftest_field / wsnark_selftest_curve (include/wsnark.h) expose them so that
// tests/test_gpu_primitives.py can run the reference's own vectors (tests/golden/fields.json, groups.json) on the GPU.
// Nothing in the product path calls this file.
#include <string.h>

#include <vector>

#include "../../include/wsnark.h"
#include "internal.h"

namespace wsnark {

template <class Packed>
__host__ __device__ inline Packed st_flag(bool v) {
    Packed r;
    uint64_t* w = reinterpret_cast<uint64_t*>(&r);
    for (size_t i = 0; i < sizeof(Packed) / 8; i++) w[i] = 0;
    w[0] = v ? 1 : 0;
    return r;
}

// ---- base fields: F = Field29<P> (internal domain, lazy) or Field<P> (reference format itself) ----
template <class F>
__host__ __device__ inline Fe st_base_op(int op, const Fe& ap, const Fe& bp, const Fe& c_rr, const Fe& c_one_plain, bool* ok) {
    typedef typename F::El El;
    const El a = F::to_internal(ap), b = F::to_internal(bp);
    *ok = true;
    switch (op) {
        case WSNARK_ST_MUL: return F::from_internal(F::mul(a, b));
        case WSNARK_ST_SQR: return F::from_internal(F::sqr(a));
        case WSNARK_ST_ADD: return F::from_internal(F::add(a, b));
        case WSNARK_ST_SUB: return F::from_internal(F::sub(a, b));
        case WSNARK_ST_NEG: return F::from_internal(F::neg(a));
        case WSNARK_ST_TOMONT: return F::from_internal(F::mul(a, F::to_internal(c_rr)));
        case WSNARK_ST_FROMMONT: return F::from_internal(F::mul(a, F::to_internal(c_one_plain)));
        case WSNARK_ST_SUB_WEAK: return F::from_internal(F::sub_weak(a, b));
        case WSNARK_ST_ADD_LAZY_MUL: return F::from_internal(F::mul(F::add_lazy(a, b), b));
        case WSNARK_ST_MULSUB2: return F::from_internal(F::mulsub2(F::sub_weak(a, b), a, b, a));
        case WSNARK_ST_SQR_WEAK: return F::from_internal(F::sqr(F::sub_weak(a, b)));
        case WSNARK_ST_MUL_WEAK_A: return F::from_internal(F::mul(F::sub_weak(a, b), b));
        case WSNARK_ST_MULSUB2_WEAK_B: return F::from_internal(F::mulsub2(a, F::sub_weak(a, b), b, a));
        case WSNARK_ST_EQ: return st_flag<Fe>(F::is_zero(F::sub(a, b)));
        case WSNARK_ST_EQ_WEAK: return st_flag<Fe>(F::is_zero_weak(F::sub_weak(a, b)));
        // the DEVICE inversion: a^(p-2) on the field the kernels use (msm_table_norm_kernel: one per point and group of table rows);
        // the reference's f1m_inverse (src/build_f1m.js:772-782, extended Euclid) gives the same unique value
        case WSNARK_ST_INVERSE: return F::from_internal(F::inv(a));
        default: break;
    }
    if constexpr (F::kHasMul2Add) {
        if (op == WSNARK_ST_NEG_WEAK_MUL) return F::from_internal(F::mul(F::neg_weak(a), b));
        if (op == WSNARK_ST_MUL2ADD) return F::from_internal(F::mul2add(a, a, b, b));
    } else {
        if (op == WSNARK_ST_NEG_WEAK_MUL) return F::from_internal(F::mul(F::neg(a), b));
        if (op == WSNARK_ST_MUL2ADD) return F::from_internal(F::add(F::mul(a, a), F::mul(b, b)));
    }
    *ok = false;
    return ap;
}

// ---- quadratic extension: F2 = Fp2T<Fq29> or Fp2T<Fq> ----
template <class F2>
__host__ __device__ inline typename F2::Packed st_ext_op(int op, const typename F2::Packed& ap, const typename F2::Packed& bp, bool* ok) {
    typedef typename F2::El El;
    typedef typename F2::Packed Pk;
    const El a = F2::to_internal(ap), b = F2::to_internal(bp);
    *ok = true;
    switch (op) {
        case WSNARK_ST_MUL: return F2::from_internal(F2::mul(a, b));
        case WSNARK_ST_SQR: return F2::from_internal(F2::sqr(a));
        case WSNARK_ST_ADD: return F2::from_internal(F2::add(a, b));
        case WSNARK_ST_SUB: return F2::from_internal(F2::sub(a, b));
        case WSNARK_ST_NEG: return F2::from_internal(F2::neg(a));
        case WSNARK_ST_SUB_WEAK: return F2::from_internal(F2::sub_weak(a, b));
        case WSNARK_ST_MULSUB2: return F2::from_internal(F2::mulsub2(F2::sub_weak(a, b), a, b, a));
        case WSNARK_ST_SQR_WEAK: return F2::from_internal(F2::sqr(F2::sub_weak(a, b)));
        case WSNARK_ST_MUL_WEAK_A: return F2::from_internal(F2::mul(F2::sub_weak(a, b), b));
        case WSNARK_ST_MULSUB2_WEAK_B: return F2::from_internal(F2::mulsub2(a, F2::sub_weak(a, b), b, a));
        case WSNARK_ST_EQ: return st_flag<Pk>(F2::is_zero(F2::sub(a, b)));
        case WSNARK_ST_EQ_WEAK: return st_flag<Pk>(F2::is_zero_weak(F2::sub_weak(a, b)));
        default: break;
    }
    *ok = false;
    return ap;
}

template <class F>
__global__ __launch_bounds__(64) void st_base_kernel(int op, const Fe* __restrict__ a, const Fe* __restrict__ b, Fe* __restrict__ out,
                                                       uint64_t n, Fe c_rr, Fe c_one_plain, int* __restrict__ bad) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool ok;
    out[i] = st_base_op<F>(op, a[i], b[i], c_rr, c_one_plain, &ok);
    if (!ok) *bad = 1;
}
template <class F2>
__global__ __launch_bounds__(64) void st_ext_kernel(int op, const typename F2::Packed* __restrict__ a, const typename F2::Packed* __restrict__ b,
                                                      typename F2::Packed* __restrict__ out, uint64_t n, int* __restrict__ bad) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool ok;
    out[i] = st_ext_op<F2>(op, a[i], b[i], &ok);
    if (!ok) *bad = 1;
}

// three device buffers (a, b, out) + the "unsupported op" flag around one launch
struct StBufs {
    DevBuf a, b, out, bad;
    int up(const uint8_t* ha, const uint8_t* hb, size_t in_bytes, size_t out_bytes, hipStream_t s) {
        WS_HIP_CHECK(a.alloc(in_bytes));
        WS_HIP_CHECK(b.alloc(in_bytes));
        WS_HIP_CHECK(out.alloc(out_bytes));
        WS_HIP_CHECK(bad.alloc(4));
        WS_HIP_CHECK(hipMemcpyAsync(a.p, ha, in_bytes, hipMemcpyHostToDevice, s));
        WS_HIP_CHECK(hipMemcpyAsync(b.p, hb, in_bytes, hipMemcpyHostToDevice, s));
        WS_HIP_CHECK(hipMemsetAsync(bad.p, 0, 4, s));
        return WS_OK;
    }
    int down(uint8_t* hout, size_t out_bytes, hipStream_t s) {
        int flag = 0;
        WS_HIP_CHECK(hipGetLastError());
        WS_HIP_CHECK(hipMemcpyAsync(hout, out.p, out_bytes, hipMemcpyDeviceToHost, s));
        WS_HIP_CHECK(hipMemcpyAsync(&flag, bad.p, 4, hipMemcpyDeviceToHost, s));
        WS_HIP_CHECK(hipStreamSynchronize(s));
        if (flag) { set_last_error("selftest: operation not defined for this field / implementation"); return WS_ERR_ARG; }
        return WS_OK;
    }
};

template <class F, class P>
static int st_base_dev(int op, const uint8_t* a, const uint8_t* b, uint8_t* out, uint64_t n, hipStream_t s) {
    StBufs B;
    int rc = B.up(a, b, n * 32, n * 32, s);
    if (rc) return rc;
    const Fe rr = {{P::RR0, P::RR1, P::RR2, P::RR3}}, one = {{1, 0, 0, 0}};
    hipLaunchKernelGGL(st_base_kernel<F>, dim3(ceil_div_u64(n, 64)), dim3(64), 0, s, op, B.a.as<Fe>(), B.b.as<Fe>(), B.out.as<Fe>(), n,
                       rr, one, B.bad.as<int>());
    return B.down(out, n * 32, s);
}
template <class P>
static int st_base_host(int op, const uint8_t* a, const uint8_t* b, uint8_t* out, uint64_t n) {
    typedef Field<P> F;
    const Fe rr = {{P::RR0, P::RR1, P::RR2, P::RR3}}, one = {{1, 0, 0, 0}};
    for (uint64_t i = 0; i < n; i++) {
        Fe x, y, r;
        memcpy(&x, a + i * 32, 32);
        memcpy(&y, b + i * 32, 32);
        bool ok = true;
        if (op == WSNARK_ST_INVERSE) r = F::inv(x);                   // a^(p-2): build_f1m.js:772-782 gives the same unique value
        else r = st_base_op<F>(op, x, y, rr, one, &ok);
        if (!ok) { set_last_error("selftest: operation not defined for this field / implementation"); return WS_ERR_ARG; }
        memcpy(out + i * 32, &r, 32);
    }
    return WS_OK;
}

int selftest_field(int which, int impl, int op, const uint8_t* a, const uint8_t* b, uint8_t* out, uint64_t n) {
    Context* X = ctx();
    if (!X) return WS_ERR_NOINIT;
    if (n == 0) return WS_OK;
    if (n > (1u << 20)) return WS_ERR_SIZE;
    hipStream_t s = X->stream;
    if (impl == 2) {
        if (which == 0) return st_base_host<FqParams>(op, a, b, out, n);
        if (which == 1) return st_base_host<FrParams>(op, a, b, out, n);
        if (which == 2) {
            for (uint64_t i = 0; i < n; i++) {
                Fe2 x, y, r;
                memcpy(&x, a + i * 64, 64);
                memcpy(&y, b + i * 64, 64);
                bool ok = true;
                if (op == WSNARK_ST_INVERSE) r = Fq2::inv(x);         // build_f2m.js:353-383
                else r = st_ext_op<Fq2>(op, x, y, &ok);
                if (!ok) { set_last_error("selftest: operation not defined for this field / implementation"); return WS_ERR_ARG; }
                memcpy(out + i * 64, &r, 64);
            }
            return WS_OK;
        }
        return WS_ERR_ARG;
    }
    if (op == WSNARK_ST_INVERSE && which == 2) { set_last_error("selftest: the extension field is inverted on the host only (impl 2)"); return WS_ERR_ARG; }
    if (which == 0 && impl == 0) return st_base_dev<Fq29, FqParams>(op, a, b, out, n, s);
    if (which == 0 && impl == 1) return st_base_dev<Fq, FqParams>(op, a, b, out, n, s);
    if (which == 1 && impl == 0) return st_base_dev<Fr29, FrParams>(op, a, b, out, n, s);
    if (which == 1 && impl == 1) return st_base_dev<Fr, FrParams>(op, a, b, out, n, s);
    if (which == 2 && (impl == 0 || impl == 1)) {
        StBufs B;
        int rc = B.up(a, b, n * 64, n * 64, s);
        if (rc) return rc;
        if (impl == 0)
            hipLaunchKernelGGL(st_ext_kernel<Fp2T<Fq29>>, dim3(ceil_div_u64(n, 64)), dim3(64), 0, s, op, B.a.as<Fe2>(), B.b.as<Fe2>(),
                               B.out.as<Fe2>(), n, B.bad.as<int>());
        else
            hipLaunchKernelGGL(st_ext_kernel<Fq2>, dim3(ceil_div_u64(n, 64)), dim3(64), 0, s, op, B.a.as<Fe2>(), B.b.as<Fe2>(),
                               B.out.as<Fe2>(), n, B.bad.as<int>());
        return B.down(out, n * 64, s);
    }
    return WS_ERR_ARG;
}

// ---- curves ----
// p, q: Jacobian-Montgomery triples in the reference format; result: XYZZ in the reference format (the form in
// which every kernel hands its sums to the host), normalised by the host like an MSM result.
template <class C>
__host__ __device__ inline typename C::PtP st_curve_op(int op, const typename C::Field::Packed* pj, const typename C::Field::Packed* qj, bool* ok) {
    typedef typename C::Field F;
    typedef typename C::El El;
    typedef typename C::Pt Pt;
    auto from_jac = [](const typename F::Packed* j) -> Pt {     // (x, y, z) -> (X, Y, z^2, z^3)
        const El z = F::to_internal(j[2]);
        if (F::is_zero(z)) return C::infinity();
        const El zz = F::sqr(z);
        return Pt{F::to_internal(j[0]), F::to_internal(j[1]), zz, F::mul(zz, z)};
    };
    auto affine_of = [](const typename F::Packed* j) -> typename C::Aff {   // z == 1 expected; z == 0 -> x = 0 (infinity)
        const El z = F::to_internal(j[2]);
        if (F::is_zero(z)) return typename C::Aff{F::zero(), F::one()};
        return typename C::Aff{F::to_internal(j[0]), F::to_internal(j[1])};
    };
    *ok = true;
    Pt r = C::infinity();
    switch (op) {
        case 0: r = C::add(from_jac(pj), from_jac(qj)); break;
        case 1: r = C::dbl(from_jac(pj)); break;
        case 2: r = C::neg(from_jac(pj)); break;
        case 3: r = from_jac(pj); break;
        case 4: r = from_jac(pj); C::madd(r, affine_of(qj), false); break;
        case 5: r = from_jac(pj); C::madd(r, affine_of(qj), true); break;
        case 6: r = from_jac(pj); C::madd_wide(r, affine_of(qj), false); C::madd_wide(r, affine_of(qj), false); C::narrow_x(r); break;
        case 7: r = from_jac(pj); C::madd_wide(r, affine_of(qj), false); C::madd_wide(r, affine_of(qj), true); C::narrow_x(r); break;
        case 8: {   // timesScalar (src/build_timesscalar.js:20-80): the second operand's bytes are a little-endian scalar of q[64] = 32 or 64 bytes
            const uint8_t* sc = reinterpret_cast<const uint8_t*>(qj);
            const int nb = sc[64] == 64 ? 64 : 32;
            r = C::mul_bytes(from_jac(pj), sc, nb);
            break;
        }
        default: *ok = false; break;
    }
    return C::pt_from_internal(r);
}
template <class C>
__global__ __launch_bounds__(64) void st_curve_kernel(int op, const typename C::Field::Packed* __restrict__ p,
                                                        const typename C::Field::Packed* __restrict__ q,
                                                        typename C::PtP* __restrict__ out, uint64_t n, int* __restrict__ bad) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool ok;
    out[i] = st_curve_op<C>(op, p + 3 * i, q + 3 * i, &ok);
    if (!ok) *bad = 1;
}

// the paired G2 curve (fp2.h: Fp2PairT): two lanes per vector, lane p of a pair reads / writes component p of every coordinate
__global__ __launch_bounds__(64) void st_curve_pair_kernel(int op, const Fe* __restrict__ p, const Fe* __restrict__ q, Fe* __restrict__ out,
                                                             uint64_t n, int* __restrict__ bad) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t i = t >> 1;
    const uint32_t h = (uint32_t)(t & 1);
    if (i >= n) return;
    const Fe pj[3] = {p[6 * i + h], p[6 * i + 2 + h], p[6 * i + 4 + h]};
    const Fe qj[3] = {q[6 * i + h], q[6 * i + 2 + h], q[6 * i + 4 + h]};
    bool ok;
    const G2P29::PtP r = st_curve_op<G2P29>(op, pj, qj, &ok);
    out[8 * i + h] = r.x; out[8 * i + 2 + h] = r.y; out[8 * i + 4 + h] = r.zz; out[8 * i + 6 + h] = r.zzz;
    if (!ok) *bad = 1;
}
// the paired G1 curve (curve_pair.h: CurvePairG1): two lanes per vector, lo = (X, ZZ), hi = (Y, ZZZ); ops 0 (add), 1 (double), 3 (copy)
__global__ __launch_bounds__(64) void st_curve_pair_g1_kernel(int op, const Fe* __restrict__ p, const Fe* __restrict__ q, Fe* __restrict__ out,
                                                                uint64_t n, int* __restrict__ bad) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t i = t >> 1;
    const uint32_t h = (uint32_t)(t & 1);
    if (i >= n) return;
    typedef G1R29I C;
    typedef C::Field F;
    auto from_jac = [](const Fe* j) -> C::Pt {                    // both lanes build the whole point, then keep their half
        const F::El z = F::to_internal(j[2]);
        if (F::is_zero(z)) return C::infinity();
        const F::El zz = F::sqr(z);
        return C::Pt{F::to_internal(j[0]), F::to_internal(j[1]), zz, F::mul(zz, z)};
    };
    const G1P29::Pt a = G1P29::split(from_jac(p + 3 * i)), b = G1P29::split(from_jac(q + 3 * i));
    G1P29::Pt r = G1P29::infinity();
    if (op == 0) r = G1P29::add(a, b);
    else if (op == 1) r = G1P29::dbl(a);
    else if (op == 3) r = a;
    else *bad = 1;
    out[4 * i + h] = F::from_internal(r.a);                       // x | y
    out[4 * i + 2 + h] = F::from_internal(r.b);                   // zz | zzz
}
static int st_curve_pair_g1_dev(int op, const uint8_t* p, const uint8_t* q, uint8_t* out, uint64_t n, hipStream_t s) {
    StBufs B;
    int rc = B.up(p, q, n * 96, n * 128, s);
    if (rc) return rc;
    hipLaunchKernelGGL(st_curve_pair_g1_kernel, dim3(ceil_div_u64(2 * n, 64)), dim3(64), 0, s, op, B.a.as<Fe>(), B.b.as<Fe>(), B.out.as<Fe>(), n, B.bad.as<int>());
    std::vector<G1::Pt> host(n);
    if ((rc = B.down(reinterpret_cast<uint8_t*>(host.data()), n * sizeof(G1::Pt), s))) return rc;
    for (uint64_t i = 0; i < n; i++) {
        auto j = G1::to_affine_jac(host[i]);
        memcpy(out + i * sizeof j, &j, sizeof j);
    }
    return WS_OK;
}
// G2 on FOUR lanes (curve_pair.h: CurvePairG1<Fp2PairT, 2>): lane (h2, h1) holds component h1 of (X, ZZ) or of (Y, ZZZ); ops 0, 1, 3
__global__ __launch_bounds__(64) void st_curve_quad_g2_kernel(int op, const Fe* __restrict__ p, const Fe* __restrict__ q, Fe* __restrict__ out,
                                                                uint64_t n, int* __restrict__ bad) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t i = t >> 2;
    const uint32_t h1 = (uint32_t)(t & 1), h2 = (uint32_t)((t >> 1) & 1);
    if (i >= n) return;
    typedef G2P29 C;
    typedef C::Field F;
    auto from_jac = [](const Fe* j) -> C::Pt {                    // (this lane's component of every coordinate)
        const F::El z = F::to_internal(j[2]);
        if (F::is_zero(z)) return C::infinity();
        const F::El zz = F::sqr(z);
        return C::Pt{F::to_internal(j[0]), F::to_internal(j[1]), zz, F::mul(zz, z)};
    };
    const Fe pj[3] = {p[6 * i + h1], p[6 * i + 2 + h1], p[6 * i + 4 + h1]};
    const Fe qj[3] = {q[6 * i + h1], q[6 * i + 2 + h1], q[6 * i + 4 + h1]};
    const G2Q29::Pt a = G2Q29::split(from_jac(pj)), b = G2Q29::split(from_jac(qj));
    G2Q29::Pt r = G2Q29::infinity();
    if (op == 0) r = G2Q29::add(a, b);
    else if (op == 1) r = G2Q29::dbl(a);
    else if (op == 3) r = a;
    else *bad = 1;
    out[8 * i + h1 + 2 * h2] = F::from_internal(r.a);             // x.c | y.c
    out[8 * i + 4 + h1 + 2 * h2] = F::from_internal(r.b);         // zz.c | zzz.c
}
static int st_curve_quad_g2_dev(int op, const uint8_t* p, const uint8_t* q, uint8_t* out, uint64_t n, hipStream_t s) {
    StBufs B;
    int rc = B.up(p, q, n * 192, n * 256, s);
    if (rc) return rc;
    hipLaunchKernelGGL(st_curve_quad_g2_kernel, dim3(ceil_div_u64(4 * n, 64)), dim3(64), 0, s, op, B.a.as<Fe>(), B.b.as<Fe>(), B.out.as<Fe>(), n, B.bad.as<int>());
    std::vector<G2::Pt> host(n);
    if ((rc = B.down(reinterpret_cast<uint8_t*>(host.data()), n * sizeof(G2::Pt), s))) return rc;
    for (uint64_t i = 0; i < n; i++) {
        auto j = G2::to_affine_jac(host[i]);
        memcpy(out + i * sizeof j, &j, sizeof j);
    }
    return WS_OK;
}
static int st_curve_pair_dev(int op, const uint8_t* p, const uint8_t* q, uint8_t* out, uint64_t n, hipStream_t s) {
    StBufs B;
    int rc = B.up(p, q, n * 192, n * 256, s);
    if (rc) return rc;
    hipLaunchKernelGGL(st_curve_pair_kernel, dim3(ceil_div_u64(2 * n, 64)), dim3(64), 0, s, op, B.a.as<Fe>(), B.b.as<Fe>(), B.out.as<Fe>(), n, B.bad.as<int>());
    std::vector<G2::Pt> host(n);
    if ((rc = B.down(reinterpret_cast<uint8_t*>(host.data()), n * sizeof(G2::Pt), s))) return rc;
    for (uint64_t i = 0; i < n; i++) {
        auto j = G2::to_affine_jac(host[i]);
        memcpy(out + i * sizeof j, &j, sizeof j);
    }
    return WS_OK;
}

template <class C, class H>
static int st_curve_dev(int op, const uint8_t* p, const uint8_t* q, uint8_t* out, uint64_t n, hipStream_t s) {
    typedef typename C::Field::Packed Pk;
    static_assert(sizeof(typename C::PtP) == sizeof(typename H::Pt), "layouts");
    StBufs B;
    int rc = B.up(p, q, n * 3 * sizeof(Pk), n * sizeof(typename C::PtP), s);
    if (rc) return rc;
    hipLaunchKernelGGL(st_curve_kernel<C>, dim3(ceil_div_u64(n, 64)), dim3(64), 0, s, op, B.a.as<Pk>(), B.b.as<Pk>(),
                       B.out.as<typename C::PtP>(), n, B.bad.as<int>());
    std::vector<typename H::Pt> host(n);
    if ((rc = B.down(reinterpret_cast<uint8_t*>(host.data()), n * sizeof(typename H::Pt), s))) return rc;
    for (uint64_t i = 0; i < n; i++) {
        auto j = H::to_affine_jac(host[i]);
        memcpy(out + i * sizeof j, &j, sizeof j);
    }
    return WS_OK;
}
template <class H>
static int st_curve_host(int op, const uint8_t* p, const uint8_t* q, uint8_t* out, uint64_t n) {
    typedef typename H::Field::Packed Pk;
    for (uint64_t i = 0; i < n; i++) {
        Pk pj[3], qj[3];
        memcpy(pj, p + i * sizeof pj, sizeof pj);
        memcpy(qj, q + i * sizeof qj, sizeof qj);
        bool ok;
        typename H::PtP r = st_curve_op<H>(op, pj, qj, &ok);
        if (!ok) return WS_ERR_ARG;
        typename H::Pt rp;
        memcpy(&rp, &r, sizeof rp);
        auto j = H::to_affine_jac(rp);
        memcpy(out + i * sizeof j, &j, sizeof j);
    }
    return WS_OK;
}

int selftest_curve(int g, int impl, int op, const uint8_t* p, const uint8_t* q, uint8_t* out, uint64_t n) {
    Context* X = ctx();
    if (!X) return WS_ERR_NOINIT;
    if (n == 0) return WS_OK;
    if (n > (1u << 20) || op < 0 || op > 8 || (op == 8 && impl == 4)) return WS_ERR_ARG;
    if (g == 1 && impl == 5) return (op == 0 || op == 1 || op == 3) ? st_curve_pair_g1_dev(op, p, q, out, n, X->stream) : (int)WS_ERR_ARG;
    if (g == 2 && impl == 6) return (op == 0 || op == 1 || op == 3) ? st_curve_quad_g2_dev(op, p, q, out, n, X->stream) : (int)WS_ERR_ARG;
    hipStream_t s = X->stream;
    if (g == 1) {
        if (impl == 0) return st_curve_dev<G1R29, G1>(op, p, q, out, n, s);
        if (impl == 1) return st_curve_dev<G1, G1>(op, p, q, out, n, s);
        if (impl == 2) return st_curve_host<G1>(op, p, q, out, n);
        if (impl == 3) return st_curve_dev<G1R29I, G1>(op, p, q, out, n, s);
    } else if (g == 2) {
        if (impl == 0) return st_curve_dev<G2R29, G2>(op, p, q, out, n, s);
        if (impl == 1) return st_curve_dev<G2, G2>(op, p, q, out, n, s);
        if (impl == 2) return st_curve_host<G2>(op, p, q, out, n);
        if (impl == 4) return st_curve_pair_dev(op, p, q, out, n, s);
    }
    return WS_ERR_ARG;
}

// ---- peak probes (bench.py: the integer roofline's peak, re-measured on the box the run is on) ----
// probe 0: every lane runs one dependent chain of the PRODUCT'S OWN radix-2^29 Montgomery product (Fq29::mul, the
//          function the accumulation kernels call): what a kernel made of nothing but products reaches = the "multiplier
//          peak" in Gmodmul/s.  probe 1: the same with the inlined body (Fq29I).  probe 2: eight independent chains of
//          v_mad_u64_u32 per lane: the raw multiply-add issue rate in Gmad/s.
template <class F>
__global__ __launch_bounds__(256) void probe_modmul_kernel(const Fe* __restrict__ in, Fe* __restrict__ out, int iters) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    typename F::El x = F::to_internal(in[t]);
    const typename F::El y = x;
    for (int i = 0; i < iters; i++) x = F::mul(x, y);
    out[t] = F::from_internal(x);
}
// probes 301..308 (G1) / 401..408 (G2): the accumulation loop's own mixed addition (curve.h madd_wide, the inlined radix-2^29 field) on
// points held in REGISTERS -- no gather, no task list -- at 1..8 wavefronts per SIMD: what msm_accumulate would reach if memory cost
// nothing.  G lane-additions per second.
template <class C>
__global__ __launch_bounds__(256) void probe_madd_kernel(const typename C::AffP* __restrict__ in, typename C::PtP* __restrict__ out, int iters) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const typename C::Aff p0 = C::unpack_aff(in[0]), p1 = C::unpack_aff(in[1]);
    typename C::Pt acc = C::infinity();
    for (int i = 0; i < iters; i++) {
        C::madd_wide(acc, (i + t) & 1 ? p0 : p1, (t & 4) != 0);
    }
    C::narrow_x(acc);
    out[t] = C::pack_pt(acc);
}
template <class F>
__global__ __launch_bounds__(256) void probe_modmul2_kernel(const Fe* __restrict__ in, Fe* __restrict__ out, int iters) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    typename F::El x = F::to_internal(in[t]);
    const typename F::El y = x;
    typename F::El z = F::add(x, y);
    for (int i = 0; i < iters; i++) { x = F::mul(x, y); z = F::mul(z, y); }
    out[t] = F::from_internal(F::add(x, z));
}
__global__ __launch_bounds__(256) void probe_mad_kernel(uint64_t* __restrict__ out, uint32_t a, uint32_t b, int iters) {
    uint64_t c0 = threadIdx.x, c1 = c0 + 1, c2 = c0 + 2, c3 = c0 + 3, c4 = c0 + 4, c5 = c0 + 5, c6 = c0 + 6, c7 = c0 + 7;
    uint32_t x = a + threadIdx.x, y = b;
    for (int i = 0; i < iters; i++) {
        c0 = (uint64_t)x * y + c0; c1 = (uint64_t)x * y + c1; c2 = (uint64_t)x * y + c2; c3 = (uint64_t)x * y + c3;
        c4 = (uint64_t)x * y + c4; c5 = (uint64_t)x * y + c5; c6 = (uint64_t)x * y + c6; c7 = (uint64_t)x * y + c7;
        x = (uint32_t)c0; y = (uint32_t)(c7 >> 32) | 1;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;
}
// probes 3 / 4: traffic calibration for the FETCH_SIZE counter (the guide: "calibrate on a known byte count in your own
// access pattern").  3 = the accumulation kernel's pattern: every lane gathers 64-byte points (one AffP load, as
// msm_accumulate does) at pseudo-random indices of a 1 GiB table -- far beyond the 256 MiB Infinity Cache -- 2^25 gathers
// = 2 GiB of known algorithmic bytes (sector granularity 64 B: nothing to over-fetch).  4 = a streaming read of the same
// table, 16 B per lane and load: the pattern the guide's x2 correction was measured on.  Result: GB/s of known bytes.
__global__ __launch_bounds__(256) void probe_gather64_kernel(const Affine<Fq>* __restrict__ table, uint32_t mask, uint32_t per_lane,
                                                              uint64_t* __restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t idx = t * 2654435761u;
    uint64_t acc = 0;
    for (uint32_t k = 0; k < per_lane; k++) {
        idx = idx * 1664525u + 1013904223u;
        const Affine<Fq> p = table[idx & mask];
        acc += p.x.l[0] ^ p.x.l[3] ^ p.y.l[1] ^ p.y.l[2];
    }
    out[t] = acc;
}
struct alignas(16) Word16 { uint32_t x, y, z, w; };
__global__ __launch_bounds__(256) void probe_stream16_kernel(const Word16* __restrict__ src, uint64_t n16, uint64_t* __restrict__ out) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t acc = 0;
    for (uint64_t i = t; i < n16; i += stride) { const Word16 v = src[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    out[t] = acc;
}
static int traffic_probe(Context* C, int probe, double* gbs) {
    hipStream_t s = C->stream;
    const size_t table_bytes = (size_t)1 << 30;
    const uint32_t n_pts = (uint32_t)(table_bytes / 64);
    DevBuf table, out;
    WS_HIP_CHECK(table.alloc(table_bytes));
    WS_HIP_CHECK(hipMemsetAsync(table.p, 0x5a, table_bytes, s));
    const uint32_t lanes = 1u << 20, per_lane = 32;
    WS_HIP_CHECK(out.alloc((size_t)lanes * 8));
    hipEvent_t a = nullptr, b = nullptr;
    WS_HIP_CHECK(hipEventCreate(&a));
    WS_HIP_CHECK(hipEventCreate(&b));
    double best = 0;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(a, s);
        if (probe == 3) hipLaunchKernelGGL(probe_gather64_kernel, dim3(lanes / 256), dim3(256), 0, s, table.as<Affine<Fq>>(), n_pts - 1, per_lane, out.as<uint64_t>());
        else hipLaunchKernelGGL(probe_stream16_kernel, dim3(lanes / 256), dim3(256), 0, s, table.as<Word16>(), (uint64_t)(table_bytes / 16), out.as<uint64_t>());
        (void)hipEventRecord(b, s);
        if (hipEventSynchronize(b) != hipSuccess) break;
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        const double bytes = probe == 3 ? (double)lanes * per_lane * 64 : (double)table_bytes;
        if (ms > 0 && bytes / ms / 1e6 > best) best = bytes / ms / 1e6;
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    WS_HIP_CHECK(hipGetLastError());
    *gbs = best;
    return WS_OK;
}

// probe 5: one field inversion per lane and iteration (Fq29::inv: Fermat, x^(q-2) by square-and-multiply on the radix-2^29 product),
// every lane busy -- the cost a LANE-PARALLEL batch inversion adds per batch (the batch-affine analysis of DESIGN.md); G inversions/s
__global__ __launch_bounds__(256) void probe_inverse_kernel(const Fe* __restrict__ in, Fe* __restrict__ out, int iters) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    Fq29::El x = Fq29::to_internal(in[t]);
    for (int i = 0; i < iters; i++) x = Fq29::inv(x);
    out[t] = Fq29::from_internal(x);
}

// probes 6..14: VALU ISSUE COST PER INSTRUCTION CLASS (round 6).  One instruction class per kernel, written in inline assembly so
// that the compiler neither folds nor re-schedules it: eight independent chains per lane (no dependent-issue stall: the next
// instruction never reads the result of the one before), 64 instructions per loop trip, eight wavefronts per SIMD.  The result is
// the class's lane-operation rate in G/s (wave-instructions/s = that / 64).  bench.py prices a kernel's instruction stream as
// sum over classes of (count / rate) instead of a flat "4 cycles per VALU instruction" -- the classes are those of
// tools/isa_histogram.py.   6 v_add_u32 (bit32: add / and / or / 32-bit shifts)   7 v_lshrrev_b64 (shift64)   8 v_and_b32
// 9 v_mul_lo_u32 (mul32)   10 v_mad_u64_u32 (mad64)   11 v_add_co_u32 + v_addc_co_u32 pairs (add64c: two instructions counted)
// 12 v_mov_b32 (mov)   13 v_cndmask_b32 on a standing mask   14 v_mov_b32 quad_perm DPP (dpp)   16 v_cmp_lt_u32 into SGPR pairs
// 17 v_cmp + the v_cndmask that reads its mask, back to back (cmp_sel: how the conditional subtractions are written)
#ifndef WSNARK_EMUL
#define WS_ISSUE8(INS) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)
template <int CLASS>
__global__ __launch_bounds__(256) void probe_issue_kernel(uint64_t* __restrict__ out, uint32_t a, uint32_t b, int iters) {
    uint32_t x0 = a + threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    uint64_t w0 = x0, w1 = x1, w2 = x2, w3 = x3, w4 = x4, w5 = x5, w6 = x6, w7 = x7;
    uint32_t y = b | 1;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (CLASS == 6) {
#define WS_I(k) "v_add_u32 %" #k ", %" #k ", %8\n\t"
                asm volatile(WS_ISSUE8(WS_I) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y));
#undef WS_I
            } else if (CLASS == 7) {
#define WS_I(k) "v_lshrrev_b64 %" #k ", 1, %" #k "\n\t"
                asm volatile(WS_ISSUE8(WS_I) : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5), "+v"(w6), "+v"(w7));
#undef WS_I
            } else if (CLASS == 8) {
#define WS_I(k) "v_and_b32 %" #k ", %" #k ", %8\n\t"
                asm volatile(WS_ISSUE8(WS_I) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y));
#undef WS_I
            } else if (CLASS == 9) {
#define WS_I(k) "v_mul_lo_u32 %" #k ", %" #k ", %8\n\t"
                asm volatile(WS_ISSUE8(WS_I) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y));
#undef WS_I
            } else if (CLASS == 10) {
#define WS_I(k) "v_mad_u64_u32 %" #k ", vcc, %8, %9, %" #k "\n\t"
                asm volatile(WS_ISSUE8(WS_I) : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5), "+v"(w6), "+v"(w7) : "v"(y), "v"(x0) : "vcc");
#undef WS_I
            } else if (CLASS == 11) {
#define WS_I(k) "v_add_co_u32 %" #k ", vcc, %" #k ", %8\n\tv_addc_co_u32 %" #k ", vcc, %" #k ", %8, vcc\n\t"
                asm volatile(WS_I(0) WS_I(1) WS_I(2) WS_I(3) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y) : "vcc");
#undef WS_I
            } else if (CLASS == 12) {
                // moves between two register sets: a chain of length one (the destination of one is never the source of the next)
                asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7\n\t"
                             "v_mov_b32 %4, %0\n\tv_mov_b32 %5, %1\n\tv_mov_b32 %6, %2\n\tv_mov_b32 %7, %3\n\t"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            } else if (CLASS == 13) {
                // selects on a mask that is already there (vcc written once per trip): v_cndmask_b32 alone
#define WS_I(k) "v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n\t"
                asm volatile(WS_ISSUE8(WS_I) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y) : "vcc");
#undef WS_I
            } else if (CLASS == 16) {
                // compares alone: eight independent v_cmp into four SGPR pairs (no instruction reads a mask the one before wrote)
                uint64_t m0, m1, m2, m3;
                asm volatile("v_cmp_lt_u32 %0, %4, %12\n\tv_cmp_lt_u32 %1, %5, %12\n\tv_cmp_lt_u32 %2, %6, %12\n\tv_cmp_lt_u32 %3, %7, %12\n\t"
                             "v_cmp_lt_u32 %0, %8, %12\n\tv_cmp_lt_u32 %1, %9, %12\n\tv_cmp_lt_u32 %2, %10, %12\n\tv_cmp_lt_u32 %3, %11, %12\n\t"
                             : "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3)
                             : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7), "v"(y));
                x0 += (uint32_t)__builtin_popcountll(m0 ^ m1 ^ m2 ^ m3) & (u == 9);      // (keeps the masks alive; never taken: u < 8)
            } else if (CLASS == 18) {
                // selects on a standing mask, each followed by an independent v_add_u32 (does other work hide the select's cost?)
                asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n\tv_add_u32 %4, %4, %8\n\tv_cndmask_b32 %1, %1, %8, vcc\n\tv_add_u32 %5, %5, %8\n\t"
                             "v_cndmask_b32 %2, %2, %8, vcc\n\tv_add_u32 %6, %6, %8\n\tv_cndmask_b32 %3, %3, %8, vcc\n\tv_add_u32 %7, %7, %8\n\t"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y) : "vcc");
            } else if (CLASS == 19) {
                // the select in its three-operand encoding with the mask in an SGPR pair other than vcc
                uint64_t m = 0x5555555555555555ull;
                asm volatile("v_cndmask_b32_e64 %0, %0, %8, %9\n\tv_cndmask_b32_e64 %1, %1, %8, %9\n\tv_cndmask_b32_e64 %2, %2, %8, %9\n\tv_cndmask_b32_e64 %3, %3, %8, %9\n\t"
                             "v_cndmask_b32_e64 %4, %4, %8, %9\n\tv_cndmask_b32_e64 %5, %5, %8, %9\n\tv_cndmask_b32_e64 %6, %6, %8, %9\n\tv_cndmask_b32_e64 %7, %7, %8, %9\n\t"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y), "s"(m));
            } else if (CLASS == 20) {
                // the same choice by arithmetic: x ^= (x ^ y) & mask with a per-lane 0 / ~0 mask register (two instructions per select)
                asm volatile("v_xor_b32 %4, %0, %8\n\tv_and_b32 %4, %4, %9\n\tv_xor_b32 %0, %0, %4\n\tv_xor_b32 %5, %1, %8\n\tv_and_b32 %5, %5, %9\n\tv_xor_b32 %1, %1, %5\n\t"
                             "v_bfi_b32 %2, %9, %8, %2\n\tv_bfi_b32 %3, %9, %8, %3\n\t"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y), "v"(x6));
            } else if (CLASS == 21) {
#define WS_I(k) "v_add3_u32 %" #k ", %" #k ", %8, %8\n\t"
                asm volatile(WS_ISSUE8(WS_I) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y));
#undef WS_I
            } else if (CLASS == 22) {
#define WS_I(k) "v_bfi_b32 %" #k ", %8, %9, %" #k "\n\t"
                asm volatile(WS_ISSUE8(WS_I) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y), "v"((uint32_t)w0));
#undef WS_I
            } else if (CLASS == 23) {
                // selects on a standing mask in runs of TWO between independent additions
                asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n\tv_cndmask_b32 %1, %1, %8, vcc\n\tv_add_u32 %4, %4, %8\n\tv_add_u32 %5, %5, %8\n\t"
                             "v_cndmask_b32 %2, %2, %8, vcc\n\tv_cndmask_b32 %3, %3, %8, vcc\n\tv_add_u32 %6, %6, %8\n\tv_add_u32 %7, %7, %8\n\t"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y) : "vcc");
            } else if (CLASS == 24) {
#define WS_I(k) "v_alignbit_b32 %" #k ", %8, %" #k ", 29\n\t"
                asm volatile(WS_ISSUE8(WS_I) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y));
#undef WS_I
            } else if (CLASS == 25) {
#define WS_I(k) "v_and_b32 %" #k ", 0x1fffffff, %" #k "\n\t"
                asm volatile(WS_ISSUE8(WS_I) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
#undef WS_I
            } else if (CLASS == 26) {
#define WS_I(k) "v_add_u32_e64 %" #k ", %" #k ", %8\n\t"
                asm volatile(WS_ISSUE8(WS_I) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y));
#undef WS_I
            } else if (CLASS == 27) {
#define WS_I(k) "v_and_b32 %" #k ", %8, %" #k "\n\t"
                asm volatile(WS_ISSUE8(WS_I) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(b | 0x10000001u));
#undef WS_I
            } else if (CLASS == 28) {
#define WS_I(k) "v_lshrrev_b32 %" #k ", 29, %" #k "\n\t"
                asm volatile(WS_ISSUE8(WS_I) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
#undef WS_I
            } else if (CLASS == 17) {
                // a compare and the select that reads its mask, back to back (the pattern of a conditional subtraction), four pairs
                asm volatile("v_cmp_lt_u32 vcc, %0, %8\n\tv_cndmask_b32 %1, %1, %8, vcc\n\tv_cmp_lt_u32 vcc, %2, %8\n\tv_cndmask_b32 %3, %3, %8, vcc\n\t"
                             "v_cmp_lt_u32 vcc, %4, %8\n\tv_cndmask_b32 %5, %5, %8, vcc\n\tv_cmp_lt_u32 vcc, %6, %8\n\tv_cndmask_b32 %7, %7, %8, vcc\n\t"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y) : "vcc");
            } else {
                asm volatile("v_mov_b32_dpp %0, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "v_mov_b32_dpp %2, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "v_mov_b32_dpp %4, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %5, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "v_mov_b32_dpp %6, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %7, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7) + (w0 ^ w1 ^ w2 ^ w3 ^ w4 ^ w5 ^ w6 ^ w7);
}
static void launch_issue_probe(int cls, uint32_t blocks, uint32_t threads, hipStream_t s, uint64_t* out, int iters) {
    switch (cls) {
#define WS_CASE(c) case c: hipLaunchKernelGGL(probe_issue_kernel<c>, dim3(blocks), dim3(threads), 0, s, out, 12345u, 777u, iters); break;
        WS_CASE(6) WS_CASE(7) WS_CASE(8) WS_CASE(9) WS_CASE(10) WS_CASE(11) WS_CASE(12) WS_CASE(13) WS_CASE(14) WS_CASE(16) WS_CASE(17) WS_CASE(18) WS_CASE(19) WS_CASE(20) WS_CASE(21) WS_CASE(22) WS_CASE(23) WS_CASE(24) WS_CASE(25) WS_CASE(26) WS_CASE(27) WS_CASE(28)
#undef WS_CASE
    }
}
#else
static void launch_issue_probe(int, uint32_t, uint32_t, hipStream_t, uint64_t*, int) {}       // (inline assembly: device only)
#endif

template <class Cv>
static int madd_probe_t(Context* C, int waves, double* gops) {
    hipStream_t s = C->stream;
#ifdef WSNARK_EMUL
    const uint32_t blocks = 1, iters = 3;
#else
    const uint32_t blocks = (uint32_t)C->num_cu * (uint32_t)waves, iters = 400;
#endif
    const uint32_t total = blocks * 256;
    // two "points": arbitrary reduced field elements as coordinates -- the mixed addition never tests curve membership, and a
    // timing probe only needs the generic path (no operand ever equals the accumulator)
    typename Cv::AffP hp[2];
    memset(&hp[0], 0x11, sizeof hp[0]);
    memset(&hp[1], 0x23, sizeof hp[1]);
    DevBuf in, out;
    WS_HIP_CHECK(in.alloc(sizeof hp));
    WS_HIP_CHECK(out.alloc((size_t)total * sizeof(typename Cv::PtP)));
    WS_HIP_CHECK(hipMemcpyAsync(in.p, hp, sizeof hp, hipMemcpyHostToDevice, s));
    WS_HIP_CHECK(hipStreamSynchronize(s));
    hipEvent_t a = nullptr, b = nullptr;
    WS_HIP_CHECK(hipEventCreate(&a));
    WS_HIP_CHECK(hipEventCreate(&b));
    double best = 0;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(a, s);
        hipLaunchKernelGGL(probe_madd_kernel<Cv>, dim3(blocks), dim3(256), 0, s, in.as<typename Cv::AffP>(), out.as<typename Cv::PtP>(), (int)iters);
        (void)hipEventRecord(b, s);
        if (hipEventSynchronize(b) != hipSuccess) break;
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        const double ops = (double)iters * total;
        if (rep && ms > 0 && ops / ms / 1e6 > best) best = ops / ms / 1e6;
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    WS_HIP_CHECK(hipGetLastError());
    *gops = best;
    return WS_OK;
}
static int madd_probe(Context* C, int g, int waves, double* gops) {
    return g == 2 ? madd_probe_t<G2R29>(C, waves, gops) : madd_probe_t<G1R29I>(C, waves, gops);
}

int peak_probe(int probe, double* gops) {
    Context* C = ctx();
    if (!C) return WS_ERR_NOINIT;
    // probes 101..108: probe 1 (the inlined product chain, ONE dependent chain per lane) at 1..8 wavefronts per SIMD instead of 8: what
    // the product's dependent multiply-add chain reaches at the occupancy of the accumulation kernels (G1: 3, G2: 2).
    // probes 201..208: the same with TWO independent product chains per lane, interleaved by the compiler.
    int waves = 8, chains = 1;
    if ((probe >= 301 && probe <= 308) || (probe >= 401 && probe <= 408)) return madd_probe(C, probe >= 401 ? 2 : 1, probe % 100, gops);
    if (probe >= 101 && probe <= 108) { waves = probe - 100; probe = 1; }
    else if (probe >= 201 && probe <= 208) { waves = probe - 200; probe = 1; chains = 2; }
    if (!gops || probe < 0 || probe > 28 || probe == 15) return WS_ERR_ARG;
    if (probe == 3 || probe == 4) return traffic_probe(C, probe, gops);
    hipStream_t s = C->stream;
#ifdef WSNARK_EMUL
    const uint32_t blocks = 2, threads = 256, total = blocks * threads;      // (CPU thread emulator: the control flow only, no rate)
    (void)waves;
#else
    const uint32_t blocks = (uint32_t)C->num_cu * (uint32_t)waves, threads = 256, total = blocks * threads;
#endif
    DevBuf in, out;
    WS_HIP_CHECK(in.alloc((size_t)total * 32));
    WS_HIP_CHECK(out.alloc((size_t)total * 32));
    WS_HIP_CHECK(hipMemsetAsync(in.p, 0x11, (size_t)total * 32, s));
    hipEvent_t a = nullptr, b = nullptr;
    WS_HIP_CHECK(hipEventCreate(&a));
    WS_HIP_CHECK(hipEventCreate(&b));
#ifdef WSNARK_EMUL
    const int iters = 2;
#else
    const int iters = probe == 2 ? 20000 : probe == 5 ? 8 : probe >= 6 ? 4000 : 2000;
#endif
    double best = 0;
    for (int rep = 0; rep < 4; rep++) {          // the first repetition warms the clocks; the best of the rest counts
        (void)hipEventRecord(a, s);
        if (probe == 0) hipLaunchKernelGGL(probe_modmul_kernel<Fq29>, dim3(blocks), dim3(threads), 0, s, in.as<Fe>(), out.as<Fe>(), iters);
        else if (probe == 1 && chains == 2) hipLaunchKernelGGL(probe_modmul2_kernel<Fq29I>, dim3(blocks), dim3(threads), 0, s, in.as<Fe>(), out.as<Fe>(), iters);
        else if (probe == 1) hipLaunchKernelGGL(probe_modmul_kernel<Fq29I>, dim3(blocks), dim3(threads), 0, s, in.as<Fe>(), out.as<Fe>(), iters);
        else if (probe == 5) hipLaunchKernelGGL(probe_inverse_kernel, dim3(blocks), dim3(threads), 0, s, in.as<Fe>(), out.as<Fe>(), iters);
        else if (probe >= 6) launch_issue_probe(probe, blocks, threads, s, out.as<uint64_t>(), iters);
        else hipLaunchKernelGGL(probe_mad_kernel, dim3(blocks), dim3(threads), 0, s, out.as<uint64_t>(), 12345u, 777u, iters);
        (void)hipEventRecord(b, s);
        if (hipEventSynchronize(b) != hipSuccess) break;
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        const double ops = (probe == 2 ? 8.0 : probe >= 6 ? 64.0 : (double)chains) * iters * (double)total;
        if (rep && ms > 0 && ops / ms / 1e6 > best) best = ops / ms / 1e6;
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    WS_HIP_CHECK(hipGetLastError());
    *gops = best;
    return WS_OK;
}

}  // namespace wsnark
