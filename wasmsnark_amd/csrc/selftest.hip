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
    if (op == WSNARK_ST_INVERSE) { set_last_error("selftest: inversion is host work (impl 2)"); return WS_ERR_ARG; }
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
    if (n > (1u << 20) || op < 0 || op > 7) return WS_ERR_ARG;
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
    }
    return WS_ERR_ARG;
}

}  // namespace wsnark
