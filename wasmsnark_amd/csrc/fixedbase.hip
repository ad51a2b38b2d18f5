// fixedbase.hip -- out[i] = scalars[i] * base, affine.  NOT part of the reference's prove path:
// it is the build's helper for generating synthetic proving keys / expected proofs at full size
// (the reference ships no proving key: SURVEY.md fact 9); wasmsnark_amd/synth.py drives it.
#include <string.h>

#include "internal.h"

namespace wsnark {

template <class C>
__global__ __launch_bounds__(256) void mul_base_kernel(typename C::Aff base, const Fe* __restrict__ scalars, uint64_t n,
                                                         typename C::Aff* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fe s = Fr::reduce_full(scalars[i]);
    typename C::Pt acc = C::infinity();
    bool started = false;
    for (int bit = 255; bit >= 0; bit--) {
        if (started) acc = C::dbl(acc);
        if ((s.l[bit >> 6] >> (bit & 63)) & 1) {
            C::madd(acc, base, false);
            started = true;
        }
    }
    typename C::Aff r;
    if (C::is_inf(acc)) {
        memset(&r, 0, sizeof r);   // x == 0 encodes infinity (src/build_multiexp.js:335-349)
    } else {
        auto j = C::to_affine_jac(acc);
        r.x = j.x;
        r.y = j.y;
    }
    out[i] = r;
}

template <class C>
static int mul_base_host(const void* base, const void* scalars, uint64_t n, void* out) {
    Context* X = ctx();
    if (!X) return WS_ERR_NOINIT;
    if (n == 0) return WS_OK;
    if (!base || !scalars || !out) return WS_ERR_ARG;
    typename C::Aff b;
    memcpy(&b, base, sizeof b);
    DevBuf ds, dout;
    WS_HIP_CHECK(ds.alloc(n * 32));
    WS_HIP_CHECK(dout.alloc(n * sizeof(typename C::Aff)));
    WS_HIP_CHECK(hipMemcpyAsync(ds.p, scalars, n * 32, hipMemcpyHostToDevice, X->stream));
    hipLaunchKernelGGL(mul_base_kernel<C>, dim3(ceil_div_u64(n, 256)), dim3(256), 0, X->stream, b, ds.as<Fe>(), n,
                       dout.as<typename C::Aff>());
    WS_HIP_CHECK(hipGetLastError());
    WS_HIP_CHECK(hipMemcpyAsync(out, dout.p, n * sizeof(typename C::Aff), hipMemcpyDeviceToHost, X->stream));
    WS_HIP_CHECK(hipStreamSynchronize(X->stream));
    return WS_OK;
}

int g1_mul_base_batch(const void* base, const void* scalars, uint64_t n, void* out) { return mul_base_host<G1>(base, scalars, n, out); }
int g2_mul_base_batch(const void* base, const void* scalars, uint64_t n, void* out) { return mul_base_host<G2>(base, scalars, n, out); }

WS_DEFINE_WARM(fixedbase)

}  // namespace wsnark
