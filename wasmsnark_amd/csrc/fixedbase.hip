// fixedbase.hip -- out[i] = scalars[i] * base, affine.  NOT part of the reference's prove path:
// it is the build's helper for generating synthetic proving keys / expected proofs at full size
// (the reference ships no proving key: SURVEY.md fact 9); wasmsnark_amd/synth.py drives it.
#include <string.h>

#include <memory>
#include <mutex>

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

// ---- resident bases (round 5): a point set that stays on the device as fixed-base window tables ----
// The reference's g1_multiexp / g2_multiexp take the points with every call (src/bn128.js:353-415); callers that sum over the SAME
// bases again and again -- what a prover's key sections are -- can load them once: the set becomes rows x n points, row w =
// 2^(c w) * P (the layout of a resident proving key's sections, prove.hip), and every sum over it is ONE bucket set, one tail and no
// doubling chain on the host: ~1.0 ms per 2^20 G1 sum instead of 2.0 (the points also no longer cross PCIe per call).
struct ResidentPoints {
    Context* owner = nullptr;
    int which = 0;                 // 0 = G1, 1 = G2
    uint64_t n = 0;
    uint32_t table_c = 0;
    DevBuf table;
};
Context* points_context(const ResidentPoints* H) { return H ? H->owner : nullptr; }
int points_load(int which, const void* h_points, uint64_t n, ResidentPoints** out) {
    Context* C = ctx();
    if (!C) return WS_ERR_NOINIT;
    if (!out || (which != 0 && which != 1) || (n && !h_points)) return WS_ERR_ARG;
    if (n == 0 || n > ((uint64_t)1 << 27)) return WS_ERR_SIZE;
    std::unique_ptr<ResidentPoints> H(new ResidentPoints());
    H->owner = C; H->which = which; H->n = n;
    const size_t psz = which ? 128 : 64;
    const uint32_t tc = msm_table_window(n), rows = msm_table_rows(tc);
    if ((uint64_t)rows * n >= ((uint64_t)1 << 31)) { set_last_error("points_load: too many points for one table"); return WS_ERR_SIZE; }
    H->table_c = tc;
    WS_HIP_CHECK(H->table.alloc((size_t)rows * n * psz));
    hipStream_t s = C->stream;
    int rc = upload_staged(H->table.p, h_points, (size_t)n * psz, s);
    if (!rc) rc = msm_prepare_points(which, H->table.p, n, s);
    // rows 1.. in the row-per-launch form through a slab of the context (held under the build mutex: key loads use the same one)
    if (!rc) {
        std::lock_guard<std::mutex> lk(C->build_mu);
        uint64_t lanes = n < (1u << 16) ? ((n + 63) & ~(uint64_t)63) : (1u << 16);
        const size_t tmp_bytes = msm_table_scratch_bytes(lanes);
        WS_HIP_CHECK(hipStreamSynchronize(C->build_q));
        if (C->build_tmp.bytes < tmp_bytes && C->build_tmp.reserve(tmp_bytes) != hipSuccess) { (void)hipGetLastError(); rc = msm_build_table(which, H->table.p, n, tc, s); }
        else rc = msm_build_table(which, H->table.p, n, tc, s, C->build_tmp.p, C->build_tmp.bytes);
        if (!rc) WS_HIP_CHECK(hipStreamSynchronize(s));
    }
    if (rc) return rc;
    *out = H.release();
    return WS_OK;
}
void points_free(ResidentPoints* H) { delete H; }
void points_info(const ResidentPoints* H, int* which, uint64_t* n, uint32_t* table_c, uint32_t* rows, uint64_t* bytes) {
    if (which) *which = H->which;
    if (n) *n = H->n;
    if (table_c) *table_c = H->table_c;
    if (rows) *rows = msm_table_rows(H->table_c);
    if (bytes) *bytes = H->table.bytes;
}
// sum_i scalars[i] * P_i over ALL n points of the set; scalars on the host (staged) or on the device
int points_msm(ResidentPoints* H, const void* scalars, bool on_device, uint64_t n, void* out, hipStream_t s) {
    Context* C = ctx();
    if (!C) return WS_ERR_NOINIT;
    if (!H || !out || !scalars) return WS_ERR_ARG;
    if (n != H->n) { set_last_error("points_msm: the set holds " + std::to_string(H->n) + " points (one scalar per point)"); return WS_ERR_SIZE; }
    LaneLock L = acquire_lane(C);
    if (!s) s = L->stream;
    const Fe* d_sc = (const Fe*)scalars;
    int rc;
    if (!on_device) {
        WS_HIP_CHECK(L->host_in[0].reserve((size_t)n * 32));
        if ((rc = upload_staged(L->host_in[0].p, scalars, (size_t)n * 32, s))) return rc;
        d_sc = L->host_in[0].as<Fe>();
    }
    msm_select_plan(*L, 0);
    if ((rc = msm_plan_dev(*L, d_sc, n, WindowShard{}, s, H->table_c))) return rc;
    int slot = -1;
    if (H->which == 0) {
        if ((rc = msm_g1_launch(*L, H->table.as<Affine<Fq>>(), true, &slot, s))) return rc;
        XYZZ<Fq> r;
        if ((rc = msm_g1_finish(*L, slot, &r))) return rc;
        const Jac<Fq> j = G1::to_affine_jac(r);
        memcpy(out, &j, sizeof j);
    } else {
        if ((rc = msm_g2_launch(*L, H->table.as<Affine<Fq2>>(), true, &slot, s))) return rc;
        XYZZ<Fq2> r;
        if ((rc = msm_g2_finish(*L, slot, &r))) return rc;
        const Jac<Fq2> j = G2::to_affine_jac(r);
        memcpy(out, &j, sizeof j);
    }
    return WS_OK;
}

WS_DEFINE_WARM(fixedbase)

}  // namespace wsnark
