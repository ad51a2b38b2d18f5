// microbench.hip -- instruction-rate probes on gfx950 for the integer roofline of the prove path
// ("report achieved int-MAD fraction against a v_mad_u64_u32 peak
// micro-benchmarked on the box").  Standalone: hipcc --offload-arch=gfx950 -O3 microbench.hip -o microbench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../wasmsnark_amd/csrc/field.h"
using namespace wsnark;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

// 8 independent chains of 32x32+64 multiply-adds
__global__ void k_mad64(uint64_t* out, uint32_t a, uint32_t b, int iters) {
    uint64_t c0 = threadIdx.x, c1 = c0 + 1, c2 = c0 + 2, c3 = c0 + 3, c4 = c0 + 4, c5 = c0 + 5, c6 = c0 + 6, c7 = c0 + 7;
    uint32_t x = a + threadIdx.x, y = b;
    for (int i = 0; i < iters; i++) {
        c0 = (uint64_t)x * y + c0; c1 = (uint64_t)x * y + c1; c2 = (uint64_t)x * y + c2; c3 = (uint64_t)x * y + c3;
        c4 = (uint64_t)x * y + c4; c5 = (uint64_t)x * y + c5; c6 = (uint64_t)x * y + c6; c7 = (uint64_t)x * y + c7;
        x = (uint32_t)c0; y = (uint32_t)(c7 >> 32) | 1;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;
}
__global__ void k_mullo(uint32_t* out, uint32_t a, int iters) {
    uint32_t c0 = threadIdx.x | 1, c1 = c0 + 2, c2 = c0 + 4, c3 = c0 + 6, c4 = c0 + 8, c5 = c0 + 10, c6 = c0 + 12, c7 = c0 + 14;
    for (int i = 0; i < iters; i++) { c0 *= a; c1 *= a; c2 *= a; c3 *= a; c4 *= a; c5 *= a; c6 *= a; c7 *= a; a += c0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;
}
__global__ void k_mulhi(uint32_t* out, uint32_t a, int iters) {
    uint32_t c0 = ~threadIdx.x, c1 = c0 - 2, c2 = c0 - 4, c3 = c0 - 6, c4 = c0 - 8, c5 = c0 - 10, c6 = c0 - 12, c7 = c0 - 14;
    for (int i = 0; i < iters; i++) {
        c0 = __umulhi(c0, a) | 0x80000000u; c1 = __umulhi(c1, a) | 0x80000000u; c2 = __umulhi(c2, a) | 0x80000000u; c3 = __umulhi(c3, a) | 0x80000000u;
        c4 = __umulhi(c4, a) | 0x80000000u; c5 = __umulhi(c5, a) | 0x80000000u; c6 = __umulhi(c6, a) | 0x80000000u; c7 = __umulhi(c7, a) | 0x80000000u;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;
}
__global__ void k_mad24(uint32_t* out, uint32_t a, int iters) {
    uint32_t c0 = threadIdx.x, c1 = c0 + 1, c2 = c0 + 2, c3 = c0 + 3, c4 = c0 + 4, c5 = c0 + 5, c6 = c0 + 6, c7 = c0 + 7;
    for (int i = 0; i < iters; i++) {
        c0 = __umul24(c0, a) + c1; c1 = __umul24(c1, a) + c2; c2 = __umul24(c2, a) + c3; c3 = __umul24(c3, a) + c4;
        c4 = __umul24(c4, a) + c5; c5 = __umul24(c5, a) + c6; c6 = __umul24(c6, a) + c7; c7 = __umul24(c7, a) + c0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;
}
__global__ void k_add64(uint64_t* out, uint64_t a, int iters) {
    uint64_t c0 = threadIdx.x, c1 = c0 + 1, c2 = c0 + 2, c3 = c0 + 3, c4 = c0 + 4, c5 = c0 + 5, c6 = c0 + 6, c7 = c0 + 7;
    for (int i = 0; i < iters; i++) { c0 += a; c1 += c0; c2 += c1; c3 += c2; c4 += c3; c5 += c4; c6 += c5; c7 += c6; a ^= c7; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;
}
__global__ void k_fma64(double* out, double a, int iters) {
    double c0 = threadIdx.x, c1 = c0 + 1, c2 = c0 + 2, c3 = c0 + 3, c4 = c0 + 4, c5 = c0 + 5, c6 = c0 + 6, c7 = c0 + 7;
    for (int i = 0; i < iters; i++) {
        c0 = fma(c0, a, c1); c1 = fma(c1, a, c2); c2 = fma(c2, a, c3); c3 = fma(c3, a, c4);
        c4 = fma(c4, a, c5); c5 = fma(c5, a, c6); c6 = fma(c6, a, c7); c7 = fma(c7, a, c0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}
// the product's Montgomery multiplier: CHAINS independent dependent-chains per lane
template <int CHAINS>
__global__ void k_modmul(Fe* out, const Fe* in, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    Fe x[CHAINS];
    Fe y = in[t];
#pragma unroll
    for (int k = 0; k < CHAINS; k++) { x[k] = in[t]; x[k].l[0] += k; }
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < CHAINS; k++) x[k] = Fq::mul(x[k], y);
    }
    Fe r = x[0];
#pragma unroll
    for (int k = 1; k < CHAINS; k++) r = Fq::add(r, x[k]);
    out[t] = r;
}

// ---- candidate multipliers (timing probes only; not canonical, not used by the product) ----
#define M29 0x1FFFFFFFu
// radix 2^29, 9 limbs, one 64-bit accumulator per column chain: no carries, no moves
__device__ __forceinline__ void mul_r29(const uint32_t a[9], const uint32_t b[9], uint32_t r[9]) {
    const uint32_t p[9] = {0x187cfd47u, 0x10460b6cu, 0x1c72a34fu, 0x02d522d0u, 0x1585d978u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
    const uint32_t np = 0x04866389u;
    uint32_t m[9];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a[i] * b[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * p[k - i];
        m[k] = ((uint32_t)acc * np) & M29;
        acc += (uint64_t)m[k] * p[0];
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i <= 8; i++) acc += (uint64_t)a[i] * b[k - i];
#pragma unroll
        for (int i = k - 8; i <= 8; i++) acc += (uint64_t)m[i] * p[k - i];
        r[k - 9] = (uint32_t)acc & M29;
        acc >>= 29;
    }
    r[8] = (uint32_t)acc;
}
// same with two interleaved accumulators (even / odd product index) to shorten the dependent chain
__device__ __forceinline__ void mul_r29x2(const uint32_t a[9], const uint32_t b[9], uint32_t r[9]) {
    const uint32_t p[9] = {0x187cfd47u, 0x10460b6cu, 0x1c72a34fu, 0x02d522d0u, 0x1585d978u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
    const uint32_t np = 0x04866389u;
    uint32_t m[9];
    uint64_t acc = 0, acc2 = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a[i] * b[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc2 += (uint64_t)m[i] * p[k - i];
        acc += acc2; acc2 = 0;
        m[k] = ((uint32_t)acc * np) & M29;
        acc += (uint64_t)m[k] * p[0];
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i <= 8; i++) acc += (uint64_t)a[i] * b[k - i];
#pragma unroll
        for (int i = k - 8; i <= 8; i++) acc2 += (uint64_t)m[i] * p[k - i];
        acc += acc2; acc2 = 0;
        r[k - 9] = (uint32_t)acc & M29;
        acc >>= 29;
    }
    r[8] = (uint32_t)acc;
}
template <int VAR, int CHAINS>
__global__ void k_r29(uint32_t* out, const uint32_t* in, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t x[CHAINS][9], y[9];
#pragma unroll
    for (int j = 0; j < 9; j++) { y[j] = in[t * 8 + (j & 7)] & M29; }
#pragma unroll
    for (int k = 0; k < CHAINS; k++)
#pragma unroll
        for (int j = 0; j < 9; j++) x[k][j] = (in[t * 8 + (j & 7)] + k) & M29;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < CHAINS; k++) { if (VAR == 0) mul_r29(x[k], y, x[k]); else mul_r29x2(x[k], y, x[k]); }
    }
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < CHAINS; k++)
#pragma unroll
        for (int j = 0; j < 9; j++) r ^= x[k][j];
    out[t] = r;
}

// One Fermat inversion per lane (x^(p-2): 253 squarings + the multiplications of the exponent's set bits), every lane
// active: what a LANE-PARALLEL batch inversion costs per batch (the batch-affine accumulation analysis in DESIGN.md).
__global__ void k_inv29(uint32_t* out, const uint32_t* in, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t e[4] = {0x3c208c16d87cfd45ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};   // q - 2
    uint32_t x[9], acc[9], base[9];
#pragma unroll
    for (int j = 0; j < 9; j++) x[j] = (in[t * 8 + (j & 7)] | 1) & M29;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 9; j++) { acc[j] = x[j]; base[j] = x[j]; }
        for (int i = 1; i < 254; i++) {
            mul_r29(base, base, base);
            if ((e[i >> 6] >> (i & 63)) & 1) mul_r29(acc, base, acc);
        }
#pragma unroll
        for (int j = 0; j < 9; j++) x[j] = acc[j] & M29;
    }
    uint32_t r = 0;
#pragma unroll
    for (int j = 0; j < 9; j++) r ^= x[j];
    out[t] = r;
}

template <class F>
static int run(const char* name, F launch, double ops_per_thread_iter, int iters, int threads_total) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    launch(iters / 10);   // warm
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a, 0));
    launch(iters);
    CHECK(hipEventRecord(b, 0));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    double ops = ops_per_thread_iter * iters * (double)threads_total;
    printf("{\"probe\": \"%s\", \"ms\": %.3f, \"Gops_per_s\": %.1f}\n", name, ms, ops / ms / 1e6);
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * 8, threads = 256, total = blocks * threads;
    printf("{\"device\": \"%s\", \"arch\": \"%s\", \"cus\": %d, \"clock_mhz\": %d}\n", prop.name, prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000);
    void* buf;
    CHECK(hipMalloc(&buf, (size_t)total * 32));
    Fe* in;
    CHECK(hipMalloc((void**)&in, (size_t)total * 32));
    CHECK(hipMemset(in, 0x11, (size_t)total * 32));
    const int it = 20000;
    run("v_mad_u64_u32 (32x32+64)", [&](int n) { hipLaunchKernelGGL(k_mad64, dim3(blocks), dim3(threads), 0, 0, (uint64_t*)buf, 12345u, 777u, n); }, 8, it, total);
    run("v_mul_lo_u32", [&](int n) { hipLaunchKernelGGL(k_mullo, dim3(blocks), dim3(threads), 0, 0, (uint32_t*)buf, 12345u, n); }, 8, it, total);
    run("v_mul_hi_u32", [&](int n) { hipLaunchKernelGGL(k_mulhi, dim3(blocks), dim3(threads), 0, 0, (uint32_t*)buf, 0xfffffff1u, n); }, 8, it, total);
    run("v_mad_u32_u24", [&](int n) { hipLaunchKernelGGL(k_mad24, dim3(blocks), dim3(threads), 0, 0, (uint32_t*)buf, 12345u, n); }, 8, it, total);
    run("add_u64", [&](int n) { hipLaunchKernelGGL(k_add64, dim3(blocks), dim3(threads), 0, 0, (uint64_t*)buf, 12345ull, n); }, 8, it, total);
    run("v_fma_f64", [&](int n) { hipLaunchKernelGGL(k_fma64, dim3(blocks), dim3(threads), 0, 0, (double*)buf, 1.0000001, n); }, 8, it, total);
    const int im = 2000;
    run("mont_mul32 x1 chain (modmul/s)", [&](int n) { hipLaunchKernelGGL(k_modmul<1>, dim3(blocks), dim3(threads), 0, 0, (Fe*)buf, in, n); }, 1, im, total);
    run("mont_mul32 x2 chains (modmul/s)", [&](int n) { hipLaunchKernelGGL(k_modmul<2>, dim3(blocks), dim3(threads), 0, 0, (Fe*)buf, in, n); }, 2, im, total);
    run("mont_mul32 x4 chains (modmul/s)", [&](int n) { hipLaunchKernelGGL(k_modmul<4>, dim3(blocks), dim3(threads), 0, 0, (Fe*)buf, in, n); }, 4, im, total);
    run("radix29 mul x1 chain (modmul/s)", [&](int n) { hipLaunchKernelGGL((k_r29<0, 1>), dim3(blocks), dim3(threads), 0, 0, (uint32_t*)buf, (const uint32_t*)in, n); }, 1, im, total);
    run("radix29 mul x2 chains (modmul/s)", [&](int n) { hipLaunchKernelGGL((k_r29<0, 2>), dim3(blocks), dim3(threads), 0, 0, (uint32_t*)buf, (const uint32_t*)in, n); }, 2, im, total);
    run("radix29 2-acc mul x1 chain (modmul/s)", [&](int n) { hipLaunchKernelGGL((k_r29<1, 1>), dim3(blocks), dim3(threads), 0, 0, (uint32_t*)buf, (const uint32_t*)in, n); }, 1, im, total);
    run("mont_mul32 x1 chain, 2 blocks/CU", [&](int n) { hipLaunchKernelGGL(k_modmul<1>, dim3(prop.multiProcessorCount * 2), dim3(threads), 0, 0, (Fe*)buf, in, n); }, 1, im, prop.multiProcessorCount * 2 * threads);
    run("radix29 x1 chain, 2 blocks/CU", [&](int n) { hipLaunchKernelGGL((k_r29<0, 1>), dim3(prop.multiProcessorCount * 2), dim3(threads), 0, 0, (uint32_t*)buf, (const uint32_t*)in, n); }, 1, im, prop.multiProcessorCount * 2 * threads);
    run("radix29 x1 chain, 4 blocks/CU", [&](int n) { hipLaunchKernelGGL((k_r29<0, 1>), dim3(prop.multiProcessorCount * 4), dim3(threads), 0, 0, (uint32_t*)buf, (const uint32_t*)in, n); }, 1, im, prop.multiProcessorCount * 4 * threads);
    run("Fermat inversion radix29, every lane active (inversions/s)", [&](int n) { hipLaunchKernelGGL(k_inv29, dim3(blocks), dim3(threads), 0, 0, (uint32_t*)buf, (const uint32_t*)in, n < 1 ? 1 : n); }, 1, 20, total);
    return 0;
}
