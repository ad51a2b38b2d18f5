// Host-to-device copy strategies for the host-pointer boundary (96 MB = a 2^20-pair G1 MSM input).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)
int main() {
    const size_t N = 96u << 20;
    char* h = (char*)malloc(N);
    memset(h, 1, N);
    void* d = nullptr;
    hipStream_t s; CK(hipStreamCreate(&s));
    for (int rep = 0; rep < 3; rep++) {
        double t0 = now(); CK(hipMalloc(&d, N)); double t1 = now(); CK(hipFree(d)); double t2 = now();
        printf("hipMalloc %.3f ms  hipFree %.3f ms\n", t1 - t0, t2 - t1);
    }
    CK(hipMalloc(&d, N));
    for (int rep = 0; rep < 3; rep++) {
        double t0 = now(); CK(hipMemcpyAsync(d, h, N, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); double t1 = now();
        printf("pageable hipMemcpyAsync 96 MB: %.3f ms (%.1f GB/s)\n", t1 - t0, N / (t1 - t0) / 1e6);
    }
    for (int rep = 0; rep < 3; rep++) {
        double t0 = now(); CK(hipHostRegister(h, N, hipHostRegisterDefault)); double t1 = now();
        CK(hipMemcpyAsync(d, h, N, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); double t2 = now();
        CK(hipHostUnregister(h)); double t3 = now();
        printf("register %.3f ms  copy %.3f ms (%.1f GB/s)  unregister %.3f ms\n", t1 - t0, t2 - t1, N / (t2 - t1) / 1e6, t3 - t2);
    }
    // staged: T threads memcpy chunks into pinned double buffers, DMA per chunk
    char* pin = nullptr; const size_t CH = 8u << 20; const int NBUF = 4;
    CK(hipHostMalloc((void**)&pin, CH * NBUF, 0));
    hipEvent_t ev[NBUF]; for (auto& e : ev) CK(hipEventCreate(&e));
    for (int T : {1, 4, 8}) {
        for (int rep = 0; rep < 2; rep++) {
            double t0 = now();
            size_t nch = (N + CH - 1) / CH;
            for (size_t c = 0; c < nch; c++) {
                int b = c % NBUF;
                if (c >= (size_t)NBUF) CK(hipEventSynchronize(ev[b]));
                size_t off = c * CH, len = off + CH <= N ? CH : N - off;
                std::vector<std::thread> th;
                for (int k = 0; k < T; k++) th.emplace_back([=] { size_t a = len * k / T, e = len * (k + 1) / T; memcpy(pin + b * CH + a, h + off + a, e - a); });
                for (auto& x : th) x.join();
                CK(hipMemcpyAsync((char*)d + off, pin + b * CH, len, hipMemcpyHostToDevice, s));
                CK(hipEventRecord(ev[b], s));
            }
            CK(hipStreamSynchronize(s));
            double t1 = now();
            printf("staged via pinned, %d memcpy threads: %.3f ms (%.1f GB/s)\n", T, t1 - t0, N / (t1 - t0) / 1e6);
        }
    }
    return 0;
}
