// hip_emul.h -- a tiny CPU stand-in for the parts of the HIP runtime and device
// language that wasmsnark_amd/csrc uses.  TEST INFRASTRUCTURE ONLY.
//
// Purpose: this build container has no GPU.  To exercise kernel INDEX MATH before
// spending GPU minutes, the CPU test-suite compiles the very same .hip sources with
// g++ and -DWSNARK_EMUL against this header into tests/emul/libwsnark_emul.so.  Each
// workgroup runs its threads as cooperative coroutines (ucontext); __syncthreads()
// yields until every live thread of the block has arrived.  Blocks run one after the
// other.  Nothing here is built into, shipped with, or loaded by the product library
// (wasmsnark_amd/libwsnark.so), and GPU parity tests never use it.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define __align__(n) __attribute__((aligned(n)))

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_emul { unsigned x, y, z; };

namespace hip_emul {
extern thread_local uint3_emul t_threadIdx, t_blockIdx;
extern thread_local dim3 t_blockDim, t_gridDim;
void* dyn_smem();
void sync_threads();
uint32_t shfl_exchange(uint32_t v, int src_lane, int width);
uint32_t pair_exchange(uint32_t v);          // with lane ^ 1 (a DPP quad_perm [1,0,3,2] swap on the device)
uint32_t pair_exchange_dist(uint32_t v, int dist);   // with lane ^ dist, dist = 1 or 2 (quad_perm [2,3,0,1])
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
}  // namespace hip_emul

#define threadIdx (::hip_emul::t_threadIdx)
#define blockIdx (::hip_emul::t_blockIdx)
#define blockDim (::hip_emul::t_blockDim)
#define gridDim (::hip_emul::t_gridDim)

static inline void __syncthreads() { ::hip_emul::sync_threads(); }
static inline unsigned __brev(unsigned v) {
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
    v = ((v >> 8) & 0x00FF00FFu) | ((v & 0x00FF00FFu) << 8);
    return (v >> 16) | (v << 16);
}
static inline int __clz(unsigned v) { return v ? __builtin_clz(v) : 32; }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }

// wave-level exchange (64 lanes): every lane of the wave must call it
static inline uint32_t __shfl(uint32_t v, int src, int width = 64) { return ::hip_emul::shfl_exchange(v, src, width); }
static inline uint32_t __shfl_down(uint32_t v, unsigned d, int width = 64) {
    int lane = (int)(threadIdx.x & 63);
    int src = lane + (int)d;
    if ((src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
    return ::hip_emul::shfl_exchange(v, src, 64);
}
static inline uint32_t __shfl_xor(uint32_t v, int m, int width = 64) {
    (void)width;
    return ::hip_emul::shfl_exchange(v, (int)(threadIdx.x & 63) ^ m, 64);
}

// wave vote: every lane of the wave must call it (butterfly OR of one-bit masks over the exchange primitive)
static inline unsigned long long __ballot(int pred) {
    const int lane = (int)(threadIdx.x & 63);
    uint32_t lo = (pred && lane < 32) ? (1u << lane) : 0u, hi = (pred && lane >= 32) ? (1u << (lane - 32)) : 0u;
    for (int m = 1; m < 64; m <<= 1) {
        lo |= ::hip_emul::shfl_exchange(lo, lane ^ m, 64);
        hi |= ::hip_emul::shfl_exchange(hi, lane ^ m, 64);
    }
    return ((unsigned long long)hi << 32) | lo;
}
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }

template <class T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T> static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> static inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
static inline void __threadfence() {}

// ---- host runtime ----
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600 };
typedef struct emul_stream* hipStream_t;
typedef struct emul_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipDeviceProp_t { int multiProcessorCount; char name[64]; char gcnArchName[64]; };

// ---- devices (round 6, closing): WSNARK_EMUL_DEVICES = how many "devices" the emulator shows (default 1).  They share the host's
// memory; what they do NOT share is what real devices do not share either: a thread's CURRENT device, and the device a queue or an
// event belongs to.  Work queued on a queue of another device than the thread's current one, and an event of one device recorded on a
// queue of another, fail the way the runtime makes them fail -- so that the CPU suite sees the device-selection mistakes of a
// several-GPU process (a helper thread that never selected its device; a record across devices) that one-GPU boxes cannot show.
enum { hipErrorInvalidDevice = 101, hipErrorInvalidHandle = 400 };
int emul_device_count();
int& emul_current_device();                     // this thread's
hipError_t& emul_last_error();                  // this thread's sticky error (hipGetLastError reads and clears it)
static inline int emul_handle_device(const void* h) { return (int)((uintptr_t)h & 63); }
static inline hipError_t emul_fail(hipError_t e) { emul_last_error() = e; return e; }
static inline const char* hipGetErrorString(hipError_t e) {
    return e == hipSuccess ? "success" : e == hipErrorInvalidHandle ? "emulated: a queue / event of another device than the thread's current one"
           : e == hipErrorInvalidDevice ? "emulated: no such device" : "emulated failure";
}
static inline hipError_t hipGetLastError() { hipError_t e = emul_last_error(); emul_last_error() = hipSuccess; return e; }
static inline hipError_t hipSetDevice(int d) { if (d < 0 || d >= emul_device_count()) return emul_fail(hipErrorInvalidDevice); emul_current_device() = d; return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = emul_current_device(); return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = emul_device_count(); return hipSuccess; }
// a queue of the thread's current device?  (nullptr = the current device's default queue)
static inline bool emul_queue_ok(const void* s) { return s == nullptr || emul_handle_device(s) == emul_current_device(); }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    p->multiProcessorCount = 4; strcpy(p->name, "cpu-emulator"); strcpy(p->gcnArchName, "emul"); return hipSuccess;
}
// WSNARK_EMUL_MAX_ALLOC (bytes): larger requests fail like a full device would (tests of the out-of-memory paths)
static inline hipError_t hipMalloc(void** p, size_t n) {
    if (const char* e = getenv("WSNARK_EMUL_MAX_ALLOC")) { if (n > (size_t)strtoull(e, nullptr, 10)) { *p = nullptr; return hipErrorOutOfMemory; } }
    *p = aligned_alloc(64, (n + 63) & ~(size_t)63);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { *p = aligned_alloc(64, (n + 63) & ~(size_t)63); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t q) { if (!emul_queue_ok(q)) return emul_fail(hipErrorInvalidHandle); memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyPeerAsync(void* d, int, const void* s, int, size_t n, hipStream_t q) { if (!emul_queue_ok(q)) return emul_fail(hipErrorInvalidHandle); memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t q) { if (!emul_queue_ok(q)) return emul_fail(hipErrorInvalidHandle); memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b) { *free_b = (size_t)8 << 30; *total_b = (size_t)8 << 30; return hipSuccess; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = 0; return hipSuccess; }
// Every queue gets a handle of its own (never dereferenced): code that compares queues -- context.hip's queue_of_context, which keeps
// an upload on the context whose queue it names -- then means on the emulator what it means on the device.
hipStream_t emul_new_stream_handle();
static inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = emul_new_stream_handle(); return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = emul_new_stream_handle(); return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = emul_new_stream_handle(); return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipEvent_t emul_new_event_handle();            // (non-null: code tests handles; carries the device it was created on)
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = emul_new_event_handle(); return hipSuccess; }
static const unsigned hipEventDisableTiming = 2;
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = emul_new_event_handle(); return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
// (the runtime compares the event's device with the queue's: libamdhip64 7.2 returns hipErrorInvalidHandle on a mismatch)
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t q) {
    const int qd = q ? emul_handle_device(q) : emul_current_device();
    if (emul_handle_device(e) != qd) return emul_fail(hipErrorInvalidHandle);
    return hipSuccess;
}
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
template <class F> static inline hipError_t hipFuncSetAttribute(F, int, int) { return hipSuccess; }
#define hipStreamNonBlocking 1

namespace hip_emul {
template <class K, class... Args>
static inline void emul_launch(K kernel, dim3 g, dim3 b, size_t sh, Args... args) {
    launch(g, b, sh, [=]() { kernel(args...); });
}
}  // namespace hip_emul
// (a launch on a queue of another device than the thread's current one is not run: it leaves the sticky error the library's
//  hipGetLastError() checks pick up)
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    (emul_queue_ok(stream) ? ::hip_emul::emul_launch(kernel, (grid), (block), (shmem), __VA_ARGS__) : (void)emul_fail(hipErrorInvalidHandle))
