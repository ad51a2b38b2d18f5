// rt.h -- thin runtime helpers shared by the .hip translation units.
//
// Normal builds include the HIP runtime.  When WSNARK_EMUL is defined the very same
// kernel sources are compiled by g++ against tests/emul/hip_emul.h, a CPU thread
// emulator used ONLY by the CPU test-suite to exercise kernel index math without a
// GPU (see DESIGN.md "CPU emulation harness").  The product library never defines
// WSNARK_EMUL.
#pragma once
#ifdef WSNARK_EMUL
#include "hip_emul.h"
#else
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>
#include <stdio.h>
#include <string>

namespace wsnark {

// status codes of the C ABI (include/wsnark.h)
enum {
    WS_OK = 0,
    WS_ERR_SIZE = 1,      // non power-of-two / too large / n inconsistent
    WS_ERR_FORMAT = 2,    // malformed proving key / pols blob / offsets out of range
    WS_ERR_HIP = 3,       // HIP runtime failure (message in wsnark_last_error)
    WS_ERR_ARG = 4,       // null pointer / bad handle
    WS_ERR_NOINIT = 5,
};

void set_last_error(const std::string& s);

#define WS_HIP_CHECK(expr)                                                                         \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            ::wsnark::set_last_error(std::string(#expr) + ": " + hipGetErrorString(_e));           \
            return ::wsnark::WS_ERR_HIP;                                                           \
        }                                                                                          \
    } while (0)

#ifdef WSNARK_EMUL
#define WS_DYN_SMEM(type, name) type* name = reinterpret_cast<type*>(::hip_emul::dyn_smem())
#else
#define WS_DYN_SMEM(type, name)                                   \
    extern __shared__ __align__(16) unsigned char _ws_dyn_smem[]; \
    type* name = reinterpret_cast<type*>(_ws_dyn_smem)
#endif

static inline uint32_t ceil_div_u64(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

// simple RAII device buffer
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() {}
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    hipError_t alloc(size_t n) {
        release();
        if (n == 0) n = 16;
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) bytes = n; else p = nullptr;
        return e;
    }
    // grow-only reuse
    hipError_t reserve(size_t n) { return (n <= bytes && p) ? hipSuccess : alloc(n); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr; bytes = 0;
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

}  // namespace wsnark
