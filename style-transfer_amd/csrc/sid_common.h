// Shared host-side helpers of libsid.so (error reporting, argument checks).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdarg>
#include <cstddef>
#include <cstdio>

#include "../../include/sid.h"

namespace sid {

// Thread-local message returned by sid_last_error().
char* error_buffer();
int fail(int code, const char* fmt, ...);

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

#define SID_CHECK_ARG(cond, ...) \
    do { if (!(cond)) return ::sid::fail(SID_EINVAL, __VA_ARGS__); } while (0)

#define SID_CHECK_HIP(expr) \
    do { hipError_t e_ = (expr); \
         if (e_ != hipSuccess) return ::sid::fail(SID_EIO, "%s: %s", #expr, hipGetErrorString(e_)); } while (0)

// Raise the dynamic-LDS limit of kernel `Kern` to `bytes` (needed above 48 KB), once per device the
// calling process uses it on.  `bytes` must be the same on every call for one kernel.
template <auto Kern>
inline int allow_dynamic_lds(size_t bytes) {
    static std::atomic<unsigned long long> done{0};       // one bit per device ordinal (mod 64)
    int dev = 0;
    SID_CHECK_HIP(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        SID_CHECK_HIP(hipFuncSetAttribute((const void*)Kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        done.fetch_or(bit, std::memory_order_release);
    }
    return SID_OK;
}

}  // namespace sid
