// Shared host-side helpers of libsid.so (error reporting, argument checks).
#pragma once
#include <hip/hip_runtime.h>
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

}  // namespace sid
