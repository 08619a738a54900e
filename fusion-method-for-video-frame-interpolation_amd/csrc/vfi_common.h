// Internal helpers shared by the translation units of libvfi_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "../../include/vfi_hip.h"

namespace vfi {

void set_error(const char *fmt, ...);

inline int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
inline int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    set_error("%s", buf);
    return code;
}

// Call after a kernel launch: reports launch-configuration errors without synchronising.
inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(VFI_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return VFI_OK;
}

inline hipStream_t as_stream(vfi_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int kWave = 64;  // gfx950 wavefront

// Per-device one-time setup (hipFuncSetAttribute applies to the current device only; the CU count is a device
// property): caches are indexed by the calling thread's current device, so a process that drives several GPUs
// sets every one of them up.  Values are idempotent: racing threads store the same thing.
constexpr int kMaxDevices = 64;
inline int current_device() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices) d = 0;
    return d;
}

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

}  // namespace vfi

#define VFI_REQUIRE(cond, code, ...)                      \
    do {                                                  \
        if (!(cond)) return vfi::fail((code), __VA_ARGS__); \
    } while (0)
