// Shared host-side helpers for the C-ABI library: error reporting and launch geometry.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/isaaclab_amd.h"

namespace ilx {

// Thread-local message returned by ilx_last_error().
char* error_buffer();
int fail(int code, const char* fmt, ...);

inline int check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return ILX_OK;
    return fail(ILX_EHIP, "%s: %s", what, hipGetErrorString(e));
}

// One thread per item, 256-thread blocks.
inline unsigned blocks_for(int64_t items, int threads = 256) {
    return static_cast<unsigned>((items + threads - 1) / threads);
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Compute units of the current device (asked once per device).
inline int device_cu_count() {
    static int cached[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) dev = 0;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}

}  // namespace ilx

#define ILX_REQUIRE(cond, code, ...)                 \
    do {                                             \
        if (!(cond)) return ilx::fail(code, __VA_ARGS__); \
    } while (0)

#define ILX_LAUNCH_CHECK(name)                                     \
    do {                                                           \
        hipError_t e__ = hipGetLastError();                        \
        if (e__ != hipSuccess) return ilx::check_hip(e__, name);   \
    } while (0)
