// Shared host-side helpers for the C-ABI library: error reporting and launch geometry.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <tuple>
#include <utility>
#include <vector>

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

// ---- launches and command lists (ilx_commands_*) -----------------------------------------------------------------
// Every kernel of the library is launched through ILX_LAUNCH.  While the calling thread records a command list the launch
// is kept -- kernel, geometry and a copy of its arguments -- instead of issued; ilx_commands_replay issues the kept
// launches in order with one hipLaunchKernel each.
struct Command {
    const void* kernel;
    dim3 grid, block;
    unsigned shmem;
    std::vector<char> blob;      // the arguments, each at its own alignment
    std::vector<void*> args;     // pointers into the blob, in parameter order
    Command(const Command&) = delete;             // (a copy's pointers would name the original's blob;
    Command(Command&&) noexcept = default;        //  a moved vector keeps its storage)
    Command& operator=(Command&&) noexcept = default;
    Command(const void* k, dim3 g, dim3 b, unsigned s) : kernel(k), grid(g), block(b), shmem(s) {}
};
struct CommandList {
    std::vector<Command> commands;
    hipStream_t stream = nullptr;   // the stream the recorded calls named
    bool any = false, mixed = false, open = true;
};
CommandList*& recording_list();     // thread-local; null: launches are issued

template <typename... P, typename... A>
inline void launch(void (*kernel)(P...), dim3 grid, dim3 block, unsigned shmem, hipStream_t stream, A&&... a) {
    static_assert(sizeof...(P) == sizeof...(A), "one argument per kernel parameter");
    std::tuple<P...> params{static_cast<P>(std::forward<A>(a))...};   // converted to the parameter types, as a call would
    CommandList* list = recording_list();
    std::apply(
        [&](auto&... p) {
            if (!list) {
                void* ptrs[sizeof...(P) + 1] = {static_cast<void*>(&p)...};
                (void)hipLaunchKernel(reinterpret_cast<const void*>(kernel), grid, block, ptrs, shmem, stream);
                return;
            }
            if (list->any && stream != list->stream) list->mixed = true;
            list->any = true, list->stream = stream;
            Command c(reinterpret_cast<const void*>(kernel), grid, block, shmem);
            size_t at = 0, offsets[sizeof...(P) + 1] = {};
            int k = 0;
            ((at = (at + alignof(P) - 1) / alignof(P) * alignof(P), offsets[k++] = at, at += sizeof(P)), ...);
            c.blob.resize(at + 1);
            k = 0;
            ((std::memcpy(c.blob.data() + offsets[k], &p, sizeof(P)), c.args.push_back(c.blob.data() + offsets[k]), ++k), ...);
            list->commands.push_back(std::move(c));   // (a moved vector keeps its storage: the pointers stay good)
        },
        params);
}

}  // namespace ilx

#define ILX_LAUNCH(kernel, grid, block, shmem, stream, ...) ilx::launch(kernel, grid, block, shmem, stream, ##__VA_ARGS__)

#define ILX_REQUIRE(cond, code, ...)                 \
    do {                                             \
        if (!(cond)) return ilx::fail(code, __VA_ARGS__); \
    } while (0)

#define ILX_LAUNCH_CHECK(name)                                     \
    do {                                                           \
        hipError_t e__ = hipGetLastError();                        \
        if (e__ != hipSuccess) return ilx::check_hip(e__, name);   \
    } while (0)
