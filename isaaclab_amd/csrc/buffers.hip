// Batched ring buffers (reference: isaaclab/utils/buffers/circular_buffer.py and delay_buffer.py): one launch appends a
// (n, d) block, hands out the ordered history (oldest first) and / or the entry a per-env number of pushes back.
// Layout as in the reference: (max_len, n, d); one thread per (env, column).
#include "common.h"

#include "../../include/isaaclab_amd.h"

namespace ilx {

struct RingArgs {
    ilx_circular_buffer cb;
    const float* data;      // (n, d) rows data_stride apart, or null: nothing is appended
    int64_t data_stride;
    float* ordered;         // (n, max_len * d) rows ordered_stride apart, or null
    int64_t ordered_stride;
    const int32_t* lags;    // (n) or null
    float* delayed;         // (n, d) rows delayed_stride apart, or null
    int64_t delayed_stride;
};

__global__ void __launch_bounds__(256) ring_append_kernel(RingArgs a) {
    const ilx_circular_buffer& cb = a.cb;
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= cb.n * cb.d) return;
    const int64_t env = i / cb.d;
    const int col = static_cast<int>(i - env * cb.d);
    const int64_t pushes = cb.num_pushes[env];   // the count before this append; it is advanced by a launch of its own
    const int64_t slot_stride = cb.n * cb.d;
    float* ring = cb.buffer + env * cb.d + col;
    const bool append = a.data != nullptr;
    const bool fill = append && pushes == 0;     // circular_buffer.py:129-133: the first push after a reset fills every slot
    float x = 0.0f;
    if (append) {
        x = a.data[env * a.data_stride + col];
        if (fill)
            for (int k = 0; k < cb.max_len; ++k) ring[k * slot_stride] = x;
        else
            ring[cb.pointer * slot_stride] = x;
    }
    if (a.ordered) {  // circular_buffer.py:78-86: oldest entry first
        float* o = a.ordered + env * a.ordered_stride + col;
        for (int k = 0; k < cb.max_len; ++k) {
            int slot = cb.pointer + 1 + k;
            if (slot >= cb.max_len) slot -= cb.max_len;
            o[k * cb.d] = (fill || (append && slot == cb.pointer)) ? x : ring[slot * slot_stride];
        }
    }
    if (a.delayed) {  // circular_buffer.py:161-166: a lag beyond the pushes since the reset reads the oldest entry
        const int64_t newest = pushes + (append ? 1 : 0) - 1;
        int64_t lag = a.lags ? a.lags[env] : 0;
        lag = lag < newest ? lag : newest;
        if (lag < 0) lag = 0;
        int slot = static_cast<int>((cb.pointer - lag) % cb.max_len);
        if (slot < 0) slot += cb.max_len;
        a.delayed[env * a.delayed_stride + col] = (fill || (append && slot == cb.pointer)) ? x : ring[slot * slot_stride];
    }
}

__global__ void __launch_bounds__(256) ring_count_kernel(int64_t* num_pushes, int64_t n) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) num_pushes[i] += 1;
}

// circular_buffer.py:92-105
__global__ void __launch_bounds__(256) ring_reset_kernel(ilx_circular_buffer cb, const int64_t* env_ids, int64_t count) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= count * cb.d) return;
    const int64_t k = i / cb.d;
    const int col = static_cast<int>(i - k * cb.d);
    const int64_t env = env_ids ? env_ids[k] : k;
    if (col == 0) cb.num_pushes[env] = 0;
    if (cb.buffer)
        for (int s = 0; s < cb.max_len; ++s) cb.buffer[(s * cb.n + env) * cb.d + col] = 0.0f;
}

static int check_ring(const ilx_circular_buffer& cb, const char* who) {
    ILX_REQUIRE(cb.n >= 0 && cb.d >= 0, ILX_EINVAL, "%s: negative shape", who);
    ILX_REQUIRE(cb.max_len >= 1, ILX_EINVAL, "The buffer size should be greater than zero. However, it is set to %d!", cb.max_len);
    ILX_REQUIRE(cb.n == 0 || cb.num_pushes, ILX_EINVAL, "%s: num_pushes is NULL", who);
    return ILX_OK;
}

}  // namespace ilx

using namespace ilx;

extern "C" int ilx_circular_buffer_append(const ilx_circular_buffer* ring, const float* data, int64_t data_stride, float* ordered_out,
                                          int64_t ordered_stride, const int32_t* time_lags, float* delayed_out, int64_t delayed_stride,
                                          void* stream) {
    ILX_REQUIRE(ring, ILX_EINVAL, "ilx_circular_buffer_append: descriptor is NULL");
    const ilx_circular_buffer& cb = *ring;
    if (const int s = check_ring(cb, "ilx_circular_buffer_append")) return s;
    if (cb.n == 0 || cb.d == 0) return ILX_OK;
    ILX_REQUIRE(cb.buffer, ILX_EINVAL, "ilx_circular_buffer_append: buffer is NULL");
    ILX_REQUIRE(cb.pointer >= 0 && cb.pointer < cb.max_len, ILX_EINVAL, "ilx_circular_buffer_append: pointer %d outside the %d slots",
                cb.pointer, cb.max_len);
    ILX_REQUIRE(!data || data_stride >= cb.d, ILX_ESHAPE, "ilx_circular_buffer_append: data row stride below the width %d", cb.d);
    ILX_REQUIRE(!ordered_out || ordered_stride >= static_cast<int64_t>(cb.max_len) * cb.d, ILX_ESHAPE,
                "ilx_circular_buffer_append: ordered row stride below max_len * width");
    ILX_REQUIRE(!delayed_out || delayed_stride >= cb.d, ILX_ESHAPE, "ilx_circular_buffer_append: delayed row stride below the width %d", cb.d);
    RingArgs a{cb, data, data_stride, ordered_out, ordered_stride, time_lags, delayed_out, delayed_stride};
    ILX_LAUNCH(ring_append_kernel, dim3(blocks_for(cb.n * cb.d)), dim3(256), 0, as_stream(stream), a);
    if (data) ILX_LAUNCH(ring_count_kernel, dim3(blocks_for(cb.n)), dim3(256), 0, as_stream(stream), cb.num_pushes, cb.n);
    ILX_LAUNCH_CHECK("ilx_circular_buffer_append");
    return ILX_OK;
}

extern "C" int ilx_circular_buffer_reset(const ilx_circular_buffer* ring, const int64_t* env_ids, int64_t count, void* stream) {
    ILX_REQUIRE(ring, ILX_EINVAL, "ilx_circular_buffer_reset: descriptor is NULL");
    const ilx_circular_buffer& cb = *ring;
    if (const int s = check_ring(cb, "ilx_circular_buffer_reset")) return s;
    ILX_REQUIRE(count >= 0, ILX_EINVAL, "ilx_circular_buffer_reset: negative count");
    if (cb.n == 0 || count == 0) return ILX_OK;
    ilx_circular_buffer c = cb;
    if (c.d == 0) c.d = 1, c.buffer = nullptr;   // nothing appended yet: only the counts
    ILX_LAUNCH(ring_reset_kernel, dim3(blocks_for(count * c.d)), dim3(256), 0, as_stream(stream), c, env_ids, count);
    ILX_LAUNCH_CHECK("ilx_circular_buffer_reset");
    return ILX_OK;
}
