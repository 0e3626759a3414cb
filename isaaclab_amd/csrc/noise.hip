// Noise functions of isaaclab.utils.noise (noise_model.py:20-98): constant, uniform and gaussian noise, each as an
// additive, multiplicative or replacing term, plus the per-env additive bias of NoiseModelWithAdditiveBias (:145-182).
// One thread per element; draws are Philox keyed by (seed, call, row, column) or handed in (parity tests replay torch's).
#include "common.h"
#include "math_dev.h"
#include "philox.h"

#include "../../include/isaaclab_amd.h"

namespace ilx {

__global__ void __launch_bounds__(256) noise_kernel(ilx_noise s, int kind, int operation) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= s.n * s.d) return;
    const int64_t k = i / s.d;
    const int col = static_cast<int>(i - k * s.d);
    const int64_t row = s.row_ids ? s.row_ids[k] : k;
    const float a = s.a_cols ? s.a_cols[col] : s.a, b = s.b_cols ? s.b_cols[col] : s.b;
    const float x = s.data[row * s.data_stride + col];
    float y;
    if (kind == ILX_NOISE_CONSTANT) {  // noise_model.py:35-40
        y = operation == ILX_NOISE_ADD ? x + a : (operation == ILX_NOISE_SCALE ? x * a : 0.0f + a);
    } else if (kind == ILX_NOISE_UNIFORM) {  // :63-68; "data + r * range + n_min" groups as (data + r * range) + n_min
        const float u = s.draws ? s.draws[i] : philox_uniform(s.seed, s.draw, static_cast<uint64_t>(row), static_cast<uint32_t>(col));
        const float ur = u * ((s.a_cols || s.b_cols) ? b - a : s.range);
        y = operation == ILX_NOISE_ADD ? (x + ur) + a : (operation == ILX_NOISE_SCALE ? x * (ur + a) : ur + a);
    } else {  // :91-96; "data + mean + std * z" groups as (data + mean) + std * z
        float z;
        if (s.draws) {
            z = s.draws[i];
        } else {  // Box-Muller: columns 2j and 2j + 1 share one Philox call
            const u4 r = philox4(s.seed, s.draw, static_cast<uint64_t>(row), static_cast<uint32_t>(col >> 1));
            const float radius = sqrt_(-2.0f * logf(1.0f - u01(r.x))), angle = 6.2831853f * u01(r.y);
            z = (col & 1) ? radius * sinf(angle) : radius * cosf(angle);
        }
        const float sz = b * z;
        y = operation == ILX_NOISE_ADD ? (x + a) + sz : (operation == ILX_NOISE_SCALE ? x * (a + sz) : a + sz);
    }
    if (s.row_bias) y += s.row_bias[row];
    s.out[row * s.out_stride + col] = y;
}

}  // namespace ilx

using namespace ilx;

extern "C" int ilx_noise_apply(int32_t kind, int32_t operation, const ilx_noise* noise, void* stream) {
    ILX_REQUIRE(noise, ILX_EINVAL, "ilx_noise_apply: descriptor is NULL");
    const ilx_noise& s = *noise;
    ILX_REQUIRE(kind >= ILX_NOISE_CONSTANT && kind <= ILX_NOISE_GAUSSIAN, ILX_EINVAL, "ilx_noise_apply: unknown kind %d", kind);
    ILX_REQUIRE(operation >= ILX_NOISE_ADD && operation <= ILX_NOISE_ABS, ILX_EINVAL, "Unknown operation in noise: %d", operation);
    ILX_REQUIRE(s.n >= 0 && s.d >= 0, ILX_EINVAL, "ilx_noise_apply: negative shape");
    if (s.n == 0 || s.d == 0) return ILX_OK;
    ILX_REQUIRE(s.data && s.out, ILX_EINVAL, "ilx_noise_apply: data or out is NULL");
    ILX_REQUIRE(s.data_stride >= s.d && s.out_stride >= s.d, ILX_ESHAPE, "ilx_noise_apply: row strides (%lld, %lld) below the width %d",
                static_cast<long long>(s.data_stride), static_cast<long long>(s.out_stride), s.d);
    ILX_LAUNCH(noise_kernel, dim3(blocks_for(s.n * s.d)), dim3(256), 0, as_stream(stream), s, kind, operation);
    ILX_LAUNCH_CHECK("ilx_noise_apply");
    return ILX_OK;
}
