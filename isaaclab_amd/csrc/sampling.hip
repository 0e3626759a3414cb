// Samplers of isaaclab.utils.math (math.py:1225-1420): uniform, log-uniform, triangular and gaussian values, random
// orientations.  The reference draws with torch.rand / torch.randn; here the draws are Philox, keyed by (seed, call,
// row, column) so that a sample does not depend on the launch geometry, or are handed in (parity tests replay torch's).
#include "common.h"
#include "math_dev.h"
#include "philox.h"

namespace ilx {

struct SampleArgs {
    int kind;
    const float* a;  // lower / mean: scalar (stride 0, 0), per column (0, 1) or per element (width, 1)
    const float* b;  // upper / std
    int64_t a_row, a_col, b_row, b_col;
    float a_scalar, b_scalar;
    const float* u;  // (rows, width) unit-interval draws, or null
    const float* z;  // (rows, width) standard normal draws, or null
    uint64_t seed, draw;
    float* out;
    int64_t rows;
    int width;
};

ILX_DEV float normal_draw(const SampleArgs& s, int64_t row, int col) {
    if (s.z) return s.z[row * s.width + col];
    // Box-Muller: columns 2k and 2k + 1 share one Philox call
    const u4 r = philox4(s.seed, s.draw, static_cast<uint64_t>(row), static_cast<uint32_t>(col >> 1));
    const float radius = sqrt_(-2.0f * logf(1.0f - u01(r.x))), angle = 6.2831853f * u01(r.y);
    return (col & 1) ? radius * sinf(angle) : radius * cosf(angle);
}

__global__ void __launch_bounds__(256) sample_kernel(SampleArgs s) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (s.kind == ILX_SAMPLE_RANDOM_ORIENTATION || s.kind == ILX_SAMPLE_RANDOM_YAW_ORIENTATION) {  // one thread per quaternion
        if (i >= s.rows) return;
        f4 q;
        if (s.kind == ILX_SAMPLE_RANDOM_ORIENTATION) {  // math.py:1251-1267
            const float g0 = normal_draw(s, i, 0), g1 = normal_draw(s, i, 1), g2 = normal_draw(s, i, 2), g3 = normal_draw(s, i, 3);
            const float nq = fmaxf(norm4(f4{g0, g1, g2, g3}), 1.0e-12f);
            q = {div_(g0, nq), div_(g1, nq), div_(g2, nq), div_(g3, nq)};
        } else {  // math.py:1270-1285
            const float u = s.u ? s.u[i] : philox_uniform(s.seed, s.draw, static_cast<uint64_t>(i), 0u);
            q = quat_from_euler_xyz(0.0f, 0.0f, 2.0f * kPi * u);
        }
        float* o = s.out + 4 * i;
        o[0] = q.w, o[1] = q.x, o[2] = q.y, o[3] = q.z;
        return;
    }
    if (i >= s.rows * s.width) return;
    const int64_t row = i / s.width;
    const int col = static_cast<int>(i - row * s.width);
    const float a = s.a ? s.a[row * s.a_row + col * s.a_col] : s.a_scalar;
    const float b = s.b ? s.b[row * s.b_row + col * s.b_col] : s.b_scalar;
    float v;
    if (s.kind == ILX_SAMPLE_GAUSSIAN) {  // math.py:1371-1390: mean + std * N(0, 1)
        v = normal_draw(s, row, col) * b + a;
    } else {
        const float u = s.u ? s.u[i] : philox_uniform(s.seed, s.draw, static_cast<uint64_t>(row), static_cast<uint32_t>(col));
        if (s.kind == ILX_SAMPLE_UNIFORM) {  // math.py:1313-1331
            v = u * (b - a) + a;
        } else if (s.kind == ILX_SAMPLE_LOG_UNIFORM) {  // math.py:1334-1368
            const float la = logf(a), lb = logf(b);
            v = expf(u * (lb - la) + la);
        } else {  // triangle, math.py:1288-1310
            float r = 2.0f * u - 1.0f;
            r = r < 0.0f ? -sqrt_(-r) : sqrt_(r);
            r = div_(r + 1.0f, 2.0f);
            v = (b - a) * r + a;
        }
    }
    s.out[i] = v;
}

}  // namespace ilx

using namespace ilx;

extern "C" int ilx_sample(int kind, const float* a, int64_t a_row_stride, int64_t a_col_stride, float a_scalar, const float* b,
                          int64_t b_row_stride, int64_t b_col_stride, float b_scalar, const float* u, const float* z, uint64_t seed,
                          uint64_t draw, float* out, int64_t rows, int width, void* stream) {
    ILX_REQUIRE(kind >= ILX_SAMPLE_UNIFORM && kind <= ILX_SAMPLE_RANDOM_YAW_ORIENTATION, ILX_EINVAL, "ilx_sample: unknown kind %d", kind);
    ILX_REQUIRE(rows >= 0 && width > 0, ILX_EINVAL, "ilx_sample: bad shape (%lld, %d)", (long long)rows, width);
    const bool quat = kind == ILX_SAMPLE_RANDOM_ORIENTATION || kind == ILX_SAMPLE_RANDOM_YAW_ORIENTATION;
    ILX_REQUIRE(!quat || width == 4, ILX_ESHAPE, "ilx_sample: orientations are 4 wide, not %d", width);
    if (rows == 0) return ILX_OK;
    ILX_REQUIRE(out, ILX_EINVAL, "ilx_sample: out is NULL");
    SampleArgs s{kind, a, b, a_row_stride, a_col_stride, b_row_stride, b_col_stride, a_scalar, b_scalar, u, z, seed, draw, out, rows, width};
    const int64_t threads = quat ? rows : rows * width;
    ILX_LAUNCH(sample_kernel, dim3(blocks_for(threads)), dim3(256), 0, as_stream(stream), s);
    ILX_LAUNCH_CHECK("ilx_sample");
    return ILX_OK;
}
