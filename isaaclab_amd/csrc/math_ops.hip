// Row-wise kernels behind the isaaclab.utils.math drop-in functions: one thread per row, one launch
// per function call.  Rows are tiny (1..9 floats) and packed, so a wave reads/writes contiguous
// memory; these launches are HBM-streaming with trivial arithmetic.
#include "common.h"
#include "math_dev.h"

namespace ilx {

template <int OP>
__global__ void __launch_bounds__(256) math_op_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      const float* __restrict__ c, const float* __restrict__ d,
                                                      float* __restrict__ o0, float* __restrict__ o1, int64_t n) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if constexpr (OP == ILX_QUAT_ROTATE) {
        store3(o0 + 3 * i, quat_rotate(load4(a + 4 * i), load3(b + 3 * i)));
    } else if constexpr (OP == ILX_QUAT_ROTATE_INVERSE) {
        store3(o0 + 3 * i, quat_rotate_inverse(load4(a + 4 * i), load3(b + 3 * i)));
    } else if constexpr (OP == ILX_QUAT_APPLY) {
        store3(o0 + 3 * i, quat_apply(load4(a + 4 * i), load3(b + 3 * i)));
    } else if constexpr (OP == ILX_QUAT_APPLY_YAW) {
        store3(o0 + 3 * i, quat_apply_yaw(load4(a + 4 * i), load3(b + 3 * i)));
    } else if constexpr (OP == ILX_YAW_QUAT) {
        store4(o0 + 4 * i, yaw_quat(load4(a + 4 * i)));
    } else if constexpr (OP == ILX_QUAT_MUL) {
        store4(o0 + 4 * i, quat_mul(load4(a + 4 * i), load4(b + 4 * i)));
    } else if constexpr (OP == ILX_QUAT_CONJUGATE) {
        store4(o0 + 4 * i, quat_conjugate(load4(a + 4 * i)));
    } else if constexpr (OP == ILX_QUAT_INV) {
        store4(o0 + 4 * i, quat_inv(load4(a + 4 * i)));
    } else if constexpr (OP == ILX_QUAT_UNIQUE) {
        store4(o0 + 4 * i, quat_unique(load4(a + 4 * i)));
    } else if constexpr (OP == ILX_NORMALIZE3) {
        store3(o0 + 3 * i, normalize3(load3(a + 3 * i)));
    } else if constexpr (OP == ILX_NORMALIZE4) {
        store4(o0 + 4 * i, normalize4(load4(a + 4 * i)));
    } else if constexpr (OP == ILX_MATRIX_FROM_QUAT) {
        float m[9];
        matrix_from_quat(load4(a + 4 * i), m);
#pragma unroll
        for (int k = 0; k < 9; ++k) o0[9 * i + k] = m[k];
    } else if constexpr (OP == ILX_QUAT_BOX_MINUS) {
        store3(o0 + 3 * i, quat_box_minus(load4(a + 4 * i), load4(b + 4 * i)));
    } else if constexpr (OP == ILX_AXIS_ANGLE_FROM_QUAT) {
        store3(o0 + 3 * i, axis_angle_from_quat(load4(a + 4 * i)));
    } else if constexpr (OP == ILX_QUAT_ERROR_MAGNITUDE) {
        o0[i] = quat_error_magnitude(load4(a + 4 * i), load4(b + 4 * i));
    } else if constexpr (OP == ILX_QUAT_FROM_ANGLE_AXIS) {
        store4(o0 + 4 * i, quat_from_angle_axis(a[i], load3(b + 3 * i)));
    } else if constexpr (OP == ILX_QUAT_FROM_EULER_XYZ) {
        store4(o0 + 4 * i, quat_from_euler_xyz(a[i], b[i], c[i]));
    } else if constexpr (OP == ILX_EULER_XYZ_FROM_QUAT) {
        store3(o0 + 3 * i, euler_xyz_from_quat(load4(a + 4 * i)));
    } else if constexpr (OP == ILX_COMBINE_FRAME) {
        f3 t;
        f4 q;
        combine_frame_transforms(load3(a + 3 * i), load4(b + 4 * i), load3(c + 3 * i), load4(d + 4 * i), t, q);
        store3(o0 + 3 * i, t);
        store4(o1 + 4 * i, q);
    } else if constexpr (OP == ILX_SUBTRACT_FRAME) {
        f3 t;
        f4 q;
        subtract_frame_transforms(load3(a + 3 * i), load4(b + 4 * i), load3(c + 3 * i), load4(d + 4 * i), t, q);
        store3(o0 + 3 * i, t);
        store4(o1 + 4 * i, q);
    } else if constexpr (OP == ILX_POSE_ERROR_AA || OP == ILX_POSE_ERROR_QUAT) {
        f3 t01 = load3(a + 3 * i), t02 = load3(c + 3 * i);
        f4 qe = quat_error(load4(b + 4 * i), load4(d + 4 * i));
        store3(o0 + 3 * i, {t02.x - t01.x, t02.y - t01.y, t02.z - t01.z});
        if constexpr (OP == ILX_POSE_ERROR_AA) {
            store3(o1 + 3 * i, axis_angle_from_quat(qe));
        } else {
            store4(o1 + 4 * i, qe);
        }
    } else if constexpr (OP == ILX_APPLY_DELTA_POSE) {
        f3 p;
        f4 q;
        apply_delta_pose(load3(a + 3 * i), load4(b + 4 * i), c + 6 * i, 1.0e-6f, p, q);
        store3(o0 + 3 * i, p);
        store4(o1 + 4 * i, q);
    } else if constexpr (OP == ILX_WRAP_TO_PI) {
        o0[i] = wrap_to_pi(a[i]);
    } else if constexpr (OP == ILX_SKEW) {
        f3 v = load3(a + 3 * i);
        float* m = o0 + 9 * i;
        m[0] = 0.0f, m[1] = -v.z, m[2] = v.y;
        m[3] = v.z, m[4] = 0.0f, m[5] = -v.x;
        m[6] = -v.y, m[7] = v.x, m[8] = 0.0f;
    }
}

template <int MODE>
__global__ void __launch_bounds__(256) range_op_kernel(const float* __restrict__ x, const float* __restrict__ lo,
                                                       const float* __restrict__ hi, int per_row,
                                                       float* __restrict__ out, int64_t total, int w) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t bi = per_row ? i : (i % w);
    const float l = lo[bi], u = hi[bi], v = x[i];
    if constexpr (MODE == 0) {  // math.py:23-40
        const float off = (l + u) * 0.5f;
        out[i] = div_(2.0f * (v - off), u - l);
    } else if constexpr (MODE == 1) {  // math.py:44-61
        const float off = (l + u) * 0.5f;
        out[i] = v * (u - l) * 0.5f + off;
    } else {  // math.py:65-78
        out[i] = fmaxf(fminf(v, u), l);
    }
}

// operand widths per op: a, b, c, d, out0, out1 (0 = unused)
struct OpShape {
    int a, b, c, d, o0, o1;
};
static const OpShape kShapes[ILX_MATH_OP_COUNT] = {
    {4, 3, 0, 0, 3, 0}, {4, 3, 0, 0, 3, 0}, {4, 3, 0, 0, 3, 0}, {4, 3, 0, 0, 3, 0}, {4, 0, 0, 0, 4, 0},
    {4, 4, 0, 0, 4, 0}, {4, 0, 0, 0, 4, 0}, {4, 0, 0, 0, 4, 0}, {4, 0, 0, 0, 4, 0}, {3, 0, 0, 0, 3, 0},
    {4, 0, 0, 0, 4, 0}, {4, 0, 0, 0, 9, 0}, {4, 4, 0, 0, 3, 0}, {4, 0, 0, 0, 3, 0}, {4, 4, 0, 0, 1, 0},
    {1, 3, 0, 0, 4, 0}, {1, 1, 1, 0, 4, 0}, {4, 0, 0, 0, 3, 0}, {3, 4, 3, 4, 3, 4}, {3, 4, 3, 4, 3, 4},
    {3, 4, 3, 4, 3, 3}, {3, 4, 3, 4, 3, 4}, {3, 4, 6, 0, 3, 4}, {1, 0, 0, 0, 1, 0}, {3, 0, 0, 0, 9, 0},
};

template <int OP>
static void launch(const float* a, const float* b, const float* c, const float* d, float* o0, float* o1, int64_t n,
                   hipStream_t s) {
    ILX_LAUNCH(math_op_kernel<OP>, dim3(blocks_for(n)), dim3(256), 0, s, a, b, c, d, o0, o1, n);
}

}  // namespace ilx

using namespace ilx;

extern "C" int ilx_math_op(int op, const float* a, const float* b, const float* c, const float* d, float* out0,
                           float* out1, int64_t n, void* stream) {
    ILX_REQUIRE(op >= 0 && op < ILX_MATH_OP_COUNT, ILX_EINVAL, "ilx_math_op: unknown op %d", op);
    ILX_REQUIRE(n >= 0, ILX_EINVAL, "ilx_math_op: negative row count");
    if (n == 0) return ILX_OK;
    const OpShape& sh = kShapes[op];
    ILX_REQUIRE((sh.a == 0 || a) && (sh.b == 0 || b) && (sh.c == 0 || c) && (sh.d == 0 || d) && (sh.o0 == 0 || out0) &&
                    (sh.o1 == 0 || out1),
                ILX_EINVAL, "ilx_math_op(%d): a required operand is NULL", op);
    // 4-wide operands are moved as float4
    auto aligned = [](const void* p, int w) { return w != 4 || (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    ILX_REQUIRE(aligned(a, sh.a) && aligned(b, sh.b) && aligned(d, sh.d) && aligned(out0, sh.o0) && aligned(out1, sh.o1),
                ILX_EINVAL, "ilx_math_op(%d): quaternion operands must be 16-byte aligned", op);
    hipStream_t s = as_stream(stream);
    switch (op) {
#define ILX_CASE(OP) \
    case OP:         \
        launch<OP>(a, b, c, d, out0, out1, n, s); \
        break;
        ILX_CASE(ILX_QUAT_ROTATE)
        ILX_CASE(ILX_QUAT_ROTATE_INVERSE)
        ILX_CASE(ILX_QUAT_APPLY)
        ILX_CASE(ILX_QUAT_APPLY_YAW)
        ILX_CASE(ILX_YAW_QUAT)
        ILX_CASE(ILX_QUAT_MUL)
        ILX_CASE(ILX_QUAT_CONJUGATE)
        ILX_CASE(ILX_QUAT_INV)
        ILX_CASE(ILX_QUAT_UNIQUE)
        ILX_CASE(ILX_NORMALIZE3)
        ILX_CASE(ILX_NORMALIZE4)
        ILX_CASE(ILX_MATRIX_FROM_QUAT)
        ILX_CASE(ILX_QUAT_BOX_MINUS)
        ILX_CASE(ILX_AXIS_ANGLE_FROM_QUAT)
        ILX_CASE(ILX_QUAT_ERROR_MAGNITUDE)
        ILX_CASE(ILX_QUAT_FROM_ANGLE_AXIS)
        ILX_CASE(ILX_QUAT_FROM_EULER_XYZ)
        ILX_CASE(ILX_EULER_XYZ_FROM_QUAT)
        ILX_CASE(ILX_COMBINE_FRAME)
        ILX_CASE(ILX_SUBTRACT_FRAME)
        ILX_CASE(ILX_POSE_ERROR_AA)
        ILX_CASE(ILX_POSE_ERROR_QUAT)
        ILX_CASE(ILX_APPLY_DELTA_POSE)
        ILX_CASE(ILX_WRAP_TO_PI)
        ILX_CASE(ILX_SKEW)
#undef ILX_CASE
    }
    ILX_LAUNCH_CHECK("ilx_math_op");
    return ILX_OK;
}

extern "C" int ilx_math_range_op(int mode, const float* x, const float* lower, const float* upper, int bounds_per_row,
                                 float* out, int64_t n, int w, void* stream) {
    ILX_REQUIRE(mode >= 0 && mode <= 2, ILX_EINVAL, "ilx_math_range_op: unknown mode %d", mode);
    ILX_REQUIRE(n >= 0 && w > 0, ILX_EINVAL, "ilx_math_range_op: bad shape (%lld, %d)", (long long)n, w);
    ILX_REQUIRE(x && lower && upper && out, ILX_EINVAL, "ilx_math_range_op: NULL operand");
    const int64_t total = n * w;
    if (total == 0) return ILX_OK;
    hipStream_t s = as_stream(stream);
    dim3 g(blocks_for(total)), b(256);
    if (mode == 0) ILX_LAUNCH(range_op_kernel<0>, g, b, 0, s, x, lower, upper, bounds_per_row, out, total, w);
    if (mode == 1) ILX_LAUNCH(range_op_kernel<1>, g, b, 0, s, x, lower, upper, bounds_per_row, out, total, w);
    if (mode == 2) ILX_LAUNCH(range_op_kernel<2>, g, b, 0, s, x, lower, upper, bounds_per_row, out, total, w);
    ILX_LAUNCH_CHECK("ilx_math_range_op");
    return ILX_OK;
}
