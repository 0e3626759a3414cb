// Homogeneous-pose and camera-geometry utilities of isaaclab.utils.math: make_pose / pose_inv / pose_in_A_to_pose_in_B
// (math.py:1564-1643), copysign (:121-135), is_identity_pose (:728-746), sample_cylinder (:1387-1420),
// orthogonalize_perspective_depth / unproject_depth / project_points (:991-1225).  One thread per matrix, pixel or point.
#include "common.h"
#include "math_dev.h"
#include "philox.h"

#include "../../include/isaaclab_amd.h"

namespace ilx {

struct PoseArgs {
    int op;
    const float* a;   // per op, see the header
    const float* b;
    int64_t a_batch, b_batch;  // 1: the operand is shared by all n
    float scalar;
    float* out;
    int32_t* flag;    // is_identity_pose: set to 1 by any element that is not close
    int64_t n;
};

__global__ void __launch_bounds__(256) pose_op_kernel(PoseArgs s) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= s.n) return;
    const int64_t ia = s.a_batch == 1 ? 0 : i, ib = s.b_batch == 1 ? 0 : i;
    switch (s.op) {
        case ILX_POSE_MAKE: {  // a: (n,3) positions, b: (n,3,3) rotations
            const float* p = s.a + 3 * ia;
            const float* r = s.b + 9 * ib;
            float* o = s.out + 16 * i;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                o[4 * k] = r[3 * k], o[4 * k + 1] = r[3 * k + 1], o[4 * k + 2] = r[3 * k + 2], o[4 * k + 3] = p[k];
            }
            o[12] = o[13] = o[14] = 0.0f, o[15] = 1.0f;
            break;
        }
        case ILX_POSE_INV: {  // [R t; 0 1]^-1 = [R^T  -R^T t; 0 1]
            const float* m = s.a + 16 * ia;
            float* o = s.out + 16 * i;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float x = -m[r], y = -m[4 + r], z = -m[8 + r];  // row r of -R^T
                o[4 * r] = m[r], o[4 * r + 1] = m[4 + r], o[4 * r + 2] = m[8 + r];
                o[4 * r + 3] = (x * m[3] + y * m[7]) + z * m[11];
            }
            o[12] = o[13] = o[14] = 0.0f, o[15] = 1.0f;
            break;
        }
        case ILX_POSE_MUL: {  // out = a @ b for 4 x 4 matrices
            const float* x = s.a + 16 * ia;
            const float* y = s.b + 16 * ib;
            float* o = s.out + 16 * i;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    o[4 * r + c] = ((x[4 * r] * y[c] + x[4 * r + 1] * y[4 + c]) + x[4 * r + 2] * y[8 + c]) + x[4 * r + 3] * y[12 + c];
            break;
        }
        case ILX_POSE_COPYSIGN: {  // |mag| * sign(other)
            s.out[i] = fabsf(s.scalar) * sign_(s.a[i]);
            break;
        }
        case ILX_POSE_IS_IDENTITY: {  // a: (n,3), b: (n,4); torch.allclose defaults: |x - ref| <= 1e-8 + 1e-5 |ref|
            const float* p = s.a + 3 * i;
            const float* q = s.b + 4 * i;
            bool ok = fabsf(p[0]) <= 1.0e-8f && fabsf(p[1]) <= 1.0e-8f && fabsf(p[2]) <= 1.0e-8f;
            ok = ok && fabsf(q[0] - 1.0f) <= 1.0e-8f + 1.0e-5f && fabsf(q[1]) <= 1.0e-8f && fabsf(q[2]) <= 1.0e-8f && fabsf(q[3]) <= 1.0e-8f;
            if (!ok) *s.flag = 1;  // NaN fails every comparison above, as in torch.allclose
            break;
        }
    }
}

struct CameraArgs {
    int op;
    const float* data;        // depth (n, h, w) or points (n, p, 3)
    const float* intrinsics;  // (k, 3, 3), k = 1 or n
    int64_t k_batch;
    float* out;
    int64_t n;
    int32_t h, w;             // image size; project: h = points per batch, w unused
};

__global__ void __launch_bounds__(256) camera_op_kernel(CameraArgs s) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t per = s.op == ILX_CAMERA_PROJECT ? s.h : static_cast<int64_t>(s.h) * s.w;
    if (i >= s.n * per) return;
    const int64_t img = i / per, px = i - img * per;
    const float* K = s.intrinsics + 9 * (s.k_batch == 1 ? 0 : img);
    if (s.op == ILX_CAMERA_ORTHOGONALIZE) {  // math.py:1046-1069; pixel px = v * w + u
        const int v = static_cast<int>(px / s.w), u = static_cast<int>(px - static_cast<int64_t>(v) * s.w);
        const float xt = div_(static_cast<float>(u) - K[2], K[0]), yt = div_(static_cast<float>(v) - K[5], K[4]);
        s.out[i] = div_(s.data[i], sqrt_((1.0f + xt * xt) + yt * yt));
    } else if (s.op == ILX_CAMERA_UNPROJECT) {  // math.py:1144-1160; output point px = u * h + v (column-major pixel order)
        const int u = static_cast<int>(px / s.h), v = static_cast<int>(px - static_cast<int64_t>(u) * s.h);
        // inverse of the calibration matrix by cofactors, in double (the reference runs an LU in float)
        double m[9], inv[9];
        for (int k = 0; k < 9; ++k) m[k] = K[k];
        const double c0 = m[4] * m[8] - m[5] * m[7], c1 = m[5] * m[6] - m[3] * m[8], c2 = m[3] * m[7] - m[4] * m[6];
        const double det = m[0] * c0 + m[1] * c1 + m[2] * c2;
        inv[0] = c0 / det, inv[1] = (m[2] * m[7] - m[1] * m[8]) / det, inv[2] = (m[1] * m[5] - m[2] * m[4]) / det;
        inv[3] = c1 / det, inv[4] = (m[0] * m[8] - m[2] * m[6]) / det, inv[5] = (m[2] * m[3] - m[0] * m[5]) / det;
        inv[6] = c2 / det, inv[7] = (m[1] * m[6] - m[0] * m[7]) / det, inv[8] = (m[0] * m[4] - m[1] * m[3]) / det;
        const float fu = static_cast<float>(u), fv = static_cast<float>(v);
        float p[3];
        for (int r = 0; r < 3; ++r) {
            const float a = static_cast<float>(inv[3 * r]), b = static_cast<float>(inv[3 * r + 1]), c = static_cast<float>(inv[3 * r + 2]);
            p[r] = (a * fu + b * fv) + c;
        }
        const float depth = s.data[(img * s.h + v) * s.w + u];
        float* o = s.out + 3 * i;
        o[0] = div_(p[0], p[2]) * depth, o[1] = div_(p[1], p[2]) * depth, o[2] = div_(p[2], p[2]) * depth;
    } else {  // project, math.py:1214-1219
        const float* X = s.data + 3 * i;
        float p[3];
        for (int r = 0; r < 3; ++r) p[r] = (K[3 * r] * X[0] + K[3 * r + 1] * X[1]) + K[3 * r + 2] * X[2];
        float* o = s.out + 3 * i;
        o[0] = div_(p[0], p[2]), o[1] = div_(p[1], p[2]), o[2] = X[2];
    }
}

// sample_cylinder, math.py:1387-1420: angle = (u0 * 2 - 1) * pi, z = u1 * (h_max - h_min) + h_min
__global__ void __launch_bounds__(256) sample_cylinder_kernel(float radius, float h_min, float h_max, const float* u, uint64_t seed,
                                                              uint64_t draw, float* out, int64_t n) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float u0, u1;
    if (u) {
        u0 = u[i], u1 = u[n + i];
    } else {
        const u4 r = philox4(seed, draw, static_cast<uint64_t>(i), 0u);
        u0 = u01(r.x), u1 = u01(r.y);
    }
    const float angle = (u0 * 2.0f - 1.0f) * kPi;
    out[3 * i] = radius * cosf(angle);
    out[3 * i + 1] = radius * sinf(angle);
    out[3 * i + 2] = u1 * (h_max - h_min) + h_min;
}

}  // namespace ilx

using namespace ilx;

extern "C" int ilx_pose_op(int32_t op, const float* a, int64_t a_batch, const float* b, int64_t b_batch, float scalar, float* out,
                           int32_t* flag, int64_t n, void* stream) {
    ILX_REQUIRE(op >= ILX_POSE_MAKE && op <= ILX_POSE_IS_IDENTITY, ILX_EINVAL, "ilx_pose_op: unknown op %d", op);
    ILX_REQUIRE(n >= 0, ILX_EINVAL, "ilx_pose_op: negative count");
    if (n == 0) return ILX_OK;
    const bool two = op == ILX_POSE_MAKE || op == ILX_POSE_MUL || op == ILX_POSE_IS_IDENTITY;
    ILX_REQUIRE(a && (b || !two), ILX_EINVAL, "ilx_pose_op: NULL operand");
    ILX_REQUIRE(op == ILX_POSE_IS_IDENTITY ? flag != nullptr : out != nullptr, ILX_EINVAL, "ilx_pose_op: NULL output");
    ILX_REQUIRE((a_batch == 1 || a_batch == n) && (!two || b_batch == 1 || b_batch == n), ILX_ESHAPE,
                "ilx_pose_op: operand batches (%lld, %lld) are neither 1 nor %lld", static_cast<long long>(a_batch),
                static_cast<long long>(b_batch), static_cast<long long>(n));
    PoseArgs s{op, a, b, a_batch, b_batch, scalar, out, flag, n};
    ILX_LAUNCH(pose_op_kernel, dim3(blocks_for(n)), dim3(256), 0, as_stream(stream), s);
    ILX_LAUNCH_CHECK("ilx_pose_op");
    return ILX_OK;
}

extern "C" int ilx_camera_op(int32_t op, const float* data, const float* intrinsics, int64_t intrinsics_batch, float* out, int64_t n,
                             int32_t h, int32_t w, void* stream) {
    ILX_REQUIRE(op >= ILX_CAMERA_ORTHOGONALIZE && op <= ILX_CAMERA_PROJECT, ILX_EINVAL, "ilx_camera_op: unknown op %d", op);
    ILX_REQUIRE(n >= 0 && h >= 0 && w >= 0, ILX_EINVAL, "ilx_camera_op: negative shape");
    const int64_t per = op == ILX_CAMERA_PROJECT ? h : static_cast<int64_t>(h) * w;
    if (n == 0 || per == 0) return ILX_OK;
    ILX_REQUIRE(data && intrinsics && out, ILX_EINVAL, "ilx_camera_op: NULL operand");
    ILX_REQUIRE(intrinsics_batch == 1 || intrinsics_batch == n, ILX_ESHAPE, "ilx_camera_op: %lld calibration matrices for %lld images",
                static_cast<long long>(intrinsics_batch), static_cast<long long>(n));
    CameraArgs s{op, data, intrinsics, intrinsics_batch, out, n, h, w};
    ILX_LAUNCH(camera_op_kernel, dim3(blocks_for(n * per)), dim3(256), 0, as_stream(stream), s);
    ILX_LAUNCH_CHECK("ilx_camera_op");
    return ILX_OK;
}

extern "C" int ilx_sample_cylinder(float radius, float h_min, float h_max, const float* u, uint64_t seed, uint64_t draw, float* out,
                                   int64_t n, void* stream) {
    ILX_REQUIRE(n >= 0, ILX_EINVAL, "ilx_sample_cylinder: negative count");
    if (n == 0) return ILX_OK;
    ILX_REQUIRE(out, ILX_EINVAL, "ilx_sample_cylinder: out is NULL");
    ILX_LAUNCH(sample_cylinder_kernel, dim3(blocks_for(n)), dim3(256), 0, as_stream(stream), radius, h_min, h_max, u, seed, draw,
                       out, n);
    ILX_LAUNCH_CHECK("ilx_sample_cylinder");
    return ILX_OK;
}
