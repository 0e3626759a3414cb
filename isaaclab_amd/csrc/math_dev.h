// Device-side quaternion / frame math with a pinned evaluation order.
//
// Every translation unit that includes this header is compiled with -ffp-contract=off, so a*b+c is
// two roundings unless written as fmaf_().  The order of operations per function is the order the
// reference's isaaclab/utils/math.py evaluates on torch CPU (cited per function), which is what makes
// termination / contact decisions bit-identical to the CPU oracle:
//   * cross products: each component is fma(a1, b2, -(a2*b1))          (torch.cross)
//   * 2-/3-vector norms: sqrt of an ascending fma chain                  (torch.norm)
//   * 4-vector norms, 3-term dots, sums of squares: left to right, unfused (torch.sum / bmm)
// Division and sqrt are the correctly rounded forms (hipcc default for HIP).
#pragma once

#include <hip/hip_runtime.h>

namespace ilx {

struct f3 {
    float x, y, z;
};
struct f4 {
    float w, x, y, z;
};

#define ILX_DEV __device__ __forceinline__

ILX_DEV float fmaf_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// __fsqrt_rn lowers to a bare v_sqrt_f32 (1 ulp); __builtin_sqrtf gets the correctly rounded fix-up sequence.
ILX_DEV float sqrt_(float x) { return __builtin_sqrtf(x); }
ILX_DEV float div_(float a, float b) { return __fdiv_rn(a, b); }

constexpr float kPi = 3.14159265358979323846f;      // fp32(pi)
constexpr float kTwoPi = 6.28318530717958647692f;   // fp32(2*pi)
constexpr float kHalfPi = 1.57079632679489661923f;  // fp32(pi/2)

ILX_DEV f3 cross3(f3 a, f3 b) {
    return {fmaf_(a.y, b.z, -(a.z * b.y)), fmaf_(a.z, b.x, -(a.x * b.z)), fmaf_(a.x, b.y, -(a.y * b.x))};
}
ILX_DEV float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
ILX_DEV float norm2(float x, float y) { return sqrt_(fmaf_(y, y, x * x)); }
ILX_DEV float norm3(float x, float y, float z) { return sqrt_(fmaf_(z, z, fmaf_(y, y, x * x))); }
ILX_DEV float norm3(f3 v) { return norm3(v.x, v.y, v.z); }
ILX_DEV float norm4(f4 q) { return sqrt_(((q.w * q.w + q.x * q.x) + q.y * q.y) + q.z * q.z); }

// torch.remainder for floats: fmod, then shift into the divisor's sign.
ILX_DEV float remainder_(float a, float b) {
    float m = fmodf(a, b);
    if (m != 0.0f && ((b < 0.0f) != (m < 0.0f))) m += b;
    return m;
}
// torch.sign
ILX_DEV float sign_(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : ((x == 0.0f) ? 0.0f : x)); }

// math.py:82-92
ILX_DEV f3 normalize3(f3 v, float eps = 1e-9f) {
    float n = fmaxf(norm3(v), eps);
    return {div_(v.x, n), div_(v.y, n), div_(v.z, n)};
}
ILX_DEV f4 normalize4(f4 q, float eps = 1e-9f) {
    float n = fmaxf(norm4(q), eps);
    return {div_(q.w, n), div_(q.x, n), div_(q.y, n), div_(q.z, n)};
}

// math.py:96-117
ILX_DEV float wrap_to_pi(float a) {
    float w = remainder_(a + kPi, kTwoPi);
    return (w == 0.0f && a > 0.0f) ? kPi : (w - kPi);
}

// math.py:224-235
ILX_DEV f4 quat_conjugate(f4 q) { return {q.w, -q.x, -q.y, -q.z}; }
// math.py:239-248
ILX_DEV f4 quat_inv(f4 q) { return normalize4(quat_conjugate(q)); }
// math.py:448-460
ILX_DEV f4 quat_unique(f4 q) { return (q.w < 0.0f) ? f4{-q.w, -q.x, -q.y, -q.z} : q; }

// math.py:464-499 (9-product form)
ILX_DEV f4 quat_mul(f4 a, f4 b) {
    float ww = (a.z + a.x) * (b.x + b.y);
    float yy = (a.w - a.y) * (b.w + b.z);
    float zz = (a.w + a.y) * (b.w - b.z);
    float xx = ww + yy + zz;
    float qq = 0.5f * (xx + (a.z - a.x) * (b.x - b.y));
    f4 r;
    r.w = qq - ww + (a.z - a.y) * (b.y - b.z);
    r.x = qq - xx + (a.x + a.w) * (b.x + b.w);
    r.y = qq - yy + (a.w - a.x) * (b.y + b.z);
    r.z = qq - zz + (a.z + a.y) * (b.w - b.x);
    return r;
}

// math.py:546-564
ILX_DEV f3 quat_apply(f4 q, f3 v) {
    f3 u = {q.x, q.y, q.z};
    f3 t = cross3(u, v);
    t = {t.x * 2.0f, t.y * 2.0f, t.z * 2.0f};
    f3 c = cross3(u, t);
    return {v.x + q.w * t.x + c.x, v.y + q.w * t.y + c.y, v.z + q.w * t.z + c.z};
}

// math.py:522-542
ILX_DEV f4 yaw_quat(f4 q) {
    float yaw = atan2f(2.0f * (q.w * q.z + q.x * q.y), 1.0f - 2.0f * (q.y * q.y + q.z * q.z));
    float h = div_(yaw, 2.0f);
    f4 r = {cosf(h), 0.0f, 0.0f, sinf(h)};
    return normalize4(r);
}
// math.py:568-579
ILX_DEV f3 quat_apply_yaw(f4 q, f3 v) { return quat_apply(yaw_quat(q), v); }

// math.py:583-625
template <bool INVERSE>
ILX_DEV f3 quat_rotate_t(f4 q, f3 v) {
    float k = 2.0f * (q.w * q.w) - 1.0f;
    f3 u = {q.x, q.y, q.z};
    f3 cr = cross3(u, v);
    float d = dot3(u, v);
    f3 a = {v.x * k, v.y * k, v.z * k};
    f3 b = {cr.x * q.w * 2.0f, cr.y * q.w * 2.0f, cr.z * q.w * 2.0f};
    f3 c = {u.x * d * 2.0f, u.y * d * 2.0f, u.z * d * 2.0f};
    if (INVERSE) return {a.x - b.x + c.x, a.y - b.y + c.y, a.z - b.z + c.z};
    return {a.x + b.x + c.x, a.y + b.y + c.y, a.z + b.z + c.z};
}
ILX_DEV f3 quat_rotate(f4 q, f3 v) { return quat_rotate_t<false>(q, v); }
ILX_DEV f3 quat_rotate_inverse(f4 q, f3 v) { return quat_rotate_t<true>(q, v); }

// math.py:629-642
ILX_DEV f4 quat_from_angle_axis(float angle, f3 axis) {
    float th = div_(angle, 2.0f);
    f3 n = normalize3(axis);
    float s = sinf(th), c = cosf(th);
    return normalize4({c, n.x * s, n.y * s, n.z * s});
}

// math.py:646-674
ILX_DEV f3 axis_angle_from_quat(f4 q, float eps = 1.0e-6f) {
    float sgn = 1.0f - 2.0f * ((q.w < 0.0f) ? 1.0f : 0.0f);
    q = {q.w * sgn, q.x * sgn, q.y * sgn, q.z * sgn};
    float mag = norm3(q.x, q.y, q.z);
    float half = atan2f(mag, q.w);
    float angle = 2.0f * half;
    float den = (fabsf(angle) > eps) ? div_(sinf(half), angle) : (0.5f - div_(angle * angle, 48.0f));
    return {div_(q.x, den), div_(q.y, den), div_(q.z, den)};
}

// math.py:678-689
ILX_DEV float quat_error_magnitude(f4 a, f4 b) { return norm3(axis_angle_from_quat(quat_mul(a, quat_conjugate(b)))); }

// math.py:503-518
ILX_DEV f3 quat_box_minus(f4 a, f4 b) {
    f4 d = quat_mul(a, quat_conjugate(b));
    float n = norm3(d.x, d.y, d.z);
    float s = 2.0f * ((n > 1.0e-7f) ? div_(atan2f(n, d.w), n) : sign_(d.w));
    return {s * d.x, s * d.y, s * d.z};
}

// math.py:252-278
ILX_DEV f4 quat_from_euler_xyz(float roll, float pitch, float yaw) {
    float cy = cosf(yaw * 0.5f), sy = sinf(yaw * 0.5f);
    float cr = cosf(roll * 0.5f), sr = sinf(roll * 0.5f);
    float cp = cosf(pitch * 0.5f), sp = sinf(pitch * 0.5f);
    return {cy * cr * cp + sy * sr * sp, cy * sr * cp - sy * cr * sp, cy * cr * sp + sy * sr * cp,
            sy * cr * cp - cy * sr * sp};
}

// math.py:414-444 (angles come back modulo 2*pi; pitch saturates through |pi/2|*sign)
ILX_DEV f3 euler_xyz_from_quat(f4 q) {
    float sin_roll = 2.0f * (q.w * q.x + q.y * q.z);
    float cos_roll = 1.0f - 2.0f * (q.x * q.x + q.y * q.y);
    float roll = atan2f(sin_roll, cos_roll);
    float sin_pitch = 2.0f * (q.w * q.y - q.z * q.x);
    float pitch = (fabsf(sin_pitch) >= 1.0f) ? (kHalfPi * sign_(sin_pitch)) : asinf(sin_pitch);
    float sin_yaw = 2.0f * (q.w * q.z + q.x * q.y);
    float cos_yaw = 1.0f - 2.0f * (q.y * q.y + q.z * q.z);
    float yaw = atan2f(sin_yaw, cos_yaw);
    return {remainder_(roll, kTwoPi), remainder_(pitch, kTwoPi), remainder_(yaw, kTwoPi)};
}

// math.py:144-174, row-major 3x3 into m[9]
ILX_DEV void matrix_from_quat(f4 q, float* m) {
    float r = q.w, i = q.x, j = q.y, k = q.z;
    float s = ((r * r + i * i) + j * j) + k * k;
    float two_s = div_(2.0f, s);
    m[0] = 1.0f - two_s * (j * j + k * k);
    m[1] = two_s * (i * j - k * r);
    m[2] = two_s * (i * k + j * r);
    m[3] = two_s * (i * j + k * r);
    m[4] = 1.0f - two_s * (i * i + k * k);
    m[5] = two_s * (j * k - i * r);
    m[6] = two_s * (i * k - j * r);
    m[7] = two_s * (j * k + i * r);
    m[8] = 1.0f - two_s * (i * i + j * j);
}

// math.py:750-781 / 785-816
ILX_DEV void combine_frame_transforms(f3 t01, f4 q01, f3 t12, f4 q12, f3& t02, f4& q02) {
    q02 = quat_mul(q01, q12);
    f3 r = quat_apply(q01, t12);
    t02 = {t01.x + r.x, t01.y + r.y, t01.z + r.z};
}
ILX_DEV void subtract_frame_transforms(f3 t01, f4 q01, f3 t02, f4 q02, f3& t12, f4& q12) {
    f4 q10 = quat_inv(q01);
    q12 = quat_mul(q10, q02);
    t12 = quat_apply(q10, {t02.x - t01.x, t02.y - t01.y, t02.z - t01.z});
}

// math.py:820-869: q_err = q02 * (conj(q01) / (q01*conj(q01)).w)
ILX_DEV f4 quat_error(f4 q01, f4 q02) {
    f4 c = quat_conjugate(q01);
    float n = quat_mul(q01, c).w;
    f4 inv = {div_(c.w, n), div_(c.x, n), div_(c.y, n), div_(c.z, n)};
    return quat_mul(q02, inv);
}

// math.py:873-910
ILX_DEV void apply_delta_pose(f3 p, f4 q, const float* d, float eps, f3& po, f4& qo) {
    po = {p.x + d[0], p.y + d[1], p.z + d[2]};
    float angle = norm3(d[3], d[4], d[5]);
    f4 dq = {1.0f, 0.0f, 0.0f, 0.0f};
    if (angle > eps) {
        f3 axis = {div_(d[3], angle), div_(d[4], angle), div_(d[5], angle)};
        dq = quat_from_angle_axis(angle, axis);
    }
    qo = quat_mul(dq, q);
}

// ---- loads / stores of packed rows -------------------------------------------------------------
ILX_DEV f4 load4(const float* p) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    return {v.x, v.y, v.z, v.w};
}
ILX_DEV f4 load4u(const float* p) { return {p[0], p[1], p[2], p[3]}; }
ILX_DEV f3 load3(const float* p) { return {p[0], p[1], p[2]}; }
ILX_DEV void store4(float* p, f4 q) { *reinterpret_cast<float4*>(p) = make_float4(q.w, q.x, q.y, q.z); }
ILX_DEV void store4u(float* p, f4 q) {
    p[0] = q.w;
    p[1] = q.x;
    p[2] = q.y;
    p[3] = q.z;
}
ILX_DEV void store3(float* p, f3 v) {
    p[0] = v.x;
    p[1] = v.y;
    p[2] = v.z;
}

}  // namespace ilx
