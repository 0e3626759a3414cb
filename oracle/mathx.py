"""Oracle for the quaternion / frame math on the rollout path (test infrastructure, CPU only).

Each public function restates one function of the reference's ``isaaclab/utils/math.py`` (cited per
function as ``math.py:<line>``) with the evaluation order written out per component, which is the order
the HIP device functions in ``isaaclab_amd/csrc/math_dev.h`` use.
"""

from __future__ import annotations

import math

import torch

PI32 = torch.tensor(math.pi, dtype=torch.float32)
TWO_PI32 = torch.tensor(2 * math.pi, dtype=torch.float32)
HALF_PI32 = torch.tensor(math.pi / 2.0, dtype=torch.float32)


# ----------------------------------------------------------------------------------------------
# exact primitives
# ----------------------------------------------------------------------------------------------
def fma(a, b, c):
    """Single-rounding a*b+c on binary32 tensors."""
    return (a.double() * b.double() + c.double()).float()


def sqrt(x):
    """Correctly rounded binary32 square root."""
    return x.double().sqrt().float()


def f32(v: float) -> torch.Tensor:
    """Host double rounded once to binary32 (how torch treats Python scalars next to fp32 tensors)."""
    return torch.tensor(float(v), dtype=torch.float32)


def cross3(a, b):
    """(ax,ay,az) x (bx,by,bz); each component is fma(a1, b2, -(a2*b1)) as torch.cross evaluates it."""
    ax, ay, az = a
    bx, by, bz = b
    return (fma(ay, bz, -(az * by)), fma(az, bx, -(ax * bz)), fma(ax, by, -(ay * bx)))


def dot3(a, b):
    """Left-to-right, unfused: (ax*bx + ay*by) + az*bz."""
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]


def norm2(x, y):
    return sqrt(fma(y, y, x * x))


def norm3(x, y, z):
    return sqrt(fma(z, z, fma(y, y, x * x)))


def norm4(w, x, y, z):
    """4-vector norm: torch reduces 4 contiguous squares left to right without fma."""
    return sqrt(((w * w + x * x) + y * y) + z * z)


def _q(q):
    return q[..., 0], q[..., 1], q[..., 2], q[..., 3]


def _v(v):
    return v[..., 0], v[..., 1], v[..., 2]


def _stack(*c):
    return torch.stack(torch.broadcast_tensors(*c), dim=-1)


# ----------------------------------------------------------------------------------------------
# general (math.py:22-135)
# ----------------------------------------------------------------------------------------------
def scale_transform(x, lower, upper):
    """math.py:23-40: 2*(x - (lower+upper)*0.5)/(upper-lower)."""
    offset = (lower + upper) * f32(0.5)
    return f32(2.0) * (x - offset) / (upper - lower)


def unscale_transform(x, lower, upper):
    """math.py:44-61."""
    offset = (lower + upper) * f32(0.5)
    return x * (upper - lower) * f32(0.5) + offset


def saturate(x, lower, upper):
    """math.py:65-78: max(min(x, upper), lower)."""
    return torch.maximum(torch.minimum(x, upper), lower)


def normalize(x, eps: float = 1e-9):
    """math.py:82-92.  Supports 3- and 4-vectors (the only widths on the path)."""
    d = x.shape[-1]
    if d == 4:
        n = norm4(*_q(x))
    elif d == 3:
        n = norm3(*_v(x))
    elif d == 2:
        n = norm2(x[..., 0], x[..., 1])
    else:
        raise ValueError("oracle.normalize: width must be 2, 3 or 4")
    n = torch.clamp(n, min=float(f32(eps)))
    return x / n.unsqueeze(-1)


def wrap_to_pi(angles):
    """math.py:96-117: ((a + pi) mod 2pi) - pi, with odd positive multiples of pi mapped to +pi."""
    wrapped = torch.remainder(angles + PI32, TWO_PI32)
    return torch.where((wrapped == 0) & (angles > 0), PI32, wrapped - PI32)


def copysign(mag: float, other):
    """math.py:121-135: |mag| * sign(other)  (sign(0) = 0, unlike C copysign)."""
    return torch.abs(f32(mag)) * torch.sign(other)


# ----------------------------------------------------------------------------------------------
# rotations (math.py:143-720)
# ----------------------------------------------------------------------------------------------
def matrix_from_quat(q):
    """math.py:144-174.  two_s = 2/sum(q*q); entries as written there, row-major (...,3,3)."""
    r, i, j, k = _q(q)
    s = ((r * r + i * i) + j * j) + k * k
    two_s = f32(2.0) / s
    one = f32(1.0)
    o = _stack(
        one - two_s * (j * j + k * k),
        two_s * (i * j - k * r),
        two_s * (i * k + j * r),
        two_s * (i * j + k * r),
        one - two_s * (i * i + k * k),
        two_s * (j * k - i * r),
        two_s * (i * k - j * r),
        two_s * (j * k + i * r),
        one - two_s * (i * i + j * j),
    )
    return o.reshape(q.shape[:-1] + (3, 3))


def convert_quat(q, to: str = "xyzw"):
    """math.py:177-220."""
    if to == "xyzw":
        return torch.roll(q, -1, dims=-1)
    if to == "wxyz":
        return torch.roll(q, 1, dims=-1)
    raise ValueError(to)


def quat_conjugate(q):
    """math.py:224-235."""
    w, x, y, z = _q(q)
    return _stack(w, -x, -y, -z)


def quat_inv(q):
    """math.py:239-248: normalize(conjugate(q))."""
    return normalize(quat_conjugate(q))


def quat_from_euler_xyz(roll, pitch, yaw):
    """math.py:252-278."""
    h = f32(0.5)
    cy, sy = torch.cos(yaw * h), torch.sin(yaw * h)
    cr, sr = torch.cos(roll * h), torch.sin(roll * h)
    cp, sp = torch.cos(pitch * h), torch.sin(pitch * h)
    qw = cy * cr * cp + sy * sr * sp
    qx = cy * sr * cp - sy * cr * sp
    qy = cy * cr * sp + sy * sr * cp
    qz = sy * cr * cp - cy * sr * sp
    return _stack(qw, qx, qy, qz)


def euler_xyz_from_quat(q):
    """math.py:414-444.  Returns angles modulo 2*pi, pitch saturated through the ``copysign`` helper."""
    w, x, y, z = _q(q)
    two, one = f32(2.0), f32(1.0)
    sin_roll = two * (w * x + y * z)
    cos_roll = one - two * (x * x + y * y)
    roll = torch.atan2(sin_roll, cos_roll)
    sin_pitch = two * (w * y - z * x)
    pitch = torch.where(torch.abs(sin_pitch) >= 1, copysign(math.pi / 2.0, sin_pitch), torch.asin(sin_pitch))
    sin_yaw = two * (w * z + x * y)
    cos_yaw = one - two * (y * y + z * z)
    yaw = torch.atan2(sin_yaw, cos_yaw)
    return torch.remainder(roll, TWO_PI32), torch.remainder(pitch, TWO_PI32), torch.remainder(yaw, TWO_PI32)


def quat_unique(q):
    """math.py:448-460."""
    return torch.where(q[..., 0:1] < 0, -q, q)


def _quat_mul_c(a, b):
    """math.py:464-499 on components: the 9-product form, evaluated in the order written there."""
    w1, x1, y1, z1 = a
    w2, x2, y2, z2 = b
    ww = (z1 + x1) * (x2 + y2)
    yy = (w1 - y1) * (w2 + z2)
    zz = (w1 + y1) * (w2 - z2)
    xx = ww + yy + zz
    qq = f32(0.5) * (xx + (z1 - x1) * (x2 - y2))
    w = qq - ww + (z1 - y1) * (y2 - z2)
    x = qq - xx + (x1 + w1) * (x2 + w2)
    y = qq - yy + (w1 - x1) * (y2 + z2)
    z = qq - zz + (z1 + y1) * (w2 - x2)
    return w, x, y, z


def quat_mul(q1, q2):
    """math.py:464-499."""
    if q1.shape != q2.shape:
        raise ValueError(f"Expected input quaternion shape mismatch: {q1.shape} != {q2.shape}.")
    return _stack(*_quat_mul_c(_q(q1), _q(q2)))


def quat_box_minus(q1, q2):
    """math.py:503-518."""
    w, x, y, z = _quat_mul_c(_q(q1), _q(quat_conjugate(q2)))
    n = norm3(x, y, z)
    scale = f32(2.0) * torch.where(n > float(f32(1.0e-7)), torch.atan2(n, w) / n, torch.sign(w))
    return _stack(scale * x, scale * y, scale * z)


def yaw_quat(q):
    """math.py:522-542: yaw = atan2(2(wz+xy), 1-2(yy+zz)); (cos(yaw/2),0,0,sin(yaw/2)) normalised."""
    w, x, y, z = _q(q)
    yaw = torch.atan2(f32(2.0) * (w * z + x * y), f32(1.0) - f32(2.0) * (y * y + z * z))
    half = yaw / f32(2.0)
    c, s = torch.cos(half), torch.sin(half)
    zero = torch.zeros_like(c)
    n = torch.clamp(norm4(c, zero, zero, s), min=float(f32(1e-9)))
    return _stack(c / n, zero / n, zero / n, s / n)


def _quat_apply_c(q, v):
    """math.py:546-564 on components: t = 2*(xyz x v); v + w*t + xyz x t."""
    w, x, y, z = q
    t = cross3((x, y, z), v)
    t = (t[0] * f32(2.0), t[1] * f32(2.0), t[2] * f32(2.0))
    u = cross3((x, y, z), t)
    return (v[0] + w * t[0] + u[0], v[1] + w * t[1] + u[1], v[2] + w * t[2] + u[2])


def quat_apply(q, v):
    """math.py:546-564."""
    return _stack(*_quat_apply_c(_q(q), _v(v)))


def quat_apply_yaw(q, v):
    """math.py:568-579."""
    return quat_apply(yaw_quat(q), v)


def _quat_rotate_c(q, v, inverse: bool):
    """math.py:583-625: a = v(2w^2-1); b = 2w (q_vec x v); c = 2 q_vec (q_vec . v); a -/+ b + c."""
    w, x, y, z = q
    k = f32(2.0) * (w * w) - f32(1.0)
    a = (v[0] * k, v[1] * k, v[2] * k)
    cr = cross3((x, y, z), v)
    b = tuple(ci * w * f32(2.0) for ci in cr)
    d = dot3((x, y, z), v)
    c = (x * d * f32(2.0), y * d * f32(2.0), z * d * f32(2.0))
    if inverse:
        return tuple(a[i] - b[i] + c[i] for i in range(3))
    return tuple(a[i] + b[i] + c[i] for i in range(3))


def quat_rotate(q, v):
    """math.py:583-602."""
    return _stack(*_quat_rotate_c(_q(q), _v(v), False))


def quat_rotate_inverse(q, v):
    """math.py:606-625."""
    return _stack(*_quat_rotate_c(_q(q), _v(v), True))


def quat_from_angle_axis(angle, axis):
    """math.py:629-642: normalize(cat(cos(a/2), normalize(axis)*sin(a/2)))."""
    theta = angle / f32(2.0)
    ax, ay, az = _v(axis)
    n = torch.clamp(norm3(ax, ay, az), min=float(f32(1e-9)))
    s, c = torch.sin(theta), torch.cos(theta)
    x, y, z = ax / n * s, ay / n * s, az / n * s
    n4 = torch.clamp(norm4(c, x, y, z), min=float(f32(1e-9)))
    return _stack(c / n4, x / n4, y / n4, z / n4)


def axis_angle_from_quat(q, eps: float = 1.0e-6):
    """math.py:646-674."""
    w, x, y, z = _q(q)
    sgn = f32(1.0) - f32(2.0) * (w < 0.0).float()
    w, x, y, z = w * sgn, x * sgn, y * sgn, z * sgn
    mag = norm3(x, y, z)
    half = torch.atan2(mag, w)
    angle = f32(2.0) * half
    denom = torch.where(angle.abs() > float(f32(eps)), torch.sin(half) / angle, f32(0.5) - angle * angle / f32(48.0))
    return _stack(x / denom, y / denom, z / denom)


def quat_error_magnitude(q1, q2):
    """math.py:678-689."""
    d = quat_mul(q1, quat_conjugate(q2))
    aa = axis_angle_from_quat(d)
    return norm3(*_v(aa))


def skew_symmetric_matrix(vec):
    """math.py:693-720."""
    if vec.ndim == 1:
        vec = vec.unsqueeze(0)
    x, y, z = _v(vec)
    o = torch.zeros_like(x)
    return _stack(o, -z, y, z, o, -x, -y, x, o).reshape(vec.shape[0], 3, 3)


# ----------------------------------------------------------------------------------------------
# transformations (math.py:749-982)
# ----------------------------------------------------------------------------------------------
def combine_frame_transforms(t01, q01, t12=None, q12=None):
    """math.py:750-781."""
    q02 = quat_mul(q01, q12) if q12 is not None else q01
    t02 = t01 + quat_apply(q01, t12) if t12 is not None else t01
    return t02, q02


def subtract_frame_transforms(t01, q01, t02=None, q02=None):
    """math.py:785-816."""
    q10 = quat_inv(q01)
    q12 = quat_mul(q10, q02) if q02 is not None else q10
    t12 = quat_apply(q10, t02 - t01) if t02 is not None else quat_apply(q10, -t01)
    return t12, q12


def compute_pose_error(t01, q01, t02, q02, rot_error_type: str = "axis_angle"):
    """math.py:820-869."""
    conj = quat_conjugate(q01)
    nrm = _quat_mul_c(_q(q01), _q(conj))[0]
    inv = conj / nrm.unsqueeze(-1)
    q_err = quat_mul(q02, inv)
    p_err = t02 - t01
    if rot_error_type == "quat":
        return p_err, q_err
    if rot_error_type == "axis_angle":
        return p_err, axis_angle_from_quat(q_err)
    raise ValueError(f"Unsupported orientation error type: {rot_error_type}. Valid: 'quat', 'axis_angle'.")


def apply_delta_pose(source_pos, source_rot, delta_pose, eps: float = 1.0e-6):
    """math.py:873-910.  The axis is delta/angle (NaN for a zero rotation) and is masked afterwards."""
    target_pos = source_pos + delta_pose[:, 0:3]
    rx, ry, rz = delta_pose[:, 3], delta_pose[:, 4], delta_pose[:, 5]
    angle = norm3(rx, ry, rz)
    axis = _stack(rx / angle, ry / angle, rz / angle)
    dq = quat_from_angle_axis(angle, axis)
    ident = torch.tensor([1.0, 0.0, 0.0, 0.0], dtype=torch.float32).expand_as(dq)
    dq = torch.where((angle > float(f32(eps))).unsqueeze(-1), dq, ident)
    return target_pos, quat_mul(dq, source_rot)


def transform_points(points, pos=None, quat=None):
    """math.py:914-982: p' = R(quat) p + pos, batched (N,P,3) or single (P,3)."""
    pts = points.clone()
    batched = pts.dim() == 3
    if pts.dim() == 2:
        pts = pts[None]
    if quat is not None:
        R = matrix_from_quat(quat)
        if R.dim() == 2:
            R = R[None]
        # (R p) row-wise: r0.p, r1.p, r2.p, left to right without fma is what a 3-term matmul yields
        px, py, pz = pts[..., 0], pts[..., 1], pts[..., 2]
        rows = []
        for r in range(3):
            a, b, c = R[:, r, 0:1], R[:, r, 1:2], R[:, r, 2:3]
            rows.append((a * px + b * py) + c * pz)
        pts = torch.stack(rows, dim=-1)
    if pos is not None:
        pts = pts + (pos[None, None, :] if pos.dim() == 1 else pos[:, None, :])
    return pts if batched else pts.squeeze(0)


# ---- samplers (math.py:1225-1390), with the unit-interval / standard-normal draws as arguments ---------------------
def sample_uniform(u, lower, upper):
    return u * (upper - lower) + lower


def sample_log_uniform(u, lower, upper):
    lower, upper = torch.as_tensor(lower, dtype=torch.float32), torch.as_tensor(upper, dtype=torch.float32)
    return torch.exp(sample_uniform(u, torch.log(lower), torch.log(upper)))


def sample_triangle(u, lower, upper):
    r = 2 * u - 1
    r = torch.where(r < 0.0, -torch.sqrt(-r), torch.sqrt(r))
    r = (r + 1.0) / 2.0
    return (upper - lower) * r + lower


def sample_gaussian(z, mean, std):
    return z * std + mean


def random_orientation(z):
    return torch.nn.functional.normalize(z, p=2.0, dim=-1, eps=1e-12)


def random_yaw_orientation(u):
    zero = torch.zeros_like(u)
    return quat_from_euler_xyz(zero, zero, 2 * torch.pi * u)


# ---- homogeneous poses and camera geometry (math.py:121-135, 991-1225, 1387-1420, 1564-1643) ------------------------------
def copysign(mag, other):
    return torch.full_like(other, abs(mag)) * torch.sign(other)


def make_pose(pos, rot):
    pose = torch.zeros(pos.shape[:-1] + (4, 4))
    pose[..., :3, :3], pose[..., :3, 3], pose[..., 3, 3] = rot, pos, 1.0
    return pose


def pose_inv(pose):
    inv = torch.zeros_like(pose)
    rt = pose[..., :3, :3].transpose(-1, -2)
    inv[..., :3, :3] = rt
    inv[..., :3, 3] = torch.matmul(-rt, pose[..., :3, 3:4])[..., 0]
    inv[..., 3, 3] = 1.0
    return inv


def pose_mul(pose_A_in_B, pose_in_A):
    return torch.matmul(pose_A_in_B, pose_in_A)


def orthogonalize_depth(depth, K):
    """depth (N, H, W), K (N or 1, 3, 3)."""
    n, h, w = depth.shape
    u = torch.arange(w, dtype=torch.float32).view(1, 1, w)
    v = torch.arange(h, dtype=torch.float32).view(1, h, 1)
    fx, fy, cx, cy = (K[:, i, j].view(-1, 1, 1) for i, j in ((0, 0), (1, 1), (0, 2), (1, 2)))
    return depth / torch.sqrt(1 + ((u - cx) / fx) ** 2 + ((v - cy) / fy) ** 2)


def unproject_depth(depth, K):
    """depth (N, H, W) orthogonal, K (N or 1, 3, 3) -> (N, W * H, 3), pixel u * H + v."""
    n, h, w = depth.shape
    uu, vv = torch.meshgrid(torch.arange(w, dtype=torch.float32), torch.arange(h, dtype=torch.float32), indexing="ij")
    pix = torch.stack([uu.reshape(-1), vv.reshape(-1), torch.ones(h * w)], dim=0)[None]
    pts = torch.matmul(torch.inverse(K.double()).float(), pix)
    pts = pts / pts[:, -1:, :]
    return pts.transpose(1, 2) * depth.transpose(1, 2).reshape(n, -1, 1)


def project_points(points, K):
    """points (N, P, 3), K (N or 1, 3, 3) -> (N, P, 3) = (u, v, Z)."""
    p = torch.matmul(K, points.transpose(1, 2))
    p = (p / p[:, -1:, :]).transpose(1, 2).clone()
    p[:, :, -1] = points[:, :, -1]
    return p


def sample_cylinder(radius, h_range, u):
    angles = (u[0] * 2 - 1) * torch.pi
    return torch.stack([radius * torch.cos(angles), radius * torch.sin(angles), u[1] * (h_range[1] - h_range[0]) + h_range[0]], dim=-1)
