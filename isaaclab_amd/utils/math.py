"""Drop-in for ``isaaclab.utils.math`` (rotation + transformation sections) on MI355X.

Same function names, argument order and shapes as ``source/isaaclab/isaaclab/utils/math.py`` of the
reference; every function is one HIP kernel launch (``ilx_math_op``) on the current torch stream instead of
the 5-20 eager kernels the reference issues.  Inputs must be GPU tensors.
"""

from __future__ import annotations

import torch

from .. import _abi
from . import rng as _rng

# op ids: keep in sync with enum ilx_math_op in include/isaaclab_amd.h
(QUAT_ROTATE, QUAT_ROTATE_INVERSE, QUAT_APPLY, QUAT_APPLY_YAW, YAW_QUAT, QUAT_MUL, QUAT_CONJUGATE, QUAT_INV,
 QUAT_UNIQUE, NORMALIZE3, NORMALIZE4, MATRIX_FROM_QUAT, QUAT_BOX_MINUS, AXIS_ANGLE_FROM_QUAT, QUAT_ERROR_MAGNITUDE,
 QUAT_FROM_ANGLE_AXIS, QUAT_FROM_EULER_XYZ, EULER_XYZ_FROM_QUAT, COMBINE_FRAME, SUBTRACT_FRAME, POSE_ERROR_AA,
 POSE_ERROR_QUAT, APPLY_DELTA_POSE, WRAP_TO_PI, SKEW) = range(25)


def _rows(t: torch.Tensor, w: int) -> torch.Tensor:
    if w == 1:
        return _abi.f32c(t).reshape(-1)
    if t.shape[-1] != w:
        raise ValueError(f"expected last dimension {w}, got shape {tuple(t.shape)}")
    return _abi.f32c(t).reshape(-1, w)


def _run(op, ins, widths, out_widths, lead_shape):
    rows = [None if t is None else _rows(t, w) for t, w in zip(ins, widths)]
    n = rows[0].shape[0] if widths[0] > 1 else rows[0].numel()
    for r, w in zip(rows[1:], widths[1:]):
        if r is not None and (r.shape[0] if w > 1 else r.numel()) != n:
            raise ValueError("operand batch sizes differ")
    dev = rows[0].device
    outs = [torch.empty((n, w) if w > 1 else (n,), dtype=torch.float32, device=dev) for w in out_widths]
    p = [_abi.ptr(r) for r in rows] + [None] * (4 - len(rows))
    o = [_abi.ptr(t) for t in outs] + [None] * (2 - len(outs))
    _abi.check(_abi.math_op(op, p[0], p[1], p[2], p[3], o[0], o[1], n, _abi.stream()), f"ilx_math_op({op})")
    shaped = []
    for t, w in zip(outs, out_widths):
        shaped.append(t.reshape(*lead_shape, w) if w > 1 else t.reshape(*lead_shape))
    return shaped


def _same(q1, q2):
    if q1.shape != q2.shape:
        raise ValueError(f"Expected input quaternion shape mismatch: {q1.shape} != {q2.shape}.")


# --- general ------------------------------------------------------------------------------------
def _range(mode, x, lower, upper):
    x = _abi.f32c(x)
    w = x.shape[-1]
    n = x.numel() // w
    lower, upper = _abi.f32c(lower), _abi.f32c(upper)
    per_row = int(lower.numel() != w)
    if per_row:
        lower, upper = lower.expand_as(x).contiguous(), upper.expand_as(x).contiguous()
    out = torch.empty_like(x)
    _abi.check(_abi.math_range_op(mode, _abi.ptr(x), _abi.ptr(lower), _abi.ptr(upper), per_row, _abi.ptr(out), n, w,
                                  _abi.stream()), "ilx_math_range_op")
    return out


def scale_transform(x, lower, upper):
    """math.py:23 — normalise ``x`` from (lower, upper) to [-1, 1]."""
    return _range(0, x, lower, upper)


def unscale_transform(x, lower, upper):
    """math.py:44 — map ``x`` from [-1, 1] to (lower, upper)."""
    return _range(1, x, lower, upper)


def saturate(x, lower, upper):
    """math.py:65 — clamp ``x`` to (lower, upper)."""
    return _range(2, x, lower, upper)


def normalize(x, eps: float = 1e-9):
    """math.py:82 — unit-length ``x`` (3- or 4-vectors)."""
    if eps != 1e-9:
        raise NotImplementedError("normalize: only the default eps is compiled in")
    w = x.shape[-1]
    if w not in (3, 4):
        raise ValueError("normalize: last dimension must be 3 or 4")
    return _run(NORMALIZE3 if w == 3 else NORMALIZE4, [x], [w], [w], x.shape[:-1])[0]


def wrap_to_pi(angles):
    """math.py:96 — wrap to [-pi, pi]; odd positive multiples of pi map to +pi."""
    return _run(WRAP_TO_PI, [angles], [1], [1], angles.shape)[0]


# --- rotations ----------------------------------------------------------------------------------
def matrix_from_quat(quaternions):
    """math.py:144 — (…,4) -> (…,3,3)."""
    m = _run(MATRIX_FROM_QUAT, [quaternions], [4], [9], quaternions.shape[:-1])[0]
    return m.reshape(quaternions.shape[:-1] + (3, 3))


def convert_quat(quat, to: str = "xyzw"):
    """math.py:177 — roll the last axis (pure data movement, stays in torch)."""
    if quat.shape[-1] != 4:
        raise ValueError(f"Expected input quaternion shape mismatch: {quat.shape} != (..., 4).")
    if to not in ("xyzw", "wxyz"):
        raise ValueError(f"Expected input argument `to` to be 'xyzw' or 'wxyz'. Received: {to}.")
    return quat.roll(-1 if to == "xyzw" else 1, dims=-1)


def quat_conjugate(q):
    """math.py:224."""
    return _run(QUAT_CONJUGATE, [q], [4], [4], q.shape[:-1])[0]


def quat_inv(q):
    """math.py:239."""
    return _run(QUAT_INV, [q], [4], [4], q.shape[:-1])[0]


def quat_from_euler_xyz(roll, pitch, yaw):
    """math.py:252."""
    return _run(QUAT_FROM_EULER_XYZ, [roll, pitch, yaw], [1, 1, 1], [4], roll.shape)[0]


def euler_xyz_from_quat(quat):
    """math.py:414 — returns (roll, pitch, yaw), each modulo 2*pi as the reference does."""
    e = _run(EULER_XYZ_FROM_QUAT, [quat], [4], [3], quat.shape[:-1])[0]
    return e[..., 0], e[..., 1], e[..., 2]


def quat_unique(q):
    """math.py:448."""
    return _run(QUAT_UNIQUE, [q], [4], [4], q.shape[:-1])[0]


def quat_mul(q1, q2):
    """math.py:464."""
    _same(q1, q2)
    return _run(QUAT_MUL, [q1, q2], [4, 4], [4], q1.shape[:-1])[0]


def quat_box_minus(q1, q2):
    """math.py:503."""
    return _run(QUAT_BOX_MINUS, [q1, q2], [4, 4], [3], q1.shape[:-1])[0]


def yaw_quat(quat):
    """math.py:522."""
    return _run(YAW_QUAT, [quat], [4], [4], quat.shape[:-1])[0]


def quat_apply(quat, vec):
    """math.py:546."""
    return _run(QUAT_APPLY, [quat, vec], [4, 3], [3], vec.shape[:-1])[0]


def quat_apply_yaw(quat, vec):
    """math.py:568."""
    return _run(QUAT_APPLY_YAW, [quat, vec], [4, 3], [3], vec.shape[:-1])[0]


def quat_rotate(q, v):
    """math.py:583."""
    return _run(QUAT_ROTATE, [q, v], [4, 3], [3], v.shape[:-1])[0]


def quat_rotate_inverse(q, v):
    """math.py:606."""
    return _run(QUAT_ROTATE_INVERSE, [q, v], [4, 3], [3], v.shape[:-1])[0]


def quat_from_angle_axis(angle, axis):
    """math.py:629."""
    return _run(QUAT_FROM_ANGLE_AXIS, [angle, axis], [1, 3], [4], angle.shape)[0]


def axis_angle_from_quat(quat, eps: float = 1.0e-6):
    """math.py:646."""
    if eps != 1.0e-6:
        raise NotImplementedError("axis_angle_from_quat: only the default eps is compiled in")
    return _run(AXIS_ANGLE_FROM_QUAT, [quat], [4], [3], quat.shape[:-1])[0]


def quat_error_magnitude(q1, q2):
    """math.py:678."""
    return _run(QUAT_ERROR_MAGNITUDE, [q1, q2], [4, 4], [1], q1.shape[:-1])[0]


def skew_symmetric_matrix(vec):
    """math.py:693."""
    if vec.shape[-1] != 3:
        raise ValueError(f"Expected input vector shape mismatch: {vec.shape} != (..., 3).")
    if vec.ndim == 1:
        vec = vec.unsqueeze(0)
    return _run(SKEW, [vec], [3], [9], vec.shape[:-1])[0].reshape(vec.shape[0], 3, 3)


# --- transformations ----------------------------------------------------------------------------
def _identity_like(n, device):
    q = torch.zeros(n, 4, dtype=torch.float32, device=device)
    q[:, 0] = 1.0
    return q


def combine_frame_transforms(t01, q01, t12=None, q12=None):
    """math.py:750 — T02 = T01 x T12."""
    if t12 is None and q12 is None:
        return t01, q01
    if q12 is None:
        return t01 + quat_apply(q01, t12), q01
    if t12 is None:
        return t01, quat_mul(q01, q12)
    t, q = _run(COMBINE_FRAME, [t01, q01, t12, q12], [3, 4, 3, 4], [3, 4], t01.shape[:-1])
    return t, q


def subtract_frame_transforms(t01, q01, t02=None, q02=None):
    """math.py:785 — T12 = T01^-1 x T02."""
    if q02 is None:
        q10 = quat_inv(q01)
        t = quat_apply(q10, (t02 - t01) if t02 is not None else -t01)
        return t, q10
    if t02 is None:
        t02 = torch.zeros_like(t01)
    t, q = _run(SUBTRACT_FRAME, [t01, q01, t02, q02], [3, 4, 3, 4], [3, 4], t01.shape[:-1])
    return t, q


def compute_pose_error(t01, q01, t02, q02, rot_error_type: str = "axis_angle"):
    """math.py:820."""
    if rot_error_type == "quat":
        p, r = _run(POSE_ERROR_QUAT, [t01, q01, t02, q02], [3, 4, 3, 4], [3, 4], t01.shape[:-1])
    elif rot_error_type == "axis_angle":
        p, r = _run(POSE_ERROR_AA, [t01, q01, t02, q02], [3, 4, 3, 4], [3, 3], t01.shape[:-1])
    else:
        raise ValueError(f"Unsupported orientation error type: {rot_error_type}. Valid: 'quat', 'axis_angle'.")
    return p, r


def apply_delta_pose(source_pos, source_rot, delta_pose, eps: float = 1.0e-6):
    """math.py:873."""
    if eps != 1.0e-6:
        raise NotImplementedError("apply_delta_pose: only the default eps is compiled in")
    p, q = _run(APPLY_DELTA_POSE, [source_pos, source_rot, delta_pose], [3, 4, 6], [3, 4], source_pos.shape[:-1])
    return p, q


def transform_points(points, pos=None, quat=None):
    """math.py:914 — p' = R(quat) p + pos for (N,P,3) or (P,3) point sets."""
    pts = points
    batched = pts.dim() == 3
    if pts.dim() == 2:
        pts = pts[None]
    if pts.dim() != 3:
        raise ValueError(f"Expected points to have dim = 2 or dim = 3: got shape {points.shape}")
    if quat is not None:
        if quat.dim() not in (1, 2):
            raise ValueError(f"Expected quat to have dim = 1 or dim = 2: got shape {quat.shape}")
        # not on the per-step path: rotation matrix from the HIP kernel, the (3x3)(3xP) product in torch
        rot = matrix_from_quat(quat[None] if quat.dim() == 1 else quat)
        pts = torch.matmul(rot, pts.transpose(1, 2)).transpose(1, 2)
    if pos is not None:
        if pos.dim() not in (1, 2):
            raise ValueError(f"Expected pos to have dim = 1 or dim = 2: got shape {pos.shape}")
        pts = pts + (pos[None, None, :] if pos.dim() == 1 else pos[:, None, :])
    elif quat is None:
        pts = pts.clone()
    return pts if batched else pts.squeeze(0)


# ---- samplers (math.py:1225-1390) ---------------------------------------------------------------------------------
# The reference draws with torch.rand / torch.randn on the global generator; these draw on the device with Philox keyed
# by (seed, call counter, row, column).  ``_u`` / ``_z`` hand in unit-interval / standard-normal draws instead.
SAMPLE_UNIFORM, SAMPLE_LOG_UNIFORM, SAMPLE_TRIANGLE, SAMPLE_GAUSSIAN, SAMPLE_RANDOM_ORIENTATION, SAMPLE_RANDOM_YAW_ORIENTATION = range(6)


def set_sampling_seed(seed: int):
    """Seed of the device generators (the samplers share it with every other drawing module; the draw counter restarts)."""
    _rng.seed_all(seed)


def _operand(v, rows: int, width: int, device):
    """(tensor pointer or None, row stride, column stride, scalar) of a sampler bound given as a float or a tensor that
    broadcasts against (rows, width)."""
    if not isinstance(v, torch.Tensor):
        return None, 0, 0, float(v), None
    t = _abi.f32c(v.to(device))
    if t.numel() == 1:
        return None, 0, 0, float(t), None
    if t.dim() >= 1 and t.shape[-1] == width and t.numel() == width:
        return _abi.ptr(t), 0, 1, 0.0, t
    if t.numel() == rows * width:
        return _abi.ptr(t), width, 1, 0.0, t
    if t.numel() == rows:  # one value per row, e.g. bounds of shape (rows, 1)
        return _abi.ptr(t), 1, 0, 0.0, t
    raise ValueError(f"sampler bound of shape {tuple(v.shape)} does not broadcast against ({rows}, {width})")


def _sample(kind: int, a, b, size, device, u=None, z=None) -> torch.Tensor:
    if isinstance(size, int):
        size = (size,)
    size = tuple(size)
    dev = torch.device(device)
    if dev.type != "cuda":
        raise _abi.IlxError("isaaclab_amd samplers draw on the GPU; there is no CPU fallback")
    width = size[-1] if len(size) > 1 else 1
    rows = 1
    for s in (size[:-1] if len(size) > 1 else size):
        rows *= s
    out = torch.empty(size, device=dev)
    ap, ar, ac, asc, _ka = _operand(a, rows, width, dev)
    bp, br, bc, bsc, _kb = _operand(b, rows, width, dev)
    _abi.check(_abi.sample(kind, ap, ar, ac, asc, bp, br, bc, bsc, None if u is None else _abi.ptr(_abi.f32c(u.to(dev))),
                           None if z is None else _abi.ptr(_abi.f32c(z.to(dev))), _rng.seed(), _rng.next_draw("sampling"), _abi.ptr(out),
                           rows, width, _abi.stream()), "ilx_sample")
    return out


def sample_uniform(lower, upper, size, device, *, _u=None) -> torch.Tensor:
    """math.py:1313-1331."""
    return _sample(SAMPLE_UNIFORM, lower, upper, size, device, u=_u)


def sample_log_uniform(lower, upper, size, device, *, _u=None) -> torch.Tensor:
    """math.py:1334-1368."""
    return _sample(SAMPLE_LOG_UNIFORM, lower, upper, size, device, u=_u)


def sample_triangle(lower: float, upper: float, size, device, *, _u=None) -> torch.Tensor:
    """math.py:1288-1310."""
    return _sample(SAMPLE_TRIANGLE, lower, upper, size, device, u=_u)


def sample_gaussian(mean, std, size, device, *, _z=None) -> torch.Tensor:
    """math.py:1371-1390 (with tensor ``mean`` / ``std`` the reference ignores ``size`` and takes their shape)."""
    if isinstance(mean, torch.Tensor) and mean.numel() > 1:
        size = tuple(mean.shape)
    return _sample(SAMPLE_GAUSSIAN, mean, std, size, device, z=_z)


def default_orientation(num: int, device) -> torch.Tensor:
    """math.py:1228-1248."""
    quat = torch.zeros((num, 4), dtype=torch.float, device=device)
    quat[..., 0] = 1.0
    return quat


def random_orientation(num: int, device, *, _z=None) -> torch.Tensor:
    """math.py:1251-1267."""
    return _sample(SAMPLE_RANDOM_ORIENTATION, 0.0, 0.0, (num, 4), device, z=_z)


def random_yaw_orientation(num: int, device, *, _u=None) -> torch.Tensor:
    """math.py:1270-1285."""
    return _sample(SAMPLE_RANDOM_YAW_ORIENTATION, 0.0, 0.0, (num, 4), device, u=_u)


# homogeneous poses, interpolation, camera geometry (math.py:121-135, 728-746, 991-1225, 1387-1420, 1564-1883)
from .pose_math import (copysign, generate_random_rotation, generate_random_transformation_matrix,  # noqa: E402,F401
                        generate_random_translation, interpolate_poses, interpolate_rotations, is_identity_pose, make_pose,
                        orthogonalize_perspective_depth, pose_in_A_to_pose_in_B, pose_inv, project_points, quat_slerp,
                        sample_cylinder, transform_poses_from_frame_A_to_frame_B, unmake_pose, unproject_depth)
