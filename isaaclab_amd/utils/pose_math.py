"""Homogeneous-pose, interpolation and camera-geometry utilities of ``isaaclab/utils/math.py`` on the HIP kernels
``ilx_pose_op`` / ``ilx_camera_op`` / ``ilx_sample_cylinder``.  Re-exported from :mod:`isaaclab_amd.utils.math`."""

from __future__ import annotations

import ctypes
import math
from ctypes import c_float, c_int32, c_int64, c_uint64

import torch

from .. import _abi
from . import rng as _rng

POSE_MAKE, POSE_INV, POSE_MUL, POSE_COPYSIGN, POSE_IS_IDENTITY = range(5)
CAMERA_ORTHOGONALIZE, CAMERA_UNPROJECT, CAMERA_PROJECT = range(3)

_pose_op = _abi._sig("ilx_pose_op", c_int32, _abi.P, c_int64, _abi.P, c_int64, c_float, _abi.P, _abi.P, c_int64, _abi.P)
_camera_op = _abi._sig("ilx_camera_op", c_int32, _abi.P, _abi.P, c_int64, _abi.P, c_int64, c_int32, c_int32, _abi.P)
_cylinder = _abi._sig("ilx_sample_cylinder", c_float, c_float, c_float, _abi.P, c_uint64, c_uint64, _abi.P, c_int64, _abi.P)


def _pose(op, a, a_batch, b, b_batch, out, n, scalar=0.0, flag=None):
    _abi.check(_pose_op(op, _abi.ptr(a), a_batch, _abi.ptr(b), b_batch, float(scalar), _abi.ptr(out),
                        None if flag is None else flag.data_ptr(), n, _abi.stream()), f"ilx_pose_op({op})")


def copysign(mag: float, other: torch.Tensor) -> torch.Tensor:
    """math.py:121-135."""
    src = _abi.f32c(other)
    out = torch.empty(src.shape[0], dtype=torch.float32, device=src.device)
    _pose(POSE_COPYSIGN, src, out.numel(), None, 1, out, out.numel(), scalar=mag)
    return out


def is_identity_pose(pos: torch.Tensor, rot: torch.Tensor) -> bool:
    """math.py:728-746: True when every position is zero and every quaternion (1, 0, 0, 0) to ``torch.allclose`` defaults."""
    p, q = _abi.f32c(pos).reshape(-1, 3), _abi.f32c(rot).reshape(-1, 4)
    if p.shape[0] != q.shape[0]:
        raise ValueError(f"{p.shape[0]} positions for {q.shape[0]} orientations")
    flag = torch.zeros(1, dtype=torch.int32, device=p.device)
    _pose(POSE_IS_IDENTITY, p, p.shape[0], q, q.shape[0], None, p.shape[0], flag=flag)
    return int(flag.item()) == 0


def make_pose(pos: torch.Tensor, rot: torch.Tensor) -> torch.Tensor:
    """math.py:1564-1583: (..., 3) positions and (..., 3, 3) rotations -> (..., 4, 4)."""
    assert isinstance(pos, torch.Tensor), "Input must be a torch tensor"
    assert isinstance(rot, torch.Tensor), "Input must be a torch tensor"
    assert pos.shape[:-1] == rot.shape[:-2]
    assert pos.shape[-1] == rot.shape[-2] == rot.shape[-1] == 3
    p, r = _abi.f32c(pos).reshape(-1, 3), _abi.f32c(rot).reshape(-1, 9)
    out = torch.empty(pos.shape[:-1] + (4, 4), dtype=torch.float32, device=p.device)
    _pose(POSE_MAKE, p, p.shape[0], r, r.shape[0], out, p.shape[0])
    return out


def unmake_pose(pose: torch.Tensor):
    """math.py:1586-1598 (views, no compute)."""
    assert isinstance(pose, torch.Tensor), "Input must be a torch tensor"
    return pose[..., :3, 3], pose[..., :3, :3]


def pose_inv(pose: torch.Tensor) -> torch.Tensor:
    """math.py:1601-1626."""
    assert isinstance(pose, torch.Tensor), "Input must be a torch tensor"
    assert len(pose.shape) >= 2
    m = _abi.f32c(pose).reshape(-1, 16)
    out = torch.empty(pose.shape, dtype=torch.float32, device=m.device)
    _pose(POSE_INV, m, m.shape[0], None, 1, out, m.shape[0])
    return out


def pose_in_A_to_pose_in_B(pose_in_A: torch.Tensor, pose_A_in_B: torch.Tensor) -> torch.Tensor:
    """math.py:1629-1643: ``pose_A_in_B @ pose_in_A``; either side may be one matrix against a batch."""
    assert isinstance(pose_in_A, torch.Tensor), "Input must be a torch tensor"
    assert isinstance(pose_A_in_B, torch.Tensor), "Input must be a torch tensor"
    lead = tuple(torch.broadcast_shapes(pose_A_in_B.shape[:-2], pose_in_A.shape[:-2]))
    n = int(math.prod(lead))

    def flat(t):
        m = _abi.f32c(t).reshape(-1, 16)
        return m if m.shape[0] in (1, n) else _abi.f32c(t.expand(lead + (4, 4))).reshape(-1, 16)

    a, b = flat(pose_A_in_B), flat(pose_in_A)
    out = torch.empty(lead + (4, 4), dtype=torch.float32, device=a.device)
    _pose(POSE_MUL, a, a.shape[0], b, b.shape[0], out, n)
    return out


def transform_poses_from_frame_A_to_frame_B(src_poses, frame_A, frame_B):
    """math.py:1794-1821: poses kept fixed relative to frame B are re-expressed against frame A."""
    rel = pose_in_A_to_pose_in_B(pose_in_A=src_poses, pose_A_in_B=pose_inv(frame_B[None]))
    return pose_in_A_to_pose_in_B(pose_in_A=rel, pose_A_in_B=frame_A[None])


# ---- camera geometry (math.py:991-1225) ----------------------------------------------------------------------------
def _depth_batch(depth: torch.Tensor):
    """(depth as (N, H, W), batched?, trailing singleton?) with the reference's shape rules (math.py:1017-1032)."""
    d = depth
    is_batched = d.dim() == 4 or (d.dim() == 3 and d.shape[-1] != 1)
    add_last_dim = False
    if d.dim() == 4 and d.shape[-1] == 1:
        add_last_dim, d = True, d.squeeze(dim=3)
    if d.dim() == 3 and d.shape[-1] == 1:
        add_last_dim, d = True, d.squeeze(dim=2)
    if d.dim() == 2:
        d = d[None]
    return d, is_batched, add_last_dim


def _intrinsics_batch(intrinsics: torch.Tensor, n: int):
    k = intrinsics[None] if intrinsics.dim() == 2 else intrinsics
    if k.dim() != 3:
        raise ValueError(f"Expected intrinsics to have shape (3, 3) or (N, 3, 3); got {intrinsics.shape}.")
    k = _abi.f32c(k)
    if k.shape[0] not in (1, n):
        raise RuntimeError(f"{k.shape[0]} calibration matrices for {n} images")
    return k


def orthogonalize_perspective_depth(depth: torch.Tensor, intrinsics: torch.Tensor) -> torch.Tensor:
    """math.py:991-1079: distance to the optical centre -> distance to the image plane."""
    d, is_batched, add_last_dim = _depth_batch(depth)
    if d.dim() != 3:
        raise ValueError(f"Expected depth images to have 2, 3, or 4 dimensions; got {depth.shape}.")
    d = _abi.f32c(d)
    k = _intrinsics_batch(intrinsics, d.shape[0])
    out = torch.empty_like(d)
    _abi.check(_camera_op(CAMERA_ORTHOGONALIZE, _abi.ptr(d), _abi.ptr(k), k.shape[0], _abi.ptr(out), d.shape[0], d.shape[1], d.shape[2],
                          _abi.stream()), "ilx_camera_op(orthogonalize)")
    if add_last_dim:
        out = out.unsqueeze(-1)
    return out if is_batched else out.squeeze(0)


def unproject_depth(depth: torch.Tensor, intrinsics: torch.Tensor, is_ortho: bool = True) -> torch.Tensor:
    """math.py:1083-1166: depth image(s) -> points (P, 3) or (N, P, 3), pixels in column-major order (u * H + v)."""
    if not is_ortho:
        depth = orthogonalize_perspective_depth(depth, intrinsics)
    d, is_batched, _ = _depth_batch(depth)
    if d.dim() != 3:
        raise ValueError(f"Expected depth images to have dim = 2 or 3 or 4: got shape {depth.shape}")
    d = _abi.f32c(d)
    k = _intrinsics_batch(intrinsics, d.shape[0])
    n, h, w = d.shape
    out = torch.empty(n, h * w, 3, dtype=torch.float32, device=d.device)
    _abi.check(_camera_op(CAMERA_UNPROJECT, _abi.ptr(d), _abi.ptr(k), k.shape[0], _abi.ptr(out), n, h, w, _abi.stream()),
               "ilx_camera_op(unproject)")
    return out if is_batched else out.squeeze(0)


def project_points(points: torch.Tensor, intrinsics: torch.Tensor) -> torch.Tensor:
    """math.py:1170-1225: (u, v, Z) of 3-D points.  As in the reference, a (P, 3) input comes back as (1, P, 3) and an
    (N, P, 3) input is squeezed on its first dimension (its ``is_batched`` test is inverted)."""
    p = points[None] if points.dim() == 2 else points
    if p.dim() != 3:
        raise ValueError(f"Expected points to have dim = 3: got shape {points.shape}.")
    p = _abi.f32c(p)
    k = _intrinsics_batch(intrinsics, p.shape[0])
    out = torch.empty_like(p)
    _abi.check(_camera_op(CAMERA_PROJECT, _abi.ptr(p), _abi.ptr(k), k.shape[0], _abi.ptr(out), p.shape[0], p.shape[1], 0, _abi.stream()),
               "ilx_camera_op(project)")
    return out if points.dim() == 2 else out.squeeze(0)


def sample_cylinder(radius: float, h_range: tuple[float, float], size, device, *, _u: torch.Tensor | None = None) -> torch.Tensor:
    """math.py:1387-1420: points on the side of a z-aligned cylinder.  ``_u`` (2, *size): the unit-interval draws to use."""
    from . import math as M

    shape = (size,) if isinstance(size, int) else tuple(size)
    n = int(math.prod(shape))
    out = torch.empty(shape + (3,), dtype=torch.float32, device=device)
    u = None if _u is None else _abi.f32c(_u.to(device)).reshape(2, n)
    _abi.check(_cylinder(float(radius), float(h_range[0]), float(h_range[1]), _abi.ptr(u), _rng.seed(), _rng.next_draw("sampling"),
                         _abi.ptr(out), n, _abi.stream()), "ilx_sample_cylinder")
    return out


# ---- interpolation and random transforms (math.py:1646-1883): unbatched helpers of the data-generation tools, composed
# ---- from the device functions above
def quat_slerp(q1: torch.Tensor, q2: torch.Tensor, tau: float) -> torch.Tensor:
    """math.py:1646-1679 (one pair of quaternions; ``q2`` is negated in place when the arc is the long one, as there)."""
    assert isinstance(q1, torch.Tensor), "Input must be a torch tensor"
    assert isinstance(q2, torch.Tensor), "Input must be a torch tensor"
    if tau == 0.0:
        return q1
    elif tau == 1.0:
        return q2
    d = torch.dot(q1, q2)
    if abs(abs(d) - 1.0) < torch.finfo(q1.dtype).eps * 4.0:
        return q1
    if d < 0.0:
        d = -d
        q2 *= -1.0
    angle = torch.acos(torch.clamp(d, -1, 1))
    if abs(angle) < torch.finfo(q1.dtype).eps * 4.0:
        return q1
    isin = 1.0 / torch.sin(angle)
    return q1 * torch.sin((1.0 - tau) * angle) * isin + q2 * torch.sin(tau * angle) * isin


def interpolate_rotations(R1: torch.Tensor, R2: torch.Tensor, num_steps: int, axis_angle: bool = True) -> torch.Tensor:
    """math.py:1682-1731: ``num_steps`` rotations from R1 towards R2, then R2 itself."""
    from . import math as M
    from .camera_math import quat_from_matrix

    assert isinstance(R1, torch.Tensor), "Input must be a torch tensor"
    assert isinstance(R2, torch.Tensor), "Input must be a torch tensor"
    if axis_angle:
        delta_quat = quat_from_matrix(torch.matmul(R2, R1.transpose(-1, -2)))
        delta_axis_angle = M.axis_angle_from_quat(delta_quat)
        delta_angle = torch.linalg.norm(delta_axis_angle)
        rot_step_size = delta_angle / num_steps
        if delta_angle < 0.05:
            rot_steps = torch.stack([R2 for _ in range(num_steps)])
        else:
            delta_axis = delta_axis_angle / delta_angle
            angles = (torch.arange(num_steps, device=R1.device, dtype=torch.float32) * rot_step_size)
            deltas = M.matrix_from_quat(M.quat_from_angle_axis(angles, delta_axis.expand(num_steps, 3).contiguous()))
            rot_steps = torch.matmul(deltas, R1)
    else:
        q1, q2 = quat_from_matrix(R1), quat_from_matrix(R2)
        rot_steps = torch.stack([M.matrix_from_quat(quat_slerp(q1, q2, tau=float(i) / num_steps)) for i in range(num_steps)])
    return torch.cat([rot_steps, R2[None]], dim=0)


def interpolate_poses(pose_1, pose_2, num_steps=None, step_size=None, perturb=False):
    """math.py:1734-1791."""
    assert isinstance(pose_1, torch.Tensor), "Input must be a torch tensor"
    assert isinstance(pose_2, torch.Tensor), "Input must be a torch tensor"
    assert step_size is None or num_steps is None
    pos1, rot1 = unmake_pose(pose_1)
    pos2, rot2 = unmake_pose(pose_2)
    if num_steps == 0:
        return torch.cat([pos1[None], pos2[None]], dim=0), torch.cat([rot1[None], rot2[None]], dim=0), num_steps
    delta_pos = pos2 - pos1
    if num_steps is None:
        assert torch.norm(delta_pos) > 0
        num_steps = math.ceil(torch.norm(delta_pos) / step_size)
    num_steps += 1
    assert num_steps >= 2
    pos_step_size = delta_pos / num_steps
    grid = torch.arange(num_steps, dtype=torch.float32, device=pose_1.device)
    if perturb:
        grid[1:-1] += torch.rand(num_steps - 2, device=pose_1.device) - 0.5
    pos_steps = torch.cat([pos1 + grid[:, None] * pos_step_size, pos2[None]], dim=0)
    rot_steps = interpolate_rotations(R1=rot1, R2=rot2, num_steps=num_steps, axis_angle=True)
    return make_pose(pos_steps, rot_steps), num_steps - 1


def generate_random_rotation(rot_boundary=(2 * math.pi), device="cuda") -> torch.Tensor:
    """math.py:1824-1849: Rz Ry Rx of three uniform angles in [0, rot_boundary)."""
    from . import math as M
    angles = M.sample_uniform(0.0, float(rot_boundary), (1, 3), device)
    # Rz @ Ry @ Rx is what quat_from_euler_xyz composes (extrinsic x-y-z)
    return M.matrix_from_quat(M.quat_from_euler_xyz(angles[:, 0], angles[:, 1], angles[:, 2]))[0]


def generate_random_translation(pos_boundary=1, device="cuda") -> torch.Tensor:
    """math.py:1852-1862."""
    from . import math as M

    return M.sample_uniform(-float(pos_boundary), float(pos_boundary), (1, 3), device)[0]


def generate_random_transformation_matrix(pos_boundary=1, rot_boundary=(2 * math.pi), device="cuda") -> torch.Tensor:
    """math.py:1865-1883."""
    return make_pose(generate_random_translation(pos_boundary, device), generate_random_rotation(rot_boundary, device))
