"""Set-up time rotation helpers of the camera sensors (``isaaclab/utils/math.py:282-412, 1428-1563``).

These run when a camera is created or re-posed (a handful of rows), not per step, and are compositions of torch
operations on the tensors' own device; the per-step camera work is the HIP kernel behind
:class:`isaaclab_amd.sensors.ray_caster_camera.RayCasterCamera`.
"""

from __future__ import annotations

import math

import torch

from .math import matrix_from_quat


def quat_from_matrix(matrix: torch.Tensor) -> torch.Tensor:
    """math.py:295-349: rotation matrices (..., 3, 3) -> quaternions (w, x, y, z); of the four algebraically equal
    candidates the one with the largest denominator is returned."""
    if matrix.size(-1) != 3 or matrix.size(-2) != 3:
        raise ValueError(f"Invalid rotation matrix shape {matrix.shape}.")
    batch = matrix.shape[:-2]
    m = matrix.reshape(batch + (9,))
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(m, dim=-1)
    arg = torch.stack([1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22, 1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22], dim=-1)
    q_abs = torch.where(arg > 0, torch.sqrt(arg.clamp(min=0)), torch.zeros_like(arg))
    cand = torch.stack([
        torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
        torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], dim=-1),
        torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], dim=-1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], dim=-1)], dim=-2)
    floor = torch.tensor(0.1, dtype=q_abs.dtype, device=q_abs.device)
    cand = cand / (2.0 * q_abs[..., None].max(floor))
    pick = q_abs.argmax(dim=-1)
    return torch.gather(cand, -2, pick[..., None, None].expand(batch + (1, 4))).reshape(batch + (4,))


def _axis_rotation(axis: str, angle: torch.Tensor) -> torch.Tensor:
    c, s, one, zero = torch.cos(angle), torch.sin(angle), torch.ones_like(angle), torch.zeros_like(angle)
    flat = {"X": (one, zero, zero, zero, c, -s, zero, s, c), "Y": (c, zero, s, zero, one, zero, -s, zero, c),
            "Z": (c, -s, zero, s, c, zero, zero, zero, one)}.get(axis)
    if flat is None:
        raise ValueError("letter must be either X, Y or Z.")
    return torch.stack(flat, -1).reshape(angle.shape + (3, 3))


def matrix_from_euler(euler_angles: torch.Tensor, convention: str) -> torch.Tensor:
    """math.py:382-412: Euler angles (..., 3) -> rotation matrices, e.g. "XYZ" = about x, then y, then z."""
    if euler_angles.dim() == 0 or euler_angles.shape[-1] != 3:
        raise ValueError("Invalid input euler angles.")
    if len(convention) != 3:
        raise ValueError("Convention must have 3 letters.")
    if convention[1] in (convention[0], convention[2]):
        raise ValueError(f"Invalid convention {convention}.")
    mats = [_axis_rotation(c, e) for c, e in zip(convention, torch.unbind(euler_angles, -1))]
    return torch.matmul(torch.matmul(mats[0], mats[1]), mats[2])


def convert_camera_frame_orientation_convention(orientation: torch.Tensor, origin: str = "opengl", target: str = "ros") -> torch.Tensor:
    """math.py:1428-1510: "opengl" (-z forward, +y up), "ros" (+z forward, -y up), "world" (+x forward, +z up)."""
    if target == origin:
        return orientation.clone()
    world_to_gl = matrix_from_euler(torch.tensor([math.pi / 2, -math.pi / 2, 0], device=orientation.device), "XYZ")
    if origin == "ros":
        rotm = matrix_from_quat(orientation.contiguous())
        rotm[:, :, 2] = -rotm[:, :, 2]
        rotm[:, :, 1] = -rotm[:, :, 1]
        quat_gl = quat_from_matrix(rotm)
    elif origin == "world":
        quat_gl = quat_from_matrix(torch.matmul(matrix_from_quat(orientation.contiguous()), world_to_gl))
    else:
        quat_gl = orientation
    if target == "ros":
        rotm = matrix_from_quat(quat_gl.contiguous())
        rotm[:, :, 2] = -rotm[:, :, 2]
        rotm[:, :, 1] = -rotm[:, :, 1]
        return quat_from_matrix(rotm)
    if target == "world":
        return quat_from_matrix(torch.matmul(matrix_from_quat(quat_gl.contiguous()), world_to_gl.T))
    return quat_gl.clone()


def create_rotation_matrix_from_view(eyes: torch.Tensor, targets: torch.Tensor, up_axis: str = "Z", device="cpu") -> torch.Tensor:
    """math.py:1513-1563: look-at rotation in the opengl convention (-z forward, +y up)."""
    if up_axis not in ("Y", "Z"):
        raise ValueError(f"Invalid up axis: {up_axis}. Valid options are 'Y' and 'Z'.")
    up = torch.tensor((0, 1, 0) if up_axis == "Y" else (0, 0, 1), device=device, dtype=torch.float32).repeat(eyes.shape[0], 1)
    unit = lambda v: torch.nn.functional.normalize(v, eps=1e-5)  # noqa: E731
    z_axis = -unit(targets - eyes)
    x_axis = unit(torch.cross(up, z_axis, dim=1))
    y_axis = unit(torch.cross(z_axis, x_axis, dim=1))
    is_close = torch.isclose(x_axis, torch.tensor(0.0, device=x_axis.device), atol=5e-3).all(dim=1, keepdim=True)
    if is_close.any():
        x_axis = torch.where(is_close, unit(torch.cross(y_axis, z_axis, dim=1)), x_axis)
    return torch.cat((x_axis[:, None, :], y_axis[:, None, :], z_axis[:, None, :]), dim=1).transpose(1, 2)
