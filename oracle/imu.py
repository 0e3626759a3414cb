"""CPU restatement of the IMU sensor refresh (test infrastructure only; never imported by the product).

Follows ``isaaclab/sensors/imu/imu.py:142-182``.  Pinned by tests/golden/imu.npz, recorded from the reference class.
"""

from __future__ import annotations

import torch

from . import mathx as M


def imu_refresh(transforms, velocities, coms, offset_pos, offset_quat, gravity_bias, prev_lin_vel_w, prev_ang_vel_w, dt: float):
    """Returns (pos_w, quat_w, lin_vel_b, ang_vel_b, lin_acc_b, ang_acc_b, lin_vel_w, ang_vel_w) for all sensors."""
    n = transforms.shape[0]
    pos, q_xyzw = transforms[:, :3], transforms[:, 3:]
    q = torch.cat([q_xyzw[:, 3:], q_xyzw[:, :3]], dim=1)
    off_p, off_q = offset_pos.expand(n, 3), offset_quat.expand(n, 4)
    pos_w = pos + M.quat_rotate(q, off_p)
    quat_w = M.quat_mul(q, off_q)
    lin, ang = velocities[:, :3], velocities[:, 3:]
    lin = lin + torch.linalg.cross(ang, M.quat_rotate(q, off_p - coms[:, :3]), dim=-1)
    lin_acc = (lin - prev_lin_vel_w) / dt + gravity_bias
    ang_acc = (ang - prev_ang_vel_w) / dt
    rot = lambda v: M.quat_rotate_inverse(quat_w, v)  # noqa: E731
    return pos_w, quat_w, rot(lin), rot(ang), rot(lin_acc), rot(ang_acc), lin, ang
