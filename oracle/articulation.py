"""Oracle for the derived ``ArticulationData`` quantities the terms read (test infrastructure, CPU only).

Follows ``source/isaaclab/isaaclab/assets/articulation/articulation_data.py`` of the reference (cited as
``articulation_data.py:<line>``).
"""

from __future__ import annotations

import torch

from . import mathx as M


def root_state_from_view(root_transforms_xyzw, root_velocities):
    """articulation_data.py:263-279: cat(pos, quat xyzw->wxyz, lin vel, ang vel) -> (N, 13)."""
    pos = root_transforms_xyzw[:, 0:3]
    quat = M.convert_quat(root_transforms_xyzw[:, 3:7], to="wxyz")
    return torch.cat([pos, quat, root_velocities], dim=-1)


def gravity_dir(gravity=(0.0, 0.0, -9.81)):
    """articulation_data.py:53-59: normalised gravity vector (3,)."""
    g = torch.tensor(gravity, dtype=torch.float32)
    return M.normalize(g.unsqueeze(0)).squeeze(0)


def root_lin_vel_b(root_state):
    """articulation_data.py:502-509."""
    return M.quat_rotate_inverse(root_state[:, 3:7], root_state[:, 7:10])


def root_ang_vel_b(root_state):
    """articulation_data.py:511-518."""
    return M.quat_rotate_inverse(root_state[:, 3:7], root_state[:, 10:13])


def projected_gravity_b(root_state, gravity_vec):
    """articulation_data.py:411-414; ``gravity_vec`` is (3,) and is repeated per env."""
    g = gravity_vec.unsqueeze(0).expand(root_state.shape[0], 3)
    return M.quat_rotate_inverse(root_state[:, 3:7], g)


def heading_w(root_state):
    """articulation_data.py:416-425: atan2 of the rotated forward vector (1,0,0)."""
    fwd = torch.tensor([1.0, 0.0, 0.0]).unsqueeze(0).expand(root_state.shape[0], 3)
    f = M.quat_apply(root_state[:, 3:7], fwd)
    return torch.atan2(f[:, 1], f[:, 0])


def joint_acc(joint_vel, previous_joint_vel, time_elapsed: float):
    """articulation_data.py:445-455: finite difference; returns (acc, new previous_joint_vel)."""
    acc = (joint_vel - previous_joint_vel) / M.f32(time_elapsed)
    return acc, joint_vel.clone()


def soft_joint_pos_limits(joint_pos_limits, factor: float):
    """articulation.py:1230-1237: mean +- 0.5 * range * factor, shape (N, J, 2)."""
    mean = (joint_pos_limits[..., 0] + joint_pos_limits[..., 1]) / 2
    rng = joint_pos_limits[..., 1] - joint_pos_limits[..., 0]
    lo = mean - 0.5 * rng * factor
    hi = mean + 0.5 * rng * factor
    return torch.stack([lo, hi], dim=-1)
