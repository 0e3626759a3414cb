"""Oracle for the reset / push event terms (test infrastructure, CPU only).

Follows ``source/isaaclab/isaaclab/envs/mdp/events.py:610-866`` of the reference.  The reference draws its random
numbers with ``torch.rand``; here the unit-interval draws ``u`` are an argument, and ``sample`` maps them to a range the
way ``isaaclab.utils.math.sample_uniform`` does (math.py:1313-1331: ``rand * (upper - lower) + lower``).
"""

from __future__ import annotations

import torch

from . import mathx as M

AXES = ("x", "y", "z", "roll", "pitch", "yaw")


def ranges(d: dict) -> torch.Tensor:
    """events.py:663-664: (6, 2) table of the per-axis ranges, (0, 0) for the axes not named."""
    return torch.tensor([d.get(k, (0.0, 0.0)) for k in AXES], dtype=torch.float32)


def sample(u, lower, upper):
    return u * (upper - lower) + lower


def reset_root_state_uniform(default_root_state, env_origins, pose_range: dict, velocity_range: dict, u_pose, u_vel):
    """events.py:638-680 for the selected rows: returns (pose (k,7), velocity (k,6))."""
    pr, vr = ranges(pose_range), ranges(velocity_range)
    rs = sample(u_pose, pr[:, 0], pr[:, 1])
    positions = default_root_state[:, 0:3] + env_origins + rs[:, 0:3]
    delta = M.quat_from_euler_xyz(rs[:, 3], rs[:, 4], rs[:, 5])
    orientations = M.quat_mul(default_root_state[:, 3:7], delta)
    velocities = default_root_state[:, 7:13] + sample(u_vel, vr[:, 0], vr[:, 1])
    return torch.cat([positions, orientations], dim=-1), velocities


def reset_joints(default_pos, default_vel, pos_limits, vel_limits, position_range, velocity_range, u_pos, u_vel, scale: bool):
    """events.py:802-832 (``scale``) / :835-866 (offset) for the selected rows: returns (joint_pos, joint_vel)."""
    sp, sv = sample(u_pos, *position_range), sample(u_vel, *velocity_range)
    pos = default_pos * sp if scale else default_pos + sp
    vel = default_vel * sv if scale else default_vel + sv
    pos = torch.clamp(pos, pos_limits[..., 0], pos_limits[..., 1])
    vel = torch.clamp(vel, -vel_limits, vel_limits)
    return pos, vel


def push_by_setting_velocity(root_vel_w, velocity_range: dict, u):
    """events.py:610-635 for the selected rows."""
    vr = ranges(velocity_range)
    return root_vel_w + sample(u, vr[:, 0], vr[:, 1])


def reset_root_state_with_random_orientation(default_root_state, env_origins, pose_range: dict, velocity_range: dict, u_pos, z, u_vel):
    """events.py:683-732 for the selected rows; ``z`` (k,4) are the standard normal draws of ``random_orientation``
    (math.py:1251-1267: ``normalize(randn(num, 4), eps=1e-12)``)."""
    pr, vr = ranges(pose_range)[:3], ranges(velocity_range)
    positions = default_root_state[:, 0:3] + env_origins + sample(u_pos, pr[:, 0], pr[:, 1])
    orientations = torch.nn.functional.normalize(z, p=2.0, dim=-1, eps=1e-12)
    velocities = default_root_state[:, 7:13] + sample(u_vel, vr[:, 0], vr[:, 1])
    return torch.cat([positions, orientations], dim=-1), velocities


def apply_external_force_torque(force_range, torque_range, u_force, u_torque):
    """events.py:601-604."""
    return sample(u_force, *force_range), sample(u_torque, *torque_range)


def reset_scene_to_default(default_root_state, env_origins, default_joint_pos, default_joint_vel):
    """events.py:922-933 for the selected rows of one articulation."""
    root = default_root_state.clone()
    root[:, 0:3] += env_origins
    return root, default_joint_pos.clone(), default_joint_vel.clone()


def reset_root_state_from_terrain(default_root_state, valid_positions, levels, types, picks, pose_range: dict, velocity_range: dict, u):
    """events.py:735-799 for the full env set: returns (pose (n,7), velocity (n,6)).  ``u`` (n,9): three angle draws, then six
    velocity draws; ``picks`` (n): the patch chosen per env."""
    pr, vr = ranges(pose_range), ranges(velocity_range)
    positions = valid_positions[levels, types, picks] + default_root_state[:, :3]
    angles = sample(u[:, :3], pr[3:, 0], pr[3:, 1])
    orientations = M.quat_from_euler_xyz(angles[:, 0], angles[:, 1], angles[:, 2])
    velocities = default_root_state[:, 7:13] + sample(u[:, 3:], vr[:, 0], vr[:, 1])
    return torch.cat([positions, orientations], dim=-1), velocities
