"""Oracle for the command terms (test infrastructure, CPU only).

Follows ``source/isaaclab/isaaclab/managers/command_manager.py:120-187`` (``CommandTerm.reset / compute /
_resample``), ``envs/mdp/commands/velocity_command.py:111-160`` (``UniformVelocityCommand``) and
``envs/mdp/commands/pose_command.py:92-127`` (``UniformPoseCommand``).

Random draws are INPUTS: torch's CPU generator cannot be reproduced on the GPU, so the values a resample would
draw are passed per env (``samples``) and consumed only by the envs that resample.  ``draw_*_samples`` produces
them with the same ``uniform_`` calls and in the same order as the reference, for envs ``env_ids``.
"""

from __future__ import annotations

from dataclasses import dataclass, field

import torch

from . import mathx as M


@dataclass
class VelocityCommandCfg:
    """velocity_env_cfg.py:94-105 defaults (Isaac-Velocity-Rough-*)."""

    resampling_time_range: tuple = (10.0, 10.0)
    rel_standing_envs: float = 0.02
    rel_heading_envs: float = 1.0
    heading_command: bool = True
    heading_control_stiffness: float = 0.5
    lin_vel_x: tuple = (-1.0, 1.0)
    lin_vel_y: tuple = (-1.0, 1.0)
    ang_vel_z: tuple = (-1.0, 1.0)
    heading: tuple = (-3.141592653589793, 3.141592653589793)


@dataclass
class VelocityCommandState:
    vel_command_b: torch.Tensor
    heading_target: torch.Tensor
    is_heading_env: torch.Tensor
    is_standing_env: torch.Tensor
    time_left: torch.Tensor
    command_counter: torch.Tensor
    metric_error_vel_xy: torch.Tensor
    metric_error_vel_yaw: torch.Tensor

    @staticmethod
    def zeros(n):
        return VelocityCommandState(torch.zeros(n, 3), torch.zeros(n), torch.zeros(n, dtype=torch.bool),
                                    torch.zeros(n, dtype=torch.bool), torch.zeros(n), torch.zeros(n, dtype=torch.long),
                                    torch.zeros(n), torch.zeros(n))

    def clone(self):
        return VelocityCommandState(*[getattr(self, f).clone() for f in self.__dataclass_fields__])


def draw_velocity_samples(cfg: VelocityCommandCfg, n: int, env_ids, generator=None):
    """The draws of ``_resample`` + ``_resample_command`` for ``env_ids`` in reference order, scattered into
    (n, 7) = [time_left, vx, vy, wz, heading, u_heading, u_standing]."""
    k = len(env_ids)
    out = torch.zeros(n, 7)
    if k == 0:
        return out
    t = torch.empty(k).uniform_(*cfg.resampling_time_range, generator=generator)
    r = torch.empty(k)
    vals = [t]
    for rng in (cfg.lin_vel_x, cfg.lin_vel_y, cfg.ang_vel_z):
        vals.append(r.uniform_(*rng, generator=generator).clone())
    if cfg.heading_command:
        vals.append(r.uniform_(*cfg.heading, generator=generator).clone())
        vals.append(r.uniform_(0.0, 1.0, generator=generator).clone())
    else:
        vals += [torch.zeros(k), torch.ones(k) * 2.0]
    vals.append(r.uniform_(0.0, 1.0, generator=generator).clone())
    out[env_ids] = torch.stack(vals, dim=-1)
    return out


def velocity_resample(cfg: VelocityCommandCfg, s: VelocityCommandState, mask, samples):
    """command_manager.py:172-187 + velocity_command.py:123-138 for the envs in ``mask``."""
    s.time_left = torch.where(mask, samples[:, 0], s.time_left)
    s.command_counter = s.command_counter + mask.long()
    s.vel_command_b = torch.where(mask.unsqueeze(-1), samples[:, 1:4], s.vel_command_b)
    if cfg.heading_command:
        s.heading_target = torch.where(mask, samples[:, 4], s.heading_target)
        s.is_heading_env = torch.where(mask, samples[:, 5] <= float(M.f32(cfg.rel_heading_envs)), s.is_heading_env)
    s.is_standing_env = torch.where(mask, samples[:, 6] <= float(M.f32(cfg.rel_standing_envs)), s.is_standing_env)


def velocity_reset(cfg, s: VelocityCommandState, reset_mask, samples):
    """command_manager.py:120-149: log + clear metrics, zero the counter, resample.
    Returns the logged means (over the reset envs) as a dict, like ``extras``."""
    extras = {}
    if reset_mask.any():
        extras["error_vel_xy"] = torch.mean(s.metric_error_vel_xy[reset_mask]).item()
        extras["error_vel_yaw"] = torch.mean(s.metric_error_vel_yaw[reset_mask]).item()
    s.metric_error_vel_xy = torch.where(reset_mask, torch.zeros(()), s.metric_error_vel_xy)
    s.metric_error_vel_yaw = torch.where(reset_mask, torch.zeros(()), s.metric_error_vel_yaw)
    s.command_counter = torch.where(reset_mask, torch.zeros((), dtype=torch.long), s.command_counter)
    velocity_resample(cfg, s, reset_mask, samples)
    return extras


def velocity_compute(cfg: VelocityCommandCfg, s: VelocityCommandState, dt: float, step_dt: float, root_lin_vel_b,
                     root_ang_vel_b, heading_w, samples):
    """command_manager.py:151-166 with the velocity term's hooks.  Returns the resample mask."""
    # _update_metrics (velocity_command.py:111-121)
    max_command_step = cfg.resampling_time_range[1] / step_dt
    dx = s.vel_command_b[:, 0] - root_lin_vel_b[:, 0]
    dy = s.vel_command_b[:, 1] - root_lin_vel_b[:, 1]
    s.metric_error_vel_xy = s.metric_error_vel_xy + M.norm2(dx, dy) / M.f32(max_command_step)
    s.metric_error_vel_yaw = s.metric_error_vel_yaw + torch.abs(s.vel_command_b[:, 2] - root_ang_vel_b[:, 2]) / M.f32(max_command_step)
    # time left, resample
    s.time_left = s.time_left - M.f32(dt)
    mask = s.time_left <= 0.0
    velocity_resample(cfg, s, mask, samples)
    # _update_command (velocity_command.py:140-160)
    if cfg.heading_command:
        err = M.wrap_to_pi(s.heading_target - heading_w)
        wz = torch.clip(M.f32(cfg.heading_control_stiffness) * err, min=float(M.f32(cfg.ang_vel_z[0])),
                        max=float(M.f32(cfg.ang_vel_z[1])))
        s.vel_command_b[:, 2] = torch.where(s.is_heading_env, wz, s.vel_command_b[:, 2])
    s.vel_command_b = torch.where(s.is_standing_env.unsqueeze(-1), torch.zeros(()), s.vel_command_b)
    return mask


# ---- pose command ----------------------------------------------------------------------------------
@dataclass
class PoseCommandCfg:
    """reach_env_cfg.py:88-104 (Isaac-Reach-Franka-v0 after franka/joint_pos_env_cfg.py overrides)."""

    resampling_time_range: tuple = (4.0, 4.0)
    pos_x: tuple = (0.35, 0.65)
    pos_y: tuple = (-0.2, 0.2)
    pos_z: tuple = (0.15, 0.5)
    roll: tuple = (0.0, 0.0)
    pitch: tuple = (3.141592653589793, 3.141592653589793)
    yaw: tuple = (-3.14, 3.14)
    make_quat_unique: bool = False


def draw_pose_samples(cfg: PoseCommandCfg, n: int, env_ids, generator=None):
    """(n, 7) = [time_left, x, y, z, roll, pitch, yaw] drawn in the order of pose_command.py:107-120."""
    k = len(env_ids)
    out = torch.zeros(n, 7)
    if k == 0:
        return out
    vals = [torch.empty(k).uniform_(*cfg.resampling_time_range, generator=generator)]
    r = torch.empty(k)
    for rng in (cfg.pos_x, cfg.pos_y, cfg.pos_z):
        vals.append(r.uniform_(*rng, generator=generator).clone())
    for rng in (cfg.roll, cfg.pitch, cfg.yaw):
        vals.append(torch.zeros(k).uniform_(*rng, generator=generator))
    out[env_ids] = torch.stack(vals, dim=-1)
    return out


def pose_metrics(pose_command_b, root_state_w, body_pos_w, body_quat_w):
    """pose_command.py:92-108: (position_error, orientation_error, pose_command_w)."""
    pw, qw = M.combine_frame_transforms(root_state_w[:, :3], root_state_w[:, 3:7], pose_command_b[:, :3], pose_command_b[:, 3:])
    pe, re = M.compute_pose_error(pw, qw, body_pos_w, body_quat_w)
    return M.norm3(pe[:, 0], pe[:, 1], pe[:, 2]), M.norm3(re[:, 0], re[:, 1], re[:, 2]), torch.cat([pw, qw], dim=-1)


def pose_compute(cfg: PoseCommandCfg, pose_command_b, time_left, command_counter, dt: float, samples):
    """command_manager.py:151-166 for the pose term (metrics are overwritten, not accumulated, so they are
    evaluated separately by :func:`pose_metrics`).  Returns (pose_command_b, time_left, counter, mask)."""
    time_left = time_left - M.f32(dt)
    mask = time_left <= 0.0
    time_left = torch.where(mask, samples[:, 0], time_left)
    command_counter = command_counter + mask.long()
    quat = M.quat_from_euler_xyz(samples[:, 4], samples[:, 5], samples[:, 6])
    if cfg.make_quat_unique:
        quat = M.quat_unique(quat)
    new = torch.cat([samples[:, 1:4], quat], dim=-1)
    pose_command_b = torch.where(mask.unsqueeze(-1), new, pose_command_b)
    return pose_command_b, time_left, command_counter, mask


# ---- 2-D pose command (envs/mdp/commands/pose_2d_command.py) ------------------------------------------------------
def heading_w(root_quat):
    """articulation_data.py:416-425."""
    fwd = M.quat_apply(root_quat, torch.tensor([[1.0, 0.0, 0.0]]).repeat(root_quat.shape[0], 1))
    return torch.atan2(fwd[:, 1], fwd[:, 0])


def pose2d_compute(state: dict, root_state_w, dt: float, samples, picks=None, *, simple_heading: bool, env_origins=None,
                   default_root_z=None, valid_targets=None, terrain_levels=None, terrain_types=None):
    """One ``CommandTerm.compute`` (command_manager.py:151-166) of the 2-D pose command (pose_2d_command.py:88-128, 177-206).

    ``state``: pos_command_w, heading_command_w, time_left, command_counter (updated in place); ``samples`` (n,4) holds the
    drawn values (time left, x, y, heading) and ``picks`` (n) the patch indices of the terrain-based variant.
    Returns (command (n,4), error_pos_2d, error_heading)."""
    root_pos, root_quat = root_state_w[:, :3], root_state_w[:, 3:7]
    hw = heading_w(root_quat)
    err_pos = torch.norm(state["pos_command_w"][:, :2] - root_pos[:, :2], dim=1)
    err_head = torch.abs(M.wrap_to_pi(state["heading_command_w"] - hw))
    state["time_left"] -= dt
    ids = (state["time_left"] <= 0.0).nonzero().flatten()
    if len(ids) > 0:
        pose2d_resample(state, ids, root_pos, hw, samples, picks, simple_heading=simple_heading, env_origins=env_origins,
                        default_root_z=default_root_z, valid_targets=valid_targets, terrain_levels=terrain_levels,
                        terrain_types=terrain_types)
    target_vec = state["pos_command_w"] - root_pos
    pos_b = M.quat_rotate_inverse(M.yaw_quat(root_quat), target_vec)
    head_b = M.wrap_to_pi(state["heading_command_w"] - hw)
    return torch.cat([pos_b, head_b.unsqueeze(1)], dim=1), err_pos, err_head


def pose2d_resample(state, ids, root_pos, hw, samples, picks, *, simple_heading, env_origins, default_root_z, valid_targets,
                    terrain_levels, terrain_types):
    """command_manager.py:172-187 + pose_2d_command.py:93-122 / :177-206."""
    state["time_left"][ids] = samples[ids, 0]
    state["command_counter"][ids] += 1
    if valid_targets is None:
        pos = env_origins[ids].clone()
        pos[:, 0] += samples[ids, 1]
        pos[:, 1] += samples[ids, 2]
    else:
        pos = valid_targets[terrain_levels[ids], terrain_types[ids], picks[ids]].clone()
    pos[:, 2] += default_root_z[ids]
    state["pos_command_w"][ids] = pos
    if simple_heading:
        vec = pos - root_pos[ids]
        direction = torch.atan2(vec[:, 1], vec[:, 0])
        flipped = M.wrap_to_pi(direction + torch.pi)
        to_dir = M.wrap_to_pi(direction - hw[ids]).abs()
        to_flipped = M.wrap_to_pi(flipped - hw[ids]).abs()
        state["heading_command_w"][ids] = torch.where(to_dir < to_flipped, direction, flipped)
    else:
        state["heading_command_w"][ids] = samples[ids, 3]


# ---- normal velocity command (envs/mdp/commands/velocity_command.py:218-287) -------------------------------------------
def normal_velocity_compute(state: dict, root_state_w, dt: float, samples, *, zero_prob, rel_standing_envs: float,
                            max_command_step: float):
    """One ``CommandTerm.compute`` (command_manager.py:151-166) of NormalVelocityCommand.

    ``state``: vel_command_b (n,3), is_zero (n,3) bool, is_standing (n) bool, time_left, command_counter, error_vel_xy,
    error_vel_yaw (all updated in place).  ``samples`` (n,11): time left, the three normal values, three sign coins, three
    zero coins, the standing coin."""
    q = root_state_w[:, 3:7]
    lin_b, ang_b = M.quat_rotate_inverse(q, root_state_w[:, 7:10]), M.quat_rotate_inverse(q, root_state_w[:, 10:13])
    v = state["vel_command_b"]
    state["error_vel_xy"] += torch.norm(v[:, :2] - lin_b[:, :2], dim=-1) / max_command_step   # velocity_command.py:111-121
    state["error_vel_yaw"] += torch.abs(v[:, 2] - ang_b[:, 2]) / max_command_step
    state["time_left"] -= dt
    ids = (state["time_left"] <= 0.0).nonzero().flatten()
    if len(ids) > 0:
        normal_velocity_resample(state, ids, samples, zero_prob=zero_prob, rel_standing_envs=rel_standing_envs)
    v[state["is_standing"]] = 0.0                                                                # :274-287
    v[state["is_zero"]] = 0.0
    return v


def normal_velocity_resample(state, ids, samples, *, zero_prob, rel_standing_envs):
    """command_manager.py:172-187 + velocity_command.py:252-272."""
    state["time_left"][ids] = samples[ids, 0]
    state["command_counter"][ids] += 1
    one, minus = torch.tensor(1.0), torch.tensor(-1.0)
    for k in range(3):
        state["vel_command_b"][ids, k] = samples[ids, 1 + k] * torch.where(samples[ids, 4 + k] <= 0.5, one, minus)
        state["is_zero"][ids, k] = samples[ids, 7 + k] <= zero_prob[k]
    state["is_standing"][ids] = samples[ids, 10] <= rel_standing_envs
