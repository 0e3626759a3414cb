"""Oracle for the MDP term functions (test infrastructure, CPU only).

Each function restates one term of the reference, taking plain tensors instead of the ``env`` object:

* ``R/...`` = ``source/isaaclab/isaaclab/envs/mdp/rewards.py``
* ``T/...`` = ``source/isaaclab/isaaclab/envs/mdp/terminations.py``
* ``O/...`` = ``source/isaaclab/isaaclab/envs/mdp/observations.py``
* ``V/...`` = ``source/isaaclab_tasks/isaaclab_tasks/manager_based/locomotion/velocity/mdp/rewards.py``
* ``RE/...`` = ``source/isaaclab_tasks/isaaclab_tasks/manager_based/manipulation/reach/mdp/rewards.py``
* ``CP/...`` = ``source/isaaclab_tasks/isaaclab_tasks/manager_based/classic/cartpole/mdp/rewards.py``

Joint / body subsets are passed as ``ids`` (``slice(None)`` or a list), exactly what a resolved ``SceneEntityCfg``
carries.  Sums over joints use ``torch.sum`` like the reference; decisions use the exact primitives of
:mod:`oracle.mathx`.
"""

from __future__ import annotations

import torch

from . import mathx as M

ALL = slice(None)


def _hist_max_norm(hist, body_ids):
    """max over the history axis of |F| for the selected bodies: (N,H,B,3) -> (N,len(ids))."""
    f = hist[:, :, body_ids]
    return torch.max(M.norm3(f[..., 0], f[..., 1], f[..., 2]), dim=1)[0]


# ---- rewards: general / root ---------------------------------------------------------------------
def is_alive(terminated):
    """R:31-33."""
    return (~terminated).float()


def is_terminated(terminated):
    """R:36-38."""
    return terminated.float()


def is_terminated_term(term_dones: list, time_outs):
    """R:41-68: number of active (non time-out) terminations, zero when the episode timed out."""
    buf = torch.zeros(time_outs.shape[0])
    for t in term_dones:
        buf = buf + t
    return (buf * (~time_outs)).float()


def lin_vel_z_l2(root_lin_vel_b):
    """R:76-80."""
    return torch.square(root_lin_vel_b[:, 2])


def ang_vel_xy_l2(root_ang_vel_b):
    """R:83-87."""
    return torch.sum(torch.square(root_ang_vel_b[:, :2]), dim=1)


def flat_orientation_l2(projected_gravity_b):
    """R:90-97."""
    return torch.sum(torch.square(projected_gravity_b[:, :2]), dim=1)


def base_height_l2(root_pos_w, target_height: float, ray_hits_w=None):
    """R:100-122."""
    if ray_hits_w is not None:
        target = target_height + torch.mean(ray_hits_w[..., 2], dim=1)
    else:
        target = target_height
    return torch.square(root_pos_w[:, 2] - target)


def body_lin_acc_l2(body_lin_acc_w, body_ids=ALL):
    """R:125-128."""
    a = body_lin_acc_w[:, body_ids, :]
    return torch.sum(M.norm3(a[..., 0], a[..., 1], a[..., 2]), dim=1)


# ---- rewards: joints -----------------------------------------------------------------------------
def joint_torques_l2(applied_torque, joint_ids=ALL):
    """R:136-143."""
    return torch.sum(torch.square(applied_torque[:, joint_ids]), dim=1)


def joint_vel_l1(joint_vel, joint_ids=ALL):
    """R:146-150."""
    return torch.sum(torch.abs(joint_vel[:, joint_ids]), dim=1)


def joint_vel_l2(joint_vel, joint_ids=ALL):
    """R:153-160."""
    return torch.sum(torch.square(joint_vel[:, joint_ids]), dim=1)


def joint_acc_l2(joint_acc, joint_ids=ALL):
    """R:163-170."""
    return torch.sum(torch.square(joint_acc[:, joint_ids]), dim=1)


def joint_deviation_l1(joint_pos, default_joint_pos, joint_ids=ALL):
    """R:173-179."""
    return torch.sum(torch.abs(joint_pos[:, joint_ids] - default_joint_pos[:, joint_ids]), dim=1)


def joint_pos_limits(joint_pos, soft_joint_pos_limits, joint_ids=ALL):
    """R:182-196."""
    out = -(joint_pos[:, joint_ids] - soft_joint_pos_limits[:, joint_ids, 0]).clip(max=0.0)
    out = out + (joint_pos[:, joint_ids] - soft_joint_pos_limits[:, joint_ids, 1]).clip(min=0.0)
    return torch.sum(out, dim=1)


def joint_vel_limits(joint_vel, soft_joint_vel_limits, soft_ratio: float, joint_ids=ALL):
    """R:199-218."""
    out = torch.abs(joint_vel[:, joint_ids]) - soft_joint_vel_limits[:, joint_ids] * soft_ratio
    return torch.sum(out.clip(min=0.0, max=1.0), dim=1)


def applied_torque_limits(applied_torque, computed_torque, joint_ids=ALL):
    """R:226-242."""
    return torch.sum(torch.abs(applied_torque[:, joint_ids] - computed_torque[:, joint_ids]), dim=1)


def action_rate_l2(action, prev_action):
    """R:245-247."""
    return torch.sum(torch.square(action - prev_action), dim=1)


def action_l2(action):
    """R:250-252."""
    return torch.sum(torch.square(action), dim=1)


# ---- rewards: contacts ---------------------------------------------------------------------------
def undesired_contacts(net_forces_w_history, threshold: float, body_ids):
    """R:260-268: number of selected bodies whose history-max force exceeds the threshold (int64)."""
    is_contact = _hist_max_norm(net_forces_w_history, body_ids) > float(M.f32(threshold))
    return torch.sum(is_contact, dim=1)


def contact_forces(net_forces_w_history, threshold: float, body_ids):
    """R:271-279."""
    violation = _hist_max_norm(net_forces_w_history, body_ids) - M.f32(threshold)
    return torch.sum(violation.clip(min=0.0), dim=1)


# ---- rewards: velocity tracking ------------------------------------------------------------------
def track_lin_vel_xy_exp(command, root_lin_vel_b, std: float):
    """R:287-298."""
    err = torch.sum(torch.square(command[:, :2] - root_lin_vel_b[:, :2]), dim=1)
    return torch.exp(-err / M.f32(std**2))


def track_ang_vel_z_exp(command, root_ang_vel_b, std: float):
    """R:301-309."""
    err = torch.square(command[:, 2] - root_ang_vel_b[:, 2])
    return torch.exp(-err / M.f32(std**2))


def feet_air_time(current_contact_time, last_air_time, command, step_dt: float, threshold: float, body_ids):
    """V:25-44."""
    from .contact import compute_first_contact

    first_contact = compute_first_contact(current_contact_time, step_dt)[:, body_ids]
    reward = torch.sum((last_air_time[:, body_ids] - M.f32(threshold)) * first_contact, dim=1)
    reward = reward * (M.norm2(command[:, 0], command[:, 1]) > float(M.f32(0.1)))
    return reward


def feet_air_time_positive_biped(current_air_time, current_contact_time, command, threshold: float, body_ids):
    """V:47-66."""
    air = current_air_time[:, body_ids]
    con = current_contact_time[:, body_ids]
    in_contact = con > 0.0
    in_mode_time = torch.where(in_contact, con, air)
    single_stance = torch.sum(in_contact.int(), dim=1) == 1
    reward = torch.min(torch.where(single_stance.unsqueeze(-1), in_mode_time, torch.zeros(())), dim=1)[0]
    reward = torch.clamp(reward, max=float(M.f32(threshold)))
    reward = reward * (M.norm2(command[:, 0], command[:, 1]) > float(M.f32(0.1)))
    return reward


def feet_slide(net_forces_w_history, body_lin_vel_w, sensor_body_ids, asset_body_ids):
    """V:69-83."""
    contacts = _hist_max_norm(net_forces_w_history, sensor_body_ids) > 1.0
    v = body_lin_vel_w[:, asset_body_ids, :2]
    return torch.sum(M.norm2(v[..., 0], v[..., 1]) * contacts, dim=1)


def track_lin_vel_xy_yaw_frame_exp(command, root_quat_w, root_lin_vel_w, std: float):
    """V:86-96."""
    vel_yaw = M.quat_rotate_inverse(M.yaw_quat(root_quat_w), root_lin_vel_w[:, :3])
    err = torch.sum(torch.square(command[:, :2] - vel_yaw[:, :2]), dim=1)
    return torch.exp(-err / M.f32(std**2))


def track_ang_vel_z_world_exp(command, root_ang_vel_w, std: float):
    """V:99-106."""
    err = torch.square(command[:, 2] - root_ang_vel_w[:, 2])
    return torch.exp(-err / M.f32(std**2))


# ---- rewards: reach / cartpole ---------------------------------------------------------------------
def _des_pos_w(command, root_state_w):
    des_pos_w, _ = M.combine_frame_transforms(root_state_w[:, :3], root_state_w[:, 3:7], command[:, :3])
    return des_pos_w


def position_command_error(command, root_state_w, body_pos_w):
    """RE:19-33: |p_body - (root (+) p_cmd)|."""
    d = body_pos_w - _des_pos_w(command, root_state_w)
    return M.norm3(d[:, 0], d[:, 1], d[:, 2])


def position_command_error_tanh(command, root_state_w, body_pos_w, std: float):
    """RE:36-52."""
    return 1 - torch.tanh(position_command_error(command, root_state_w, body_pos_w) / M.f32(std))


def orientation_command_error(command, root_state_w, body_quat_w):
    """RE:55-69."""
    des_quat_w = M.quat_mul(root_state_w[:, 3:7], command[:, 3:7])
    return M.quat_error_magnitude(body_quat_w, des_quat_w)


def joint_pos_target_l2(joint_pos, target: float, joint_ids=ALL):
    """CP:19-26."""
    q = M.wrap_to_pi(joint_pos[:, joint_ids])
    return torch.sum(torch.square(q - M.f32(target)), dim=1)


# ---- terminations --------------------------------------------------------------------------------
def time_out(episode_length_buf, max_episode_length: int):
    """T:30-32."""
    return episode_length_buf >= max_episode_length


def command_resample(time_left, command_counter, step_dt: float, num_resamples: int = 1):
    """T:35-42."""
    return torch.logical_and(time_left <= float(M.f32(step_dt)), command_counter == num_resamples)


def bad_orientation(projected_gravity_b, limit_angle: float):
    """T:50-59 (goes through acos: not part of a bit-exact claim)."""
    return torch.acos(-projected_gravity_b[:, 2]).abs() > float(M.f32(limit_angle))


def root_height_below_minimum(root_pos_w, minimum_height: float):
    """T:62-72."""
    return root_pos_w[:, 2] < float(M.f32(minimum_height))


def joint_pos_out_of_manual_limit(joint_pos, bounds, joint_ids=ALL):
    """T:90-105."""
    q = joint_pos[:, joint_ids]
    return torch.logical_or(torch.any(q > float(M.f32(bounds[1])), dim=1), torch.any(q < float(M.f32(bounds[0])), dim=1))


def joint_vel_out_of_limit(joint_vel, soft_joint_vel_limits, joint_ids=ALL):
    """T:108-114."""
    return torch.any(torch.abs(joint_vel[:, joint_ids]) > soft_joint_vel_limits[:, joint_ids], dim=1)


def joint_vel_out_of_manual_limit(joint_vel, max_velocity: float, joint_ids=ALL):
    """T:117-124."""
    return torch.any(torch.abs(joint_vel[:, joint_ids]) > float(M.f32(max_velocity)), dim=1)


def joint_effort_out_of_limit(computed_torque, applied_torque, joint_ids=ALL):
    """T:127-142.  As written in the reference this is True when some joint was NOT clipped
    (``any(isclose(computed, applied))``), the opposite of its docstring; restated as is."""
    return torch.any(torch.isclose(computed_torque[:, joint_ids], applied_torque[:, joint_ids]), dim=1)


def illegal_contact(net_forces_w_history, threshold: float, body_ids):
    """T:150-158."""
    return torch.any(_hist_max_norm(net_forces_w_history, body_ids) > float(M.f32(threshold)), dim=1)


# ---- observations --------------------------------------------------------------------------------
def joint_pos_rel(joint_pos, default_joint_pos, joint_ids=ALL):
    """O:114-121."""
    return joint_pos[:, joint_ids] - default_joint_pos[:, joint_ids]


def joint_vel_rel(joint_vel, default_joint_vel, joint_ids=ALL):
    """O:150-157."""
    return joint_vel[:, joint_ids] - default_joint_vel[:, joint_ids]


def joint_pos_limit_normalized(joint_pos, soft_joint_pos_limits, joint_ids=ALL):
    """O:124-137."""
    return M.scale_transform(joint_pos[:, joint_ids], soft_joint_pos_limits[:, joint_ids, 0],
                             soft_joint_pos_limits[:, joint_ids, 1])


def height_scan(sensor_pos_w, ray_hits_w, offset: float = 0.5):
    """O:165-173."""
    return sensor_pos_w[:, 2].unsqueeze(1) - ray_hits_w[..., 2] - M.f32(offset)


def root_pos_rel(root_pos_w, env_origins):
    """O:61-65."""
    return root_pos_w - env_origins


def apply_obs_pipeline(obs, noise_u=None, noise_range=None, clip=None, scale=None):
    """observation_manager.py:308-318: clone -> uniform noise (noise_model.py:63-64) -> clip -> scale.

    ``noise_u`` are the U[0,1) draws ``torch.rand_like`` would have produced (injected for parity)."""
    obs = obs.clone()
    if noise_range is not None:
        n_min, n_max = noise_range
        obs = obs + noise_u * M.f32(n_max - n_min) + M.f32(n_min)
    if clip is not None:
        obs = obs.clip(min=float(M.f32(clip[0])), max=float(M.f32(clip[1])))
    if scale is not None:
        obs = obs * M.f32(scale)
    return obs
