"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the in-hand manipulation direct env's non-physics step
(``isaaclab_tasks/direct/inhand_manipulation/inhand_manipulation_env.py``; cfg
``isaaclab_tasks/direct/shadow_hand/shadow_hand_env_cfg.py:117-227``).  Plain torch CPU ops in the reference's order.
Pinned by ``tests/golden/inhand.npz`` (outputs of the reference's own functions and methods).
Random draws are inputs (``*_samples`` in [-1, 1], what ``sample_uniform(-1, 1, ...)`` returned)."""

from __future__ import annotations

import math
from dataclasses import dataclass

import torch

from . import mathx as M


@dataclass
class InHandCfg:
    """shadow_hand_env_cfg.py:208-227."""
    dist_reward_scale: float = -10.0
    rot_reward_scale: float = 1.0
    rot_eps: float = 0.1
    action_penalty_scale: float = -0.0002
    reach_goal_bonus: float = 250.0
    fall_penalty: float = 0.0
    fall_dist: float = 0.24
    vel_obs_scale: float = 0.2
    success_tolerance: float = 0.1
    max_consecutive_success: int = 0
    av_factor: float = 0.1
    act_moving_average: float = 1.0
    max_episode_length: int = 600


def scale(x, lower, upper):  # :325-327
    return 0.5 * (x + 1.0) * (upper - lower) + lower


def unscale(x, lower, upper):  # :330-332
    return (2.0 * x - upper - lower) / (upper - lower)


def randomize_rotation(rand0, rand1):  # :335-339 with the x / y unit axes
    n = rand0.shape[0]
    x = torch.tensor([1.0, 0.0, 0.0]).repeat(n, 1)
    y = torch.tensor([0.0, 1.0, 0.0]).repeat(n, 1)
    return M.quat_mul(M.quat_from_angle_axis(rand0 * math.pi, x), M.quat_from_angle_axis(rand1 * math.pi, y))


def rotation_distance(object_rot, target_rot):  # :342-346
    d = M.quat_mul(object_rot, M.quat_conjugate(target_rot))
    return 2.0 * torch.asin(torch.clamp(M.norm3(d[:, 1], d[:, 2], d[:, 3]), max=1.0))


def apply_action(actions, cur_targets, prev_targets, lower, upper, actuated, moving_average: float):
    """:97-117 -> (cur_targets, prev_targets), full-width copies."""
    cur, prev = cur_targets.clone(), prev_targets.clone()
    lo, hi = lower[:, actuated], upper[:, actuated]
    t = scale(actions, lo, hi)
    t = moving_average * t + (1.0 - moving_average) * prev[:, actuated]
    t = torch.clamp(t, lo, hi)  # math.py:61-78 (saturate)
    cur[:, actuated] = t
    prev[:, actuated] = t
    return cur, prev


def get_dones(object_pos, in_hand_pos, object_rot, goal_rot, successes, episode_length_buf, cfg: InHandCfg):
    """:179-199 -> (out_of_reach, time_out, episode_length_buf)."""
    d = object_pos - in_hand_pos
    goal_dist = M.norm3(d[:, 0], d[:, 1], d[:, 2])
    out_of_reach = goal_dist >= cfg.fall_dist
    ep = episode_length_buf
    if cfg.max_consecutive_success > 0:
        rot_dist = rotation_distance(object_rot, goal_rot)
        ep = torch.where(torch.abs(rot_dist) <= cfg.success_tolerance, torch.zeros_like(ep), ep)
    time_out = ep >= cfg.max_episode_length - 1
    if cfg.max_consecutive_success > 0:
        time_out = time_out | (successes >= cfg.max_consecutive_success)
    return out_of_reach, time_out, ep


def reward_terms(reset_buf, reset_goal_buf, successes, object_pos, object_rot, target_pos, target_rot, actions, cfg: InHandCfg):
    """Per-env part of compute_rewards (:349-420) -> (reward, goal_resets, successes, resets)."""
    d = object_pos - target_pos
    goal_dist = M.norm3(d[:, 0], d[:, 1], d[:, 2])
    rot_dist = rotation_distance(object_rot, target_rot)
    dist_rew = goal_dist * cfg.dist_reward_scale
    rot_rew = 1.0 / (torch.abs(rot_dist) + cfg.rot_eps) * cfg.rot_reward_scale
    action_penalty = torch.sum(actions**2, dim=-1)
    reward = dist_rew + rot_rew + action_penalty * cfg.action_penalty_scale
    goal_resets = torch.where(torch.abs(rot_dist) <= cfg.success_tolerance, torch.ones_like(reset_goal_buf), reset_goal_buf)
    successes = successes + goal_resets
    reward = torch.where(goal_resets == 1, reward + cfg.reach_goal_bonus, reward)
    reward = torch.where(goal_dist >= cfg.fall_dist, reward + cfg.fall_penalty, reward)
    resets = torch.where(goal_dist >= cfg.fall_dist, torch.ones_like(reset_buf), reset_buf)
    return reward, goal_resets, successes, resets


def success_stats(successes, resets):
    """The two global sums of compute_rewards (:422-424): (num_resets, finished_cons_successes)."""
    return torch.stack([torch.sum(resets).float(), torch.sum(successes * resets.float())])


def consecutive_successes(stats, previous, av_factor: float):
    """:426-431 from the (possibly all-reduced) sums."""
    return torch.where(stats[0] > 0, av_factor * stats[1] / stats[0] + (1.0 - av_factor) * previous, previous)


def full_observations(hand_dof_pos, hand_dof_vel, lower, upper, object_pos, object_rot, object_linvel, object_angvel,
                      in_hand_pos, goal_rot, fingertip_pos, fingertip_rot, fingertip_vel, actions, vel_obs_scale: float):
    """:298-322."""
    n = hand_dof_pos.shape[0]
    return torch.cat((unscale(hand_dof_pos, lower, upper), vel_obs_scale * hand_dof_vel, object_pos, object_rot, object_linvel,
                      vel_obs_scale * object_angvel, in_hand_pos, goal_rot, M.quat_mul(object_rot, M.quat_conjugate(goal_rot)),
                      fingertip_pos.reshape(n, -1), fingertip_rot.reshape(n, -1), fingertip_vel.reshape(n, -1), actions), dim=-1)


class InHandOracle:
    """One env step of the reference minus the simulator: DirectRLEnv.step order (direct_rl_env.py:330-386) --
    apply_action x decimation, episode_length += 1, dones, rewards, goal resets, env resets (the bookkeeping that
    does not write into the simulator), observations."""

    def __init__(self, cfg: InHandCfg, state: dict, actuated, finger_bodies, decimation: int = 2):
        self.cfg, self.actuated, self.fingers, self.decimation = cfg, list(actuated), list(finger_bodies), decimation
        self.s = {k: v.clone() for k, v in state.items() if torch.is_tensor(v)}

    def step(self, actions, goal_samples, reset_goal_samples, reduce_stats=None):
        """``reduce_stats`` (optional): applied to the two success sums before they enter ``consecutive_successes`` -- the
        all-reduce over the env shards of a multi-GPU run; the sums it returned are kept in ``self.last_stats``."""
        s, c = self.s, self.cfg
        for _ in range(self.decimation):
            s["cur_targets"], s["prev_targets"] = apply_action(actions, s["cur_targets"], s["prev_targets"], s["lower"], s["upper"],
                                                               self.actuated, c.act_moving_average)
        s["episode_length_buf"] = s["episode_length_buf"] + 1
        object_pos = s["object_state"][:, 0:3] - s["env_origins"]
        object_rot = s["object_state"][:, 3:7]
        out_of_reach, time_out, s["episode_length_buf"] = get_dones(object_pos, s["in_hand_pos"], object_rot, s["goal_rot"],
                                                                    s["successes"], s["episode_length_buf"], c)
        reset_buf = out_of_reach | time_out
        reward, goal_resets, s["successes"], resets = reward_terms(reset_buf, s["reset_goal_buf"], s["successes"], object_pos,
                                                                   object_rot, s["in_hand_pos"], s["goal_rot"], actions, c)
        stats = success_stats(s["successes"], resets)
        if reduce_stats is not None:
            stats = reduce_stats(stats)
        self.last_stats = stats
        s["consecutive_successes"] = consecutive_successes(stats, s["consecutive_successes"], c.av_factor)
        s["reset_goal_buf"] = goal_resets
        # :168-171 goal resets, then direct_rl_env.py:369-371 + :200-247 env resets
        new_goal = randomize_rotation(goal_samples[:, 0], goal_samples[:, 1])
        s["goal_rot"] = torch.where(goal_resets.bool().unsqueeze(-1), new_goal, s["goal_rot"])
        s["reset_goal_buf"] = torch.where(goal_resets.bool(), torch.zeros_like(goal_resets), s["reset_goal_buf"])
        new_goal = randomize_rotation(reset_goal_samples[:, 0], reset_goal_samples[:, 1])
        s["goal_rot"] = torch.where(reset_buf.unsqueeze(-1), new_goal, s["goal_rot"])
        s["reset_goal_buf"] = torch.where(reset_buf, torch.zeros_like(goal_resets), s["reset_goal_buf"])
        s["successes"] = torch.where(reset_buf, torch.zeros_like(s["successes"]), s["successes"])
        s["episode_length_buf"] = torch.where(reset_buf, torch.zeros_like(s["episode_length_buf"]), s["episode_length_buf"])
        body = s["body_state"][:, self.fingers]
        obs = full_observations(s["joint_pos"], s["joint_vel"], s["lower"], s["upper"], object_pos, object_rot,
                                s["object_state"][:, 7:10], s["object_state"][:, 10:13], s["in_hand_pos"], s["goal_rot"],
                                body[..., 0:3] - s["env_origins"].unsqueeze(1), body[..., 3:7], body[..., 7:13], actions,
                                c.vel_obs_scale)
        return dict(obs=obs, reward=reward, terminated=out_of_reach, time_outs=time_out, stats=stats, goal_resets=goal_resets)
