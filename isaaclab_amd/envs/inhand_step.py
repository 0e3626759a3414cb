"""Non-physics half of ``DirectRLEnv.step`` for the in-hand manipulation env (Shadow / Allegro hand) on HIP kernels
(``isaaclab_tasks/direct/inhand_manipulation/inhand_manipulation_env.py``; step order ``direct_rl_env.py:330-386``).

Buffers carry the reference env's attribute names (``cur_targets``, ``goal_rot``, ``successes``, ``reset_goal_buf``,
``consecutive_successes`` ...).  The simulator's part -- hand / object state of the step -- is handed in with
:meth:`set_sim_state`.  Per env step: ``decimation`` x ``ilx_inhand_apply_action``, one ``ilx_inhand_step``, then
``ilx_inhand_finish`` (after the 2-float all-reduce when several GPUs share the run).
"""

from __future__ import annotations

import ctypes
from dataclasses import dataclass

import torch

from .. import _abi
from ..distributed import all_reduce_log_sums


def _world_size() -> int:
    import torch.distributed as dist

    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

c_f, c_i32, c_i64, c_u64, P = ctypes.c_float, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64, ctypes.c_void_p


class InHandDesc(ctypes.Structure):
    """Mirror of ``ilx_inhand`` (include/isaaclab_amd_inhand.h)."""

    _fields_ = [
        ("num_envs", c_i64), ("num_joints", c_i32), ("num_bodies", c_i32), ("num_actuated", c_i32), ("num_fingertips", c_i32),
        ("actuated_ids", P), ("fingertip_ids", P),
        ("joint_pos", P), ("joint_vel", P), ("lower_limits", P), ("upper_limits", P), ("body_state_w", P), ("object_state_w", P),
        ("env_origins", P), ("in_hand_pos", P), ("actions", P),
        ("cur_targets", P), ("prev_targets", P), ("goal_rot", P), ("successes", P), ("reset_goal_buf", P),
        ("episode_length_buf", P), ("consecutive_successes", P), ("stats", P),
        ("obs", P), ("reward", P), ("terminated", P), ("time_outs", P), ("reset_buf", P),
        ("dist_reward_scale", c_f), ("rot_reward_scale", c_f), ("rot_eps", c_f), ("action_penalty_scale", c_f),
        ("reach_goal_bonus", c_f), ("fall_penalty", c_f), ("fall_dist", c_f), ("vel_obs_scale", c_f), ("success_tolerance", c_f),
        ("av_factor", c_f), ("one_minus_av_factor", c_f), ("act_moving_average", c_f), ("one_minus_act_moving_average", c_f),
        ("max_consecutive_success", c_i32), ("max_episode_length", c_i64),
        ("goal_samples", P), ("reset_goal_samples", P), ("seed", c_u64), ("step_index", c_u64), ("step_index_device", P),
        ("block_sums", P),
    ]


_apply_action = _abi._sig("ilx_inhand_apply_action", ctypes.POINTER(InHandDesc), P)
_step = _abi._sig("ilx_inhand_step", ctypes.POINTER(InHandDesc), P)
_finish = _abi._sig("ilx_inhand_finish", ctypes.POINTER(InHandDesc), P)


@dataclass
class InHandManipulationCfg:
    """The fields of ``ShadowHandEnvCfg`` the non-physics step reads (shadow_hand_env_cfg.py:117-227)."""

    decimation: int = 2
    episode_length_s: float = 10.0
    sim_dt: float = 1 / 120
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


class InHandManipulationStep:
    def __init__(self, cfg: InHandManipulationCfg, num_envs: int, device, *, num_joints: int, num_bodies: int,
                 actuated_dof_indices, finger_bodies, lower_limits: torch.Tensor, upper_limits: torch.Tensor,
                 in_hand_pos: torch.Tensor, env_origins: torch.Tensor, seed: int = 0):
        import math

        self.cfg, self.num_envs, self.device = cfg, num_envs, torch.device(device)
        n, dev = num_envs, self.device
        self.actuated_dof_indices = sorted(actuated_dof_indices)
        self.finger_bodies = sorted(finger_bodies)
        self.num_hand_dofs, self.num_fingertips = num_joints, len(self.finger_bodies)
        a = len(self.actuated_dof_indices)
        # direct_rl_env.py: max_episode_length = ceil(episode_length_s / (sim.dt * decimation))
        self.max_episode_length = math.ceil(cfg.episode_length_s / (cfg.sim_dt * cfg.decimation))
        self._act_ids = torch.tensor(self.actuated_dof_indices, dtype=torch.int32, device=dev)
        self._tip_ids = torch.tensor(self.finger_bodies, dtype=torch.int32, device=dev)
        self.hand_dof_lower_limits = _abi.f32c(lower_limits.to(dev))
        self.hand_dof_upper_limits = _abi.f32c(upper_limits.to(dev))
        self.in_hand_pos, self.env_origins = _abi.f32c(in_hand_pos.to(dev)), _abi.f32c(env_origins.to(dev))
        z = lambda *s, **k: torch.zeros(*s, device=dev, **k)  # noqa: E731
        self.prev_targets, self.cur_targets = z(n, num_joints), z(n, num_joints)
        self.goal_rot = z(n, 4)
        self.goal_rot[:, 0] = 1.0
        self.successes, self.consecutive_successes = z(n), z(1)
        self.reset_goal_buf = z(n, dtype=torch.bool)
        self.episode_length_buf = z(n, dtype=torch.long)
        self.actions = z(n, a)
        self.obs_dim = 2 * num_joints + 24 + 13 * self.num_fingertips + a
        self.obs, self.reward = z(n, self.obs_dim), z(n)
        self.reset_terminated, self.reset_time_outs, self.reset_buf = (z(n, dtype=torch.bool) for _ in range(3))
        self.stats = z(2)
        self._block_sums = z(2 * ((n + 15) // 16))   # per-workgroup sums of the step launch (one GPU: see _fill)
        self.sums_per_workgroup = True
        self.joint_pos = self.joint_vel = self.body_state_w = self.object_state_w = None
        self._step_counter = None   # see count_steps_on_device
        d = self.desc = InHandDesc(num_envs=n, num_joints=num_joints, num_bodies=num_bodies, num_actuated=a,
                                   num_fingertips=self.num_fingertips, seed=seed, step_index=0)
        for k in ("dist_reward_scale", "rot_reward_scale", "rot_eps", "action_penalty_scale", "reach_goal_bonus", "fall_penalty",
                  "fall_dist", "vel_obs_scale", "success_tolerance", "av_factor", "act_moving_average"):
            setattr(d, k, float(getattr(cfg, k)))
        d.one_minus_av_factor = 1.0 - cfg.av_factor                       # host double, rounded once like the reference's scalar
        d.one_minus_act_moving_average = 1.0 - cfg.act_moving_average
        d.max_consecutive_success, d.max_episode_length = cfg.max_consecutive_success, self.max_episode_length

    def set_sim_state(self, joint_pos, joint_vel, body_state_w, object_state_w):
        """Hand joint state (N,J), hand body states (N,B,13) and the object's root state (N,13) of this step."""
        self.joint_pos, self.joint_vel = _abi.f32c(joint_pos), _abi.f32c(joint_vel)
        self.body_state_w, self.object_state_w = _abi.f32c(body_state_w), _abi.f32c(object_state_w)

    def _fill(self, goal_samples=None, reset_goal_samples=None):
        d, p = self.desc, _abi.ptr
        d.actuated_ids, d.fingertip_ids = p(self._act_ids), p(self._tip_ids)
        d.joint_pos, d.joint_vel = p(self.joint_pos), p(self.joint_vel)
        d.lower_limits, d.upper_limits = p(self.hand_dof_lower_limits), p(self.hand_dof_upper_limits)
        d.body_state_w, d.object_state_w = p(self.body_state_w), p(self.object_state_w)
        d.env_origins, d.in_hand_pos, d.actions = p(self.env_origins), p(self.in_hand_pos), p(self.actions)
        d.cur_targets, d.prev_targets, d.goal_rot, d.successes = p(self.cur_targets), p(self.prev_targets), p(self.goal_rot), p(self.successes)
        d.reset_goal_buf, d.episode_length_buf = self.reset_goal_buf.data_ptr(), p(self.episode_length_buf)
        d.consecutive_successes, d.stats = p(self.consecutive_successes), p(self.stats)
        d.obs, d.reward = p(self.obs), p(self.reward)
        d.terminated, d.time_outs, d.reset_buf = (t.data_ptr() for t in (self.reset_terminated, self.reset_time_outs, self.reset_buf))
        d.goal_samples, d.reset_goal_samples = p(goal_samples), p(reset_goal_samples)
        # on several GPUs `stats` is all-reduced between the step and the finish launch: the workgroups add to it themselves
        d.block_sums = p(self._block_sums) if self.sums_per_workgroup and _world_size() == 1 else None

    def step(self, actions: torch.Tensor, goal_samples=None, reset_goal_samples=None):
        """direct_rl_env.py:330-386 without the simulator: returns (obs, reward, terminated, time_outs)."""
        self.actions = _abi.f32c(actions)                      # _pre_physics_step (:94-95)
        self._fill(goal_samples, reset_goal_samples)
        ref, s = ctypes.byref(self.desc), _abi.stream()
        for _ in range(self.cfg.decimation):
            _abi.check(_apply_action(ref, s), "ilx_inhand_apply_action")
        _abi.check(_step(ref, s), "ilx_inhand_step")
        all_reduce_log_sums(self.stats)                        # no-op on one GPU
        _abi.check(_finish(ref, s), "ilx_inhand_finish")
        if self._step_counter is None:
            self.desc.step_index += 1
        return self.obs, self.reward, self.reset_terminated, self.reset_time_outs

    def count_steps_on_device(self):
        """Keep the generator's step count in device memory (``ilx_inhand_finish`` advances it): a step recorded in a HIP
        graph then draws fresh goal rotations at every replay; the numbers are those of the host-side count."""
        if self._step_counter is None:
            self._step_counter = torch.full((1,), int(self.desc.step_index), dtype=torch.int64, device=self.device)
            self.desc.step_index = 0
            self.desc.step_index_device = self._step_counter.data_ptr()
