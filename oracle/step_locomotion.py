"""Oracle for one non-physics env step of the rough-terrain velocity tasks (test infrastructure, CPU only).

Plain PyTorch CPU ops in the reference's per-term, unfused order
(``source/isaaclab/isaaclab/envs/manager_based_rl_env.py:152-241`` without ``sim.step``), composed from the
oracle modules.  It is what the GPU tests compare ``LocomotionVelocityStep`` against and what ``bench.py`` times as
the CPU baseline.  Constants: ``velocity_env_cfg.py`` / ``config/anymal_c/rough_env_cfg.py`` of the reference.
"""

from __future__ import annotations

import math

import torch

from . import actuators as A
from . import articulation as AD
from . import command as C
from . import contact as CT
from . import managers as MG
from . import raycast as RC
from . import terms as T

STEP_DT, PHYSICS_DT, DECIMATION, MAX_LEN = 0.02, 0.005, 4, 1000
FEET, THIGH, BASE = [13, 14, 15, 16], [5, 6, 7, 8], [0]
WEIGHTS = [1.0, 0.5, -2.0, -0.05, -1.0e-5, -2.5e-7, -0.01, 0.125, -1.0, 0.0, 0.0]
SQ = math.sqrt(0.25)


class AnymalRoughOracle:
    def __init__(self, n, state: dict, net_weights, net_biases, default_joint_pos, cmesh: RC.CMesh):
        self.n = n
        s = {k: v.clone() for k, v in state.items()}
        self.s = s
        self.w, self.b = net_weights, net_biases
        self.default_joint_pos = default_joint_pos.reshape(1, -1).expand(n, -1).contiguous()
        self.cmesh = cmesh
        self.cmd_cfg = C.VelocityCommandCfg()
        self.cmd = C.VelocityCommandState(s["command"], s["heading_target"], s["is_heading_env"], s["is_standing_env"],
                                          s["time_left"], s["command_counter"], torch.zeros(n), torch.zeros(n))
        self.episode_sums = torch.zeros(len(WEIGHTS), n)
        self.step_reward = torch.zeros(n, len(WEIGHTS))
        self.ts, self.ts_last = torch.zeros(n), torch.zeros(n)
        self.outdated = torch.zeros(n, dtype=torch.bool)
        self.net_w = torch.zeros_like(s["net_forces"][0])
        self.lim = torch.full((n, 12), 80.0)
        self.vlim = torch.full((n, 12), 7.5)
        self.ray_starts, self.ray_dirs = RC.sensor_rays(*RC.grid_pattern((1.6, 1.0), 0.1), (0.0, 0.0, 20.0), (1.0, 0.0, 0.0, 0.0))
        self.sim_time = 0.0
        self.acc_time = -1.0

    def step(self, raw_action, noise_u, command_samples, reset_command_samples):
        s, n = self.s, self.n
        # process_action (action_manager.py:331-332, joint_actions.py:134, :156)
        s["prev_action"] = s["action"]
        s["action"] = raw_action.clone()
        q_des = raw_action * 0.5 + self.default_joint_pos
        for k in range(DECIMATION):
            q, qd = s["joint_pos"][k], s["joint_vel"][k]
            comp, app, s["pos_error_history"], s["vel_history"] = A.actuator_net_mlp(
                q_des, q, qd, s["pos_error_history"], s["vel_history"], self.w, self.b, "softsign", [0, 1, 2], "pos_vel",
                -1.0, 1.0, 1.0, self.lim, self.vlim, 120.0)
            self.sim_time += PHYSICS_DT
            acc, s["previous_joint_vel"] = AD.joint_acc(qd, s["previous_joint_vel"], self.sim_time - self.acc_time)
            self.acc_time = self.sim_time
            CT.contact_update(s["net_forces"][k], self.net_w, s["net_forces_w_history"], s["current_air_time"],
                              s["last_air_time"], s["current_contact_time"], s["last_contact_time"], self.ts, self.ts_last,
                              self.outdated, PHYSICS_DT, PHYSICS_DT, 1.0)
        rs = s["root_state_w"]
        lin_b, ang_b = AD.root_lin_vel_b(rs), AD.root_ang_vel_b(rs)
        grav_b = AD.projected_gravity_b(rs, torch.tensor([0.0, 0.0, -1.0]))
        heading = AD.heading_w(rs)
        # height scanner (ray_caster.py:220-260)
        pos, quat = rs[:, :3], rs[:, 3:7]
        sw, dw = RC.world_rays(pos, quat, self.ray_starts, self.ray_dirs, True)
        hits, _, _ = RC.raycast_c(sw, dw, self.cmesh)
        ray_hits = hits.view(n, -1, 3)
        # episode length, terminations, rewards (manager_based_rl_env.py:199-206)
        s["episode_length_buf"] = s["episode_length_buf"] + 1
        hist = s["net_forces_w_history"]
        dones, terminated, truncated, term_dones = MG.termination_compute(
            [T.time_out(s["episode_length_buf"], MAX_LEN), T.illegal_contact(hist, 1.0, BASE)], [True, False])
        cmd = self.cmd.vel_command_b
        vals = [T.track_lin_vel_xy_exp(cmd, lin_b, SQ), T.track_ang_vel_z_exp(cmd, ang_b, SQ), T.lin_vel_z_l2(lin_b),
                T.ang_vel_xy_l2(ang_b), T.joint_torques_l2(app), T.joint_acc_l2(acc), T.action_rate_l2(s["action"], s["prev_action"]),
                T.feet_air_time(s["current_contact_time"], s["last_air_time"], cmd, STEP_DT, 0.5, FEET),
                T.undesired_contacts(hist, 1.0, THIGH), None, None]
        reward = MG.reward_compute(vals, WEIGHTS, STEP_DT, self.episode_sums, self.step_reward)
        # manager resets for the done envs (manager_based_rl_env.py:364-391)
        reward_log = MG.reward_reset(self.episode_sums, dones, MAX_LEN * STEP_DT)
        term_log = MG.termination_reset_counts(term_dones, dones)
        cmd_log = C.velocity_reset(self.cmd_cfg, self.cmd, dones, reset_command_samples)
        # action_manager.py:309-311: the action history of the done envs restarts at zero
        s["action"] = torch.where(dones[:, None], torch.zeros_like(s["action"]), s["action"])
        s["prev_action"] = torch.where(dones[:, None], torch.zeros_like(s["prev_action"]), s["prev_action"])
        s["episode_length_buf"] = torch.where(dones, torch.zeros((), dtype=torch.long), s["episode_length_buf"])
        # command, observations (:232-238)
        C.velocity_compute(self.cmd_cfg, self.cmd, STEP_DT, STEP_DT, lin_b, ang_b, heading, command_samples)
        parts = [(lin_b, (-0.1, 0.1), None), (ang_b, (-0.2, 0.2), None), (grav_b, (-0.05, 0.05), None),
                 (self.cmd.vel_command_b, None, None), (T.joint_pos_rel(q, self.default_joint_pos), (-0.01, 0.01), None),
                 (T.joint_vel_rel(qd, torch.zeros_like(qd)), (-1.5, 1.5), None), (s["action"], None, None),
                 (T.height_scan(pos, ray_hits, 0.5), (-0.1, 0.1), (-1.0, 1.0))]
        cols, col = [], 0
        for val, noise, clip in parts:
            w = val.shape[1]
            cols.append(T.apply_obs_pipeline(val, noise_u[:, col:col + w], noise, clip))
            col += w
        obs = MG.observation_concat(cols)
        return dict(obs=obs, reward=reward, terminated=terminated, time_outs=truncated, dones=dones, term_dones=term_dones,
                    computed_torque=comp, applied_torque=app, joint_acc=acc, ray_hits_w=ray_hits, reward_log=reward_log,
                    term_log=term_log, cmd_log=cmd_log, step_reward=self.step_reward.clone(),
                    episode_sums=self.episode_sums.clone())
