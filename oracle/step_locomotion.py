"""Oracle for one non-physics env step of the rough-terrain velocity tasks (test infrastructure, CPU only).

Plain PyTorch CPU ops in the reference's per-term, unfused order
(``source/isaaclab/isaaclab/envs/manager_based_rl_env.py:152-241`` without ``sim.step``), composed from the
oracle modules.  It is what the GPU tests compare ``LocomotionVelocityStep`` against and what ``bench.py`` times as
the CPU baseline.  Constants: ``velocity_env_cfg.py``, ``config/anymal_c/rough_env_cfg.py`` and
``config/h1/rough_env_cfg.py`` of the reference.

``AnymalRoughOracle``: Isaac-Velocity-Rough-Anymal-C-v0 with the ActuatorNetMLP actuator.
``H1RoughOracle``: Isaac-Velocity-Rough-H1-v0 (``config/h1/rough_env_cfg.py:19-115``) with DC-motor actuators on the
H1 gains and limits (``isaaclab_assets/robots/unitree.py:216-260``), the height scanner on ``torso_link``.
"""

from __future__ import annotations

import math
import re

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

# H1: joint / body order as the asset lists them (breadth first), constants of unitree.py:199-260
H1_JOINTS = ["left_hip_yaw", "right_hip_yaw", "torso", "left_hip_roll", "right_hip_roll", "left_shoulder_pitch",
             "right_shoulder_pitch", "left_hip_pitch", "right_hip_pitch", "left_shoulder_roll", "right_shoulder_roll",
             "left_knee", "right_knee", "left_shoulder_yaw", "right_shoulder_yaw", "left_ankle", "right_ankle",
             "left_elbow", "right_elbow"]
H1_BODIES = ["pelvis"] + [j + "_link" for j in H1_JOINTS]
H1_DEFAULT = {".*_hip_yaw": 0.0, ".*_hip_roll": 0.0, ".*_hip_pitch": -0.28, ".*_knee": 0.79, ".*_ankle": -0.52, "torso": 0.0,
              ".*_shoulder_pitch": 0.28, ".*_shoulder_roll": 0.0, ".*_shoulder_yaw": 0.0, ".*_elbow": 0.52}
H1_KP = {".*_hip_yaw": 150.0, ".*_hip_roll": 150.0, ".*_hip_pitch": 200.0, ".*_knee": 200.0, "torso": 200.0, ".*_ankle": 20.0,
         ".*_shoulder_pitch": 40.0, ".*_shoulder_roll": 40.0, ".*_shoulder_yaw": 40.0, ".*_elbow": 40.0}
H1_KD = {".*_hip_yaw": 5.0, ".*_hip_roll": 5.0, ".*_hip_pitch": 5.0, ".*_knee": 5.0, "torso": 5.0, ".*_ankle": 4.0,
         ".*_shoulder_pitch": 10.0, ".*_shoulder_roll": 10.0, ".*_shoulder_yaw": 10.0, ".*_elbow": 10.0}
H1_EFFORT = {".*_ankle": 100.0, ".*": 300.0}   # first match wins below
# rewards of H1Rewards + the base task, in manager order (velocity_env_cfg.py:226-254 with the H1 overrides applied, then
# the terms H1Rewards adds): name, weight
H1_WEIGHTS = [1.0, 1.0, -0.05, 0.0, -1.25e-7, -0.005, 0.25, -1.0, -1.0, -200.0, -0.25, -0.2, -0.2, -0.1]


def _match(names, patterns):
    pats = [patterns] if isinstance(patterns, str) else patterns
    return [i for i, n in enumerate(names) if any(re.fullmatch(p, n) for p in pats)]


def _per_joint(table: dict, names):
    out = []
    for n in names:
        out.append(next(v for p, v in table.items() if re.fullmatch(p, n)))
    return torch.tensor(out, dtype=torch.float32)


class _RoughVelocityOracle:
    """What both robots share: substep loop, scanner, terminations, reward / reset / command / observation order."""

    num_rewards = 0

    def __init__(self, n, state: dict, default_joint_pos, cmesh: RC.CMesh, cmd_cfg=None):
        self.n = n
        s = {k: v.clone() for k, v in state.items()}
        self.s = s
        self.default_joint_pos = default_joint_pos.reshape(1, -1).expand(n, -1).contiguous()
        self.cmesh = cmesh
        self.cmd_cfg = cmd_cfg or C.VelocityCommandCfg()
        self.cmd = C.VelocityCommandState(s["command"], s["heading_target"], s["is_heading_env"], s["is_standing_env"],
                                          s["time_left"], s["command_counter"], torch.zeros(n), torch.zeros(n))
        self.episode_sums = torch.zeros(self.num_rewards, n)
        self.step_reward = torch.zeros(n, self.num_rewards)
        self.ts, self.ts_last = torch.zeros(n), torch.zeros(n)
        self.outdated = torch.zeros(n, dtype=torch.bool)
        self.net_w = torch.zeros_like(s["net_forces"][0])
        self.ray_starts, self.ray_dirs = RC.sensor_rays(*RC.grid_pattern((1.6, 1.0), 0.1), (0.0, 0.0, 20.0), (1.0, 0.0, 0.0, 0.0))
        self.sim_time = 0.0
        self.acc_time = -1.0

    # -- per robot
    def actuator(self, q_des, q, qd):
        raise NotImplementedError

    def scanner_pose(self):
        rs = self.s["root_state_w"]
        return rs[:, :3], rs[:, 3:7]

    def terminations(self, hist):
        raise NotImplementedError

    def rewards(self, **k):
        raise NotImplementedError

    def world_rays(self):
        """World ray starts / directions of the height scanner (ray_caster.py:220-252)."""
        pos, quat = self.scanner_pose()
        return RC.world_rays(pos, quat, self.ray_starts, self.ray_dirs, True)

    def step(self, raw_action, noise_u, command_samples, reset_command_samples, ray_xy=None):
        """``ray_xy`` (n, R, 2), optional: x / y of the world ray starts to cast instead of the oracle's own -- the
        yaw rotation of the starts goes through atan2 / sin / cos, whose last bit differs between libraries, and a ray
        that close to a facet edge lands on the other facet.  Entries that are not finite keep the oracle's value."""
        s, n = self.s, self.n
        # process_action (action_manager.py:331-332, joint_actions.py:134, :156)
        s["prev_action"] = s["action"]
        s["action"] = raw_action.clone()
        q_des = raw_action * 0.5 + self.default_joint_pos
        for k in range(DECIMATION):
            q, qd = s["joint_pos"][k], s["joint_vel"][k]
            comp, app = self.actuator(q_des, q, qd)
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
        sensor_pos, _ = self.scanner_pose()
        sw, dw = self.world_rays()
        if ray_xy is not None:
            sw = sw.clone()
            sw[..., :2] = torch.where(torch.isfinite(ray_xy), ray_xy, sw[..., :2])
        hits, _, _ = RC.raycast_c(sw, dw, self.cmesh)
        ray_hits = hits.view(n, -1, 3)
        # episode length, terminations, rewards (manager_based_rl_env.py:199-206)
        s["episode_length_buf"] = s["episode_length_buf"] + 1
        hist = s["net_forces_w_history"]
        dones, terminated, truncated, term_dones = self.terminations(hist)
        vals, weights = self.rewards(lin_b=lin_b, ang_b=ang_b, grav_b=grav_b, q=q, qd=qd, app=app, acc=acc, hist=hist,
                                     terminated=terminated)
        reward = MG.reward_compute(vals, weights, STEP_DT, self.episode_sums, self.step_reward)
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
                 (T.height_scan(sensor_pos, ray_hits, 0.5), (-0.1, 0.1), (-1.0, 1.0))]
        cols, col = [], 0
        for val, noise, clip in parts:
            w = val.shape[1]
            cols.append(T.apply_obs_pipeline(val, noise_u[:, col:col + w], noise, clip))
            col += w
        obs = MG.observation_concat(cols)
        return dict(obs=obs, reward=reward, terminated=terminated, time_outs=truncated, dones=dones, term_dones=term_dones,
                    computed_torque=comp, applied_torque=app, joint_acc=acc, ray_hits_w=ray_hits, ray_starts_w=sw,
                    reward_log=reward_log, term_log=term_log, cmd_log=cmd_log, step_reward=self.step_reward.clone(),
                    episode_sums=self.episode_sums.clone())


class AnymalRoughOracle(_RoughVelocityOracle):
    num_rewards = len(WEIGHTS)

    def __init__(self, n, state: dict, net_weights, net_biases, default_joint_pos, cmesh: RC.CMesh):
        super().__init__(n, state, default_joint_pos, cmesh)
        self.w, self.b = net_weights, net_biases
        self.lim = torch.full((n, 12), 80.0)
        self.vlim = torch.full((n, 12), 7.5)

    def actuator(self, q_des, q, qd):
        s = self.s
        comp, app, s["pos_error_history"], s["vel_history"] = A.actuator_net_mlp(
            q_des, q, qd, s["pos_error_history"], s["vel_history"], self.w, self.b, "softsign", [0, 1, 2], "pos_vel",
            -1.0, 1.0, 1.0, self.lim, self.vlim, 120.0)
        return comp, app

    def terminations(self, hist):
        return MG.termination_compute([T.time_out(self.s["episode_length_buf"], MAX_LEN), T.illegal_contact(hist, 1.0, BASE)],
                                      [True, False])

    def rewards(self, *, lin_b, ang_b, app, acc, hist, **_):
        s, cmd = self.s, self.cmd.vel_command_b
        vals = [T.track_lin_vel_xy_exp(cmd, lin_b, SQ), T.track_ang_vel_z_exp(cmd, ang_b, SQ), T.lin_vel_z_l2(lin_b),
                T.ang_vel_xy_l2(ang_b), T.joint_torques_l2(app), T.joint_acc_l2(acc), T.action_rate_l2(s["action"], s["prev_action"]),
                T.feet_air_time(s["current_contact_time"], s["last_air_time"], cmd, STEP_DT, 0.5, FEET),
                T.undesired_contacts(hist, 1.0, THIGH), None, None]
        return vals, WEIGHTS


class H1RoughOracle(_RoughVelocityOracle):
    """``state`` additionally holds ``body_state_w`` (n, 20, 13): the height scanner hangs on ``torso_link`` and
    ``feet_slide`` reads the ankle links' velocities."""

    num_rewards = len(H1_WEIGHTS)

    def __init__(self, n, state: dict, cmesh: RC.CMesh):
        default = torch.zeros(19)
        for i, name in enumerate(H1_JOINTS):
            default[i] = next(v for p, v in H1_DEFAULT.items() if re.fullmatch(p, name))
        cfg = C.VelocityCommandCfg(lin_vel_x=(0.0, 1.0), lin_vel_y=(0.0, 0.0), ang_vel_z=(-1.0, 1.0))  # h1/rough_env_cfg.py:108-110
        super().__init__(n, state, default, cmesh, cfg)
        self.kp, self.kd = _per_joint(H1_KP, H1_JOINTS).expand(n, -1), _per_joint(H1_KD, H1_JOINTS).expand(n, -1)
        self.effort = _per_joint(H1_EFFORT, H1_JOINTS).expand(n, -1)
        self.vlim = torch.full((n, 19), 100.0)
        self.feet = _match(H1_BODIES, ".*ankle_link")
        self.torso = _match(H1_BODIES, ".*torso_link")
        self.jd_hip = _match(H1_JOINTS, [".*_hip_yaw", ".*_hip_roll"])
        self.jd_arms = _match(H1_JOINTS, [".*_shoulder_.*", ".*_elbow"])
        self.jd_torso = _match(H1_JOINTS, "torso")
        self.ankles = _match(H1_JOINTS, ".*_ankle")
        # articulation.py:1230-1237 on hard limits of default +- 1.5 rad (the asset's limits live in its USD file)
        hard = torch.stack([self.default_joint_pos - 1.5, self.default_joint_pos + 1.5], dim=-1)
        self.soft_limits = AD.soft_joint_pos_limits(hard, 0.9)   # unitree.py:213

    def actuator(self, q_des, q, qd):
        zero = torch.zeros_like(q)
        return A.dc_motor(q_des, zero, zero, q, qd, self.kp, self.kd, self.effort, self.vlim, 300.0)

    def scanner_pose(self):
        b = self.s["body_state_w"][:, self.torso[0]]
        return b[:, :3], b[:, 3:7]

    def terminations(self, hist):
        return MG.termination_compute([T.time_out(self.s["episode_length_buf"], MAX_LEN), T.illegal_contact(hist, 1.0, self.torso)],
                                      [True, False])

    def rewards(self, *, ang_b, grav_b, q, acc, hist, terminated, **_):
        s, cmd, rs = self.s, self.cmd.vel_command_b, self.s["root_state_w"]
        d = self.default_joint_pos
        vals = [T.track_lin_vel_xy_yaw_frame_exp(cmd, rs[:, 3:7], rs[:, 7:10], 0.5), T.track_ang_vel_z_world_exp(cmd, rs[:, 10:13], 0.5),
                T.ang_vel_xy_l2(ang_b), None, T.joint_acc_l2(acc), T.action_rate_l2(s["action"], s["prev_action"]),
                T.feet_air_time_positive_biped(s["current_air_time"], s["current_contact_time"], cmd, 0.4, self.feet),
                T.flat_orientation_l2(grav_b), T.joint_pos_limits(q, self.soft_limits, self.ankles), T.is_terminated(terminated),
                T.feet_slide(hist, s["body_state_w"][:, :, 7:10], self.feet, self.feet),
                T.joint_deviation_l1(q, d, self.jd_hip), T.joint_deviation_l1(q, d, self.jd_arms),
                T.joint_deviation_l1(q, d, self.jd_torso)]
        return vals, H1_WEIGHTS
