"""Rough-terrain velocity-tracking tasks as fused term tables.

Constants follow ``source/isaaclab_tasks/isaaclab_tasks/manager_based/locomotion/velocity/velocity_env_cfg.py``
(base task), ``config/anymal_c/rough_env_cfg.py`` and ``config/h1/rough_env_cfg.py`` of the reference; robot
constants follow ``source/isaaclab_assets/isaaclab_assets/robots/anymal.py`` / ``unitree.py``.
"""

from __future__ import annotations

import math

from ..envs.fused_mdp import TermSpec, obs_term

STEP_DT = 0.02          # decimation 4 x sim dt 0.005 (velocity_env_cfg.py:299-302)
PHYSICS_DT = 0.005
DECIMATION = 4
EPISODE_LENGTH_S = 20.0
MAX_EPISODE_LENGTH = math.ceil(EPISODE_LENGTH_S / STEP_DT)  # manager_based_rl_env.py:100-102

# ---- ANYmal-C (anymal.py:114-120; PhysX orders joints breadth first) ---------------------------------
ANYMAL_JOINTS = ["LF_HAA", "LH_HAA", "RF_HAA", "RH_HAA", "LF_HFE", "LH_HFE", "RF_HFE", "RH_HFE", "LF_KFE", "LH_KFE",
                 "RF_KFE", "RH_KFE"]
ANYMAL_BODIES = ["base", "LF_HIP", "LH_HIP", "RF_HIP", "RH_HIP", "LF_THIGH", "LH_THIGH", "RF_THIGH", "RH_THIGH",
                 "LF_SHANK", "LH_SHANK", "RF_SHANK", "RH_SHANK", "LF_FOOT", "LH_FOOT", "RF_FOOT", "RH_FOOT"]
ANYMAL_DEFAULT_JOINT_POS = {".*HAA": 0.0, ".*F_HFE": 0.4, ".*H_HFE": -0.4, ".*F_KFE": -0.8, ".*H_KFE": 0.8}
ANYMAL_ACTUATOR = dict(saturation_effort=120.0, effort_limit=80.0, velocity_limit=7.5)  # anymal.py:45-51
# actuator-net scalars of the only ActuatorNetMLPCfg in the reference (unitree.py:30-41)
ACTUATOR_NET = dict(pos_scale=-1.0, vel_scale=1.0, torque_scale=1.0, input_order="pos_vel", input_idx=[0, 1, 2])

VELOCITY_COMMAND = dict(  # velocity_env_cfg.py:94-105
    kind="velocity", resampling_time_range=(10.0, 10.0), rel_standing_envs=0.02, rel_heading_envs=1.0,
    heading_command=True, heading_control_stiffness=0.5,
    ranges=[(-1.0, 1.0), (-1.0, 1.0), (-1.0, 1.0), (-math.pi, math.pi)])

HEIGHT_SCAN = dict(resolution=0.1, size=(1.6, 1.0), offset_z=20.0, update_period=STEP_DT)  # velocity_env_cfg.py:66-72,309
CONTACT_SENSOR = dict(history_length=3, track_air_time=True, update_period=PHYSICS_DT, force_threshold=1.0)  # :74, :311


def _ids(names, all_names):
    from ..utils.string import resolve_matching_names

    return resolve_matching_names(names, all_names)[0]


def anymal_c_rough_terms(bodies=ANYMAL_BODIES):
    """velocity_env_cfg.py:119-143 (policy observations), :226-254 (rewards), :261-265 (terminations)."""
    feet, thigh, base = _ids(".*FOOT", bodies), _ids(".*THIGH", bodies), _ids("base", bodies)
    std2 = math.sqrt(0.25) ** 2
    rewards = [
        TermSpec("track_lin_vel_xy_exp", 1.0, (std2, 0, 0, 0), name="track_lin_vel_xy_exp"),
        TermSpec("track_ang_vel_z_exp", 0.5, (std2, 0, 0, 0), name="track_ang_vel_z_exp"),
        TermSpec("lin_vel_z_l2", -2.0, name="lin_vel_z_l2"),
        TermSpec("ang_vel_xy_l2", -0.05, name="ang_vel_xy_l2"),
        TermSpec("joint_torques_l2", -1.0e-5, name="dof_torques_l2"),
        TermSpec("joint_acc_l2", -2.5e-7, name="dof_acc_l2"),
        TermSpec("action_rate_l2", -0.01, name="action_rate_l2"),
        TermSpec("feet_air_time", 0.125, (0.5, STEP_DT + 1.0e-8, 0, 0), ids=feet, name="feet_air_time"),
        TermSpec("undesired_contacts", -1.0, (1.0, 0, 0, 0), ids=thigh, name="undesired_contacts"),
        TermSpec("flat_orientation_l2", 0.0, name="flat_orientation_l2"),
        TermSpec("joint_pos_limits", 0.0, name="dof_pos_limits"),
    ]
    terminations = [
        TermSpec("time_out", flags=1, name="time_out"),
        TermSpec("illegal_contact", p=(1.0, 0, 0, 0), ids=base, name="base_contact"),
    ]
    observations = [
        obs_term("base_lin_vel", noise=(-0.1, 0.1)),
        obs_term("base_ang_vel", noise=(-0.2, 0.2)),
        obs_term("projected_gravity", noise=(-0.05, 0.05)),
        obs_term("generated_commands", name="velocity_commands"),
        obs_term("joint_pos_rel", noise=(-0.01, 0.01), name="joint_pos"),
        obs_term("joint_vel_rel", noise=(-1.5, 1.5), name="joint_vel"),
        obs_term("last_action", name="actions"),
        obs_term("height_scan", noise=(-0.1, 0.1), clip=(-1.0, 1.0), offset=0.5),
    ]
    return dict(rewards=rewards, terminations=terminations, observations=observations)


# ---- Unitree H1 (config/h1/rough_env_cfg.py:19-115; 19 joints, unitree.py:216-260) --------------------
H1_JOINTS = ["left_hip_yaw", "right_hip_yaw", "torso", "left_hip_roll", "right_hip_roll", "left_shoulder_pitch",
             "right_shoulder_pitch", "left_hip_pitch", "right_hip_pitch", "left_shoulder_roll", "right_shoulder_roll",
             "left_knee", "right_knee", "left_shoulder_yaw", "right_shoulder_yaw", "left_ankle", "right_ankle",
             "left_elbow", "right_elbow"]
H1_BODIES = ["pelvis", "left_hip_yaw_link", "right_hip_yaw_link", "torso_link", "left_hip_roll_link", "right_hip_roll_link",
             "left_shoulder_pitch_link", "right_shoulder_pitch_link", "left_hip_pitch_link", "right_hip_pitch_link",
             "left_shoulder_roll_link", "right_shoulder_roll_link", "left_knee_link", "right_knee_link",
             "left_shoulder_yaw_link", "right_shoulder_yaw_link", "left_ankle_link", "right_ankle_link", "left_elbow_link",
             "right_elbow_link"]
H1_STIFFNESS = {".*_hip_yaw": 150.0, ".*_hip_roll": 150.0, ".*_hip_pitch": 200.0, ".*_knee": 200.0, "torso": 200.0,
                ".*_ankle": 20.0, ".*_shoulder_pitch": 40.0, ".*_shoulder_roll": 40.0, ".*_shoulder_yaw": 40.0,
                ".*_elbow": 40.0}
H1_DAMPING = {".*_hip_yaw": 5.0, ".*_hip_roll": 5.0, ".*_hip_pitch": 5.0, ".*_knee": 5.0, "torso": 5.0, ".*_ankle": 4.0,
              ".*_shoulder_pitch": 10.0, ".*_shoulder_roll": 10.0, ".*_shoulder_yaw": 10.0, ".*_elbow": 10.0}
H1_EFFORT_LIMIT = {".*_hip_.*": 300.0, ".*_knee": 300.0, "torso": 300.0, ".*_ankle": 100.0, ".*_shoulder_.*": 300.0,
                   ".*_elbow": 300.0}
H1_VELOCITY_LIMIT = 100.0
H1_DEFAULT_JOINT_POS = {".*_hip_yaw": 0.0, ".*_hip_roll": 0.0, ".*_hip_pitch": -0.28, ".*_knee": 0.79, ".*_ankle": -0.52,
                        "torso": 0.0, ".*_shoulder_pitch": 0.28, ".*_shoulder_roll": 0.0, ".*_shoulder_yaw": 0.0,
                        ".*_elbow": 0.52}  # unitree.py:199-212
H1_SOFT_JOINT_POS_LIMIT_FACTOR = 0.9  # unitree.py:213
H1_VELOCITY_COMMAND = dict(VELOCITY_COMMAND, ranges=[(0.0, 1.0), (0.0, 0.0), (-1.0, 1.0), (-math.pi, math.pi)])  # h1/rough_env_cfg.py:108-110
H1_SCANNER_BODY = "torso_link"  # h1/rough_env_cfg.py:80-81


def h1_rough_terms(joints=H1_JOINTS, bodies=H1_BODIES):
    """config/h1/rough_env_cfg.py: base rewards minus lin_vel_z_l2 / undesired_contacts, plus termination_penalty,
    biped air time, feet_slide, yaw-frame tracking and three joint_deviation groups; base_contact on the torso."""
    feet = _ids(".*ankle_link", bodies)
    torso = _ids(".*torso_link", bodies)
    jd_hip = _ids([".*_hip_yaw", ".*_hip_roll"], joints)
    jd_arms = _ids([".*_shoulder_.*", ".*_elbow"], joints)
    jd_torso = _ids("torso", joints)
    ankles = _ids(".*_ankle", joints)
    rewards = [
        TermSpec("track_lin_vel_xy_yaw_frame_exp", 1.0, (0.5 ** 2, 0, 0, 0), name="track_lin_vel_xy_exp"),
        TermSpec("track_ang_vel_z_world_exp", 1.0, (0.5 ** 2, 0, 0, 0), name="track_ang_vel_z_exp"),
        TermSpec("ang_vel_xy_l2", -0.05, name="ang_vel_xy_l2"),
        TermSpec("joint_torques_l2", 0.0, name="dof_torques_l2"),
        TermSpec("joint_acc_l2", -1.25e-7, name="dof_acc_l2"),
        TermSpec("action_rate_l2", -0.005, name="action_rate_l2"),
        TermSpec("feet_air_time_positive_biped", 0.25, (0.4, 0, 0, 0), ids=feet, name="feet_air_time"),
        TermSpec("flat_orientation_l2", -1.0, name="flat_orientation_l2"),
        TermSpec("joint_pos_limits", -1.0, ids=ankles, name="dof_pos_limits"),
        TermSpec("is_terminated", -200.0, name="termination_penalty"),
        TermSpec("feet_slide", -0.25, ids=feet, ids2=feet, name="feet_slide"),
        TermSpec("joint_deviation_l1", -0.2, ids=jd_hip, name="joint_deviation_hip"),
        TermSpec("joint_deviation_l1", -0.2, ids=jd_arms, name="joint_deviation_arms"),
        TermSpec("joint_deviation_l1", -0.1, ids=jd_torso, name="joint_deviation_torso"),
    ]
    terminations = [
        TermSpec("time_out", flags=1, name="time_out"),
        TermSpec("illegal_contact", p=(1.0, 0, 0, 0), ids=torso, name="base_contact"),
    ]
    observations = anymal_c_rough_terms()["observations"]
    return dict(rewards=rewards, terminations=terminations, observations=observations)
