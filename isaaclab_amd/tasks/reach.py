"""Isaac-Reach-Franka-v0 term tables and constants (``isaaclab_tasks/manager_based/manipulation/reach/reach_env_cfg.py``
with ``config/franka/joint_pos_env_cfg.py``; the IK variants swap the arm action for the DifferentialIK action term,
``config/franka/ik_rel_env_cfg.py:29-36`` / ``ik_abs_env_cfg.py:29-34``)."""

from __future__ import annotations

import math

from ..envs.fused_mdp import TermSpec, obs_term

FRANKA_JOINTS = [f"panda_joint{i}" for i in range(1, 8)] + ["panda_finger_joint1", "panda_finger_joint2"]
FRANKA_BODIES = ["panda_link0"] + [f"panda_link{i}" for i in range(1, 8)] + ["panda_hand", "panda_leftfinger", "panda_rightfinger"]
EE_BODY = "panda_hand"
DECIMATION = 2                                   # :203
PHYSICS_DT = 1.0 / 60.0                          # :208
STEP_DT = PHYSICS_DT * DECIMATION
EPISODE_LENGTH_S = 12.0                          # :205
MAX_EPISODE_LENGTH = math.ceil(EPISODE_LENGTH_S / STEP_DT)
ARM_ACTION_SCALE = 0.5                           # JointPositionActionCfg(scale=0.5, use_default_offset=True)
IK_BODY_OFFSET = ((0.0, 0.0, 0.107), (1.0, 0.0, 0.0, 0.0))
POSE_COMMAND = dict(kind="pose", resampling_time_range=(4.0, 4.0),      # :88-104 + franka override of pitch
                    ranges=[(0.35, 0.65), (-0.2, 0.2), (0.15, 0.5), (0.0, 0.0), (math.pi, math.pi), (-3.14, 3.14)],
                    make_quat_unique=False)


def reach_terms(bodies=FRANKA_BODIES):
    """Rewards :134-156, terminations :163, observations :112-124 (uniform noise on joint state)."""
    ee = [bodies.index(EE_BODY)]
    rewards = [
        TermSpec("position_command_error", -0.2, ids=ee, name="end_effector_position_tracking"),
        TermSpec("position_command_error_tanh", 0.1, (0.1, 0, 0, 0), ids=ee, name="end_effector_position_tracking_fine_grained"),
        TermSpec("orientation_command_error", -0.1, ids=ee, name="end_effector_orientation_tracking"),
        TermSpec("action_rate_l2", -0.0001, name="action_rate"),
        TermSpec("joint_vel_l2", -0.0001, name="joint_vel"),
    ]
    terminations = [TermSpec("time_out", flags=1, name="time_out")]
    observations = [obs_term("joint_pos_rel", noise=(-0.01, 0.01), name="joint_pos"),
                    obs_term("joint_vel_rel", noise=(-0.01, 0.01), name="joint_vel"),
                    obs_term("generated_commands", name="pose_command"), obs_term("last_action", name="actions")]
    return dict(rewards=rewards, terminations=terminations, observations=observations)
