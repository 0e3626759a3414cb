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


# ---- UFactory xArm6 with gripper: the articulation of the fork's own robot configuration (IsaacLab_UFactory_Xarm6_config at
# the reference's root: 6 arm + 2 finger joints, implicit actuators with stiffness 2000 / damping 100) with the kinematics
# of ROBOT/xarm/xarm6_with_gripper.urdf:79-225 (isaaclab_amd.synth.XARM6_JOINT_ORIGINS).  The fork registers no reach task
# for it; the task terms are the Franka task's, on the arm's last link.
XARM6_JOINTS = [f"joint{i}" for i in range(1, 7)] + ["left_finger_joint", "right_finger_joint"]
XARM6_BODIES = ["link_base"] + [f"link{i}" for i in range(1, 7)] + ["xarm_gripper_base_link", "left_finger", "right_finger"]
XARM6_EE_BODY = "link6"
XARM6_DEFAULT_JOINT_POS = {"joint1": 0.0, "joint2": -1.22, "joint3": -0.78, "joint4": 0.0, "joint5": 1.30, "joint6": 0.0,
                           "left_finger_joint": 0.0, "right_finger_joint": 0.0}
XARM6_ACTUATORS = dict(effort_limit={"joint[1-3]": 87.0, "joint[4-6]": 12.0, ".*_finger_joint": 200.0},
                       velocity_limit={"joint[1-6]": 2.61, ".*_finger_joint": 0.2},
                       stiffness={"joint[1-6]": 2000.0, ".*_finger_joint": 2e3}, damping={"joint[1-6]": 100.0, ".*_finger_joint": 1e2})
FRANKA_ACTUATORS = dict(effort_limit={"panda_joint[1-4]": 87.0, "panda_joint[5-7]": 12.0, "panda_finger_joint.*": 200.0},   # franka.py:50-72
                        velocity_limit={"panda_joint[1-4]": 2.175, "panda_joint[5-7]": 2.61, "panda_finger_joint.*": 0.2},
                        stiffness={"panda_joint.*": 80.0, "panda_finger_joint.*": 2e3},
                        damping={"panda_joint.*": 4.0, "panda_finger_joint.*": 1e2})
ROBOTS = {
    "franka": dict(joints=FRANKA_JOINTS, bodies=FRANKA_BODIES, ee_body=EE_BODY, arm_joints=["panda_joint.*"], actuators=FRANKA_ACTUATORS,
                   ik_body_offset=IK_BODY_OFFSET),
    "xarm6": dict(joints=XARM6_JOINTS, bodies=XARM6_BODIES, ee_body=XARM6_EE_BODY, arm_joints=["joint[1-6]"], actuators=XARM6_ACTUATORS,
                  ik_body_offset=None),   # the gripper base is fixed to link6 with no offset (urdf:311-315)
}


def reach_terms(bodies=FRANKA_BODIES, ee_body=EE_BODY):
    """Rewards :134-156, terminations :163, observations :112-124 (uniform noise on joint state)."""
    ee = [bodies.index(ee_body)]
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
