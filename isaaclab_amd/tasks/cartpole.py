"""Isaac-Cartpole-v0 term tables and constants (``isaaclab_tasks/manager_based/classic/cartpole/cartpole_env_cfg.py``)."""

from __future__ import annotations

import math

from ..envs.fused_mdp import TermSpec, obs_term

JOINTS = ["slider_to_cart", "cart_to_pole"]      # cartpole.py asset: one prismatic, one revolute joint
BODIES = ["rail", "cart", "pole"]
DECIMATION = 2                                   # :175
PHYSICS_DT = 1 / 120                             # :180
STEP_DT = PHYSICS_DT * DECIMATION
EPISODE_LENGTH_S = 5.0                           # :176
MAX_EPISODE_LENGTH = math.ceil(EPISODE_LENGTH_S / STEP_DT)
ACTION_SCALE = 100.0                             # JointEffortActionCfg on slider_to_cart (:60)


def cartpole_terms():
    """Rewards :116-136, terminations :144-149, observations :70-79 (no corruption)."""
    rewards = [
        TermSpec("is_alive", 1.0, name="alive"),
        TermSpec("is_terminated", -2.0, name="terminating"),
        TermSpec("joint_pos_target_l2", -1.0, (0.0, 0, 0, 0), ids=[1], name="pole_pos"),
        TermSpec("joint_vel_l1", -0.01, ids=[0], name="cart_vel"),
        TermSpec("joint_vel_l1", -0.005, ids=[1], name="pole_vel"),
    ]
    terminations = [
        TermSpec("time_out", flags=1, name="time_out"),
        TermSpec("joint_pos_out_of_manual_limit", p=(-3.0, 3.0, 0, 0), ids=[0], name="cart_out_of_bounds"),
    ]
    observations = [obs_term("joint_pos_rel"), obs_term("joint_vel_rel")]
    return dict(rewards=rewards, terminations=terminations, observations=observations)
