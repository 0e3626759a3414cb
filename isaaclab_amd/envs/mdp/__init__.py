"""Term library with the reference's module layout: ``isaaclab.envs.mdp`` (+ the task packages' extra terms)."""

from .observations import *  # noqa: F401,F403
from .rewards import *  # noqa: F401,F403
from .terminations import *  # noqa: F401,F403
from .events import (apply_external_force_torque, push_by_setting_velocity, reset_joints_by_offset,  # noqa: F401
                     reset_joints_by_scale, reset_root_state_from_terrain, reset_root_state_uniform,
                     reset_root_state_with_random_orientation,
                     reset_scene_to_default)
from .curriculums import modify_reward_weight, terrain_levels_vel  # noqa: F401
