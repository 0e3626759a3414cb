"""Term library with the reference's module layout: ``isaaclab.envs.mdp`` (+ the task packages' extra terms)."""

from .observations import *  # noqa: F401,F403
from .rewards import *  # noqa: F401,F403
from .terminations import *  # noqa: F401,F403
from .events import (push_by_setting_velocity, reset_joints_by_offset, reset_joints_by_scale,  # noqa: F401
                     reset_root_state_uniform)
from .curriculums import terrain_levels_vel  # noqa: F401
