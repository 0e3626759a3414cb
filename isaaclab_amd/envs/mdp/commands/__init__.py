"""Command terms with the reference's module layout (``isaaclab/envs/mdp/commands``)."""

from .commands_cfg import NormalVelocityCommandCfg, NullCommandCfg, UniformPoseCommandCfg, UniformVelocityCommandCfg  # noqa: F401
from .null_command import NullCommand  # noqa: F401
from .pose_command import UniformPoseCommand  # noqa: F401
from .velocity_command import NormalVelocityCommand, UniformVelocityCommand  # noqa: F401
from .pose_2d_command import (TerrainBasedPose2dCommand, TerrainBasedPose2dCommandCfg, UniformPose2dCommand,  # noqa: F401
                              UniformPose2dCommandCfg)
