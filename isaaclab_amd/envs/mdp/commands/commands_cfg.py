"""Command term configuration records (``isaaclab/envs/mdp/commands/commands_cfg.py``)."""

from __future__ import annotations

import math
from dataclasses import dataclass, field

from ....managers.command_manager import CommandTermCfg


@dataclass
class UniformVelocityCommandCfg(CommandTermCfg):
    """commands_cfg.py (UniformVelocityCommandCfg)."""

    @dataclass
    class Ranges:
        lin_vel_x: tuple[float, float] = (0.0, 0.0)
        lin_vel_y: tuple[float, float] = (0.0, 0.0)
        ang_vel_z: tuple[float, float] = (0.0, 0.0)
        heading: tuple[float, float] | None = None

    asset_name: str = "robot"
    heading_command: bool = False
    heading_control_stiffness: float = 1.0
    rel_standing_envs: float = 0.0
    rel_heading_envs: float = 1.0
    ranges: Ranges = field(default_factory=Ranges)

    def __post_init__(self):
        if self.class_type is None:
            from .velocity_command import UniformVelocityCommand

            self.class_type = UniformVelocityCommand


@dataclass
class NormalVelocityCommandCfg(UniformVelocityCommandCfg):
    """commands_cfg.py:99-129."""

    @dataclass
    class Ranges:
        mean_vel: tuple[float, float, float] = (0.0, 0.0, 0.0)
        std_vel: tuple[float, float, float] = (0.0, 0.0, 0.0)
        zero_prob: tuple[float, float, float] = (0.0, 0.0, 0.0)

    heading_command: bool = False
    ranges: Ranges = field(default_factory=Ranges)

    def __post_init__(self):
        if self.class_type is None:
            from .velocity_command import NormalVelocityCommand

            self.class_type = NormalVelocityCommand


@dataclass
class NullCommandCfg(CommandTermCfg):
    """commands_cfg.py:22-32: never resamples."""

    def __post_init__(self):
        self.resampling_time_range = (math.inf, math.inf)
        if self.class_type is None:
            from .null_command import NullCommand

            self.class_type = NullCommand


@dataclass
class UniformPoseCommandCfg(CommandTermCfg):
    """commands_cfg.py (UniformPoseCommandCfg)."""

    @dataclass
    class Ranges:
        pos_x: tuple[float, float] = (0.0, 0.0)
        pos_y: tuple[float, float] = (0.0, 0.0)
        pos_z: tuple[float, float] = (0.0, 0.0)
        roll: tuple[float, float] = (0.0, 0.0)
        pitch: tuple[float, float] = (0.0, 0.0)
        yaw: tuple[float, float] = (0.0, 0.0)

    asset_name: str = "robot"
    body_name: str = ""
    make_quat_unique: bool = False
    ranges: Ranges = field(default_factory=Ranges)

    def __post_init__(self):
        if self.class_type is None:
            from .pose_command import UniformPoseCommand

            self.class_type = UniformPoseCommand
