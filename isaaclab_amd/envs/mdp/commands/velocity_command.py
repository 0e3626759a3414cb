"""``UniformVelocityCommand`` (interface of ``isaaclab/envs/mdp/commands/velocity_command.py:27-160``)."""

from __future__ import annotations

import torch

from ....managers.command_manager import CommandTerm


class UniformVelocityCommand(CommandTerm):
    """Base-frame velocity command (x, y, yaw rate) with optional heading control and standing envs.

    ``metrics["error_vel_xy"]`` / ``["error_vel_yaw"]`` accumulate as in velocity_command.py:111-121; resampling draws
    follow :123-138 (lin x, lin y, yaw rate, heading, heading-env coin, standing-env coin) and the post-processing
    :140-160 (heading error -> yaw rate, standing envs -> zero).
    """

    metric_names = ("error_vel_xy", "error_vel_yaw")

    def __init__(self, cfg, env):
        if cfg.heading_command and cfg.ranges.heading is None:  # velocity_command.py:62-66
            raise ValueError("The velocity command has heading commands active (heading_command=True) but the `ranges.heading`"
                             " parameter is set to None.")
        self.robot = env.scene[cfg.asset_name]
        super().__init__(cfg, env)
        m = self._mdp
        self.vel_command_b, self.heading_target = m.command, m.heading_target
        self.is_heading_env, self.is_standing_env = m.is_heading_env, m.is_standing_env

    def _kernel_cfg(self) -> dict:
        c, r = self.cfg, self.cfg.ranges
        heading = r.heading if r.heading is not None else (0.0, 0.0)
        return dict(kind="velocity", command_dim=3, resampling_time_range=tuple(c.resampling_time_range),
                    ranges=[r.lin_vel_x, r.lin_vel_y, r.ang_vel_z, heading],
                    rel_heading_envs=c.rel_heading_envs, rel_standing_envs=c.rel_standing_envs,
                    heading_control_stiffness=c.heading_control_stiffness, heading_command=c.heading_command)

    def _bind_state(self):
        data = getattr(self.robot, "data", self.robot)
        self._mdp.bind(root_state_w=data.root_state_w)

    @property
    def command(self) -> torch.Tensor:
        """The desired base velocity command in the base frame. Shape is (num_envs, 3)."""
        return self.vel_command_b
