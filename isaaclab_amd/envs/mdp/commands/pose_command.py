"""``UniformPoseCommand`` (interface of ``isaaclab/envs/mdp/commands/pose_command.py:25-127``)."""

from __future__ import annotations

import torch

from ....managers.command_manager import CommandTerm


class UniformPoseCommand(CommandTerm):
    """Pose command (x, y, z, qw, qx, qy, qz) in the root frame for one body of the asset.

    ``metrics["position_error"]`` / ``["orientation_error"]`` are the norms of ``compute_pose_error`` between the command
    moved into the world frame and the body's pose (pose_command.py:92-108); resampling draws position then roll, pitch,
    yaw -> ``quat_from_euler_xyz`` [-> ``quat_unique``] (:110-124).
    """

    metric_names = ("position_error", "orientation_error")

    def __init__(self, cfg, env):
        self.robot = env.scene[cfg.asset_name]
        self.body_idx = self.robot.find_bodies(cfg.body_name)[0][0]
        super().__init__(cfg, env)
        self.pose_command_b = self._mdp.command

    def _kernel_cfg(self) -> dict:
        c, r = self.cfg, self.cfg.ranges
        return dict(kind="pose", command_dim=7, num_bodies=self.robot.num_bodies, resampling_time_range=tuple(c.resampling_time_range),
                    ranges=[r.pos_x, r.pos_y, r.pos_z, r.roll, r.pitch, r.yaw], body_index=self.body_idx,
                    make_quat_unique=c.make_quat_unique)

    def _bind_state(self):
        data = getattr(self.robot, "data", self.robot)
        self._mdp.bind(root_state_w=data.root_state_w, body_state_w=data.body_state_w)

    @property
    def command(self) -> torch.Tensor:
        """The desired pose command. Shape is (num_envs, 7): position, then the quaternion in (w, x, y, z)."""
        return self.pose_command_b

    @property
    def pose_command_w(self) -> torch.Tensor:
        """The command in the world frame (pose_command.py:94-99), evaluated on demand."""
        from ....utils.math import combine_frame_transforms

        data = getattr(self.robot, "data", self.robot)
        pos, quat = combine_frame_transforms(data.root_state_w[:, :3].contiguous(), data.root_state_w[:, 3:7].contiguous(),
                                             self.pose_command_b[:, :3].contiguous(), self.pose_command_b[:, 3:].contiguous())
        return torch.cat([pos, quat], dim=-1)
