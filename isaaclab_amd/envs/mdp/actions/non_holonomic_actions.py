"""Planar base driven by forward speed and yaw rate (interface of ``isaaclab/envs/mdp/actions/non_holonomic_actions.py``):
``process_actions`` is the affine ``ilx_joint_action_process``, ``apply_actions`` one launch of
``ilx_non_holonomic_action_apply`` reading the body quaternion in place."""

from __future__ import annotations

from collections.abc import Sequence
from dataclasses import dataclass

import torch

from .... import _abi
from ....managers.action_manager import ActionTerm, ActionTermCfg
from ....utils import string as string_utils


@dataclass
class NonHolonomicActionCfg(ActionTermCfg):
    body_name: str = ""
    x_joint_name: str = ""
    y_joint_name: str = ""
    yaw_joint_name: str = ""
    scale: tuple[float, float] = (1.0, 1.0)
    offset: tuple[float, float] = (0.0, 0.0)

    def __post_init__(self):
        self.class_type = self.class_type or NonHolonomicAction


class NonHolonomicAction(ActionTerm):
    """non_holonomic_actions.py:23-159."""

    def __init__(self, cfg: NonHolonomicActionCfg, env):
        super().__init__(cfg, env)
        ids = {}
        for axis, name in (("x", cfg.x_joint_name), ("y", cfg.y_joint_name), ("yaw", cfg.yaw_joint_name)):
            joint_id, joint_name = self._asset.find_joints(name)
            if len(joint_id) != 1:
                raise ValueError(f"Expected a single joint match for the {axis} joint name: {name}, got {len(joint_id)}")
            ids[axis] = (joint_id[0], joint_name[0])
        self._body_idx, self._body_name = self._asset.find_bodies(cfg.body_name)
        if len(self._body_idx) != 1:
            raise ValueError(f"Found more than one body match for the body name: {cfg.body_name}")
        self._joint_ids = [ids["x"][0], ids["y"][0], ids["yaw"][0]]
        self._joint_names = [ids["x"][1], ids["y"][1], ids["yaw"][1]]
        n, dev = self.num_envs, self.device
        self._raw_actions = torch.zeros(n, 2, device=dev)
        self._processed_actions = torch.zeros_like(self._raw_actions)
        self._joint_vel_command = torch.zeros(n, 3, device=dev)
        self._scale = torch.tensor(cfg.scale, device=dev).unsqueeze(0).repeat(n, 1).contiguous()
        self._offset = torch.tensor(cfg.offset, device=dev).unsqueeze(0).repeat(n, 1).contiguous()
        self._clip = None
        if cfg.clip is not None:
            if not isinstance(cfg.clip, dict):
                raise ValueError(f"Unsupported clip type: {type(cfg.clip)}. Supported types are dict.")
            self._clip = torch.tensor([[-float("inf"), float("inf")]], device=dev).repeat(n, 2, 1)
            index_list, _, value_list = string_utils.resolve_matching_names_values(cfg.clip, self._joint_names)
            self._clip[:, index_list] = torch.tensor(value_list, device=dev)

    @property
    def action_dim(self) -> int:
        return 2

    @property
    def raw_actions(self) -> torch.Tensor:
        return self._raw_actions

    @property
    def processed_actions(self) -> torch.Tensor:
        return self._processed_actions

    def process_actions(self, actions):
        self._raw_actions[:] = actions
        _abi.check(_abi.joint_action_process(_abi.ptr(self._raw_actions), None, None, _abi.ptr(self._scale), 1.0, _abi.ptr(self._offset),
                                             0.0, _abi.ptr(self._clip), _abi.ptr(self._processed_actions), self.num_envs, 2,
                                             _abi.stream()), "ilx_joint_action_process")

    def apply_actions(self):
        body_state = self._asset.data.body_state_w          # (N, B, 13): the body's quaternion is columns 3..6 of its row
        quat = body_state[:, self._body_idx[0], 3:7]
        qp, qs = _abi.row_view(quat, 4)
        _abi.check(_abi.non_holonomic_action_apply(qp, qs, _abi.ptr(self._processed_actions), _abi.ptr(self._joint_vel_command),
                                                   self.num_envs, _abi.stream()), "ilx_non_holonomic_action_apply")
        self._asset.set_joint_velocity_target(self._joint_vel_command, joint_ids=self._joint_ids)

    def reset(self, env_ids: Sequence[int] | None = None) -> None:
        self._raw_actions[slice(None) if env_ids is None else env_ids] = 0.0
