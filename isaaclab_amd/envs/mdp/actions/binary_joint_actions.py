"""Open / close actions (interface of ``isaaclab/envs/mdp/actions/binary_joint_actions.py``): one launch of
``ilx_binary_joint_action`` per ``process_actions``."""

from __future__ import annotations

from collections.abc import Sequence
from dataclasses import dataclass, field

import torch

from .... import _abi
from ....managers.action_manager import ActionTerm, ActionTermCfg
from ....utils import string as string_utils


@dataclass
class BinaryJointActionCfg(ActionTermCfg):
    joint_names: list[str] = field(default_factory=list)
    open_command_expr: dict[str, float] = field(default_factory=dict)
    close_command_expr: dict[str, float] = field(default_factory=dict)


class BinaryJointAction(ActionTerm):
    """binary_joint_actions.py:23-134."""

    def __init__(self, cfg: BinaryJointActionCfg, env):
        super().__init__(cfg, env)
        self._joint_ids, self._joint_names = self._asset.find_joints(cfg.joint_names)
        self._num_joints = len(self._joint_ids)
        n, dev = self.num_envs, self.device
        self._raw_actions = torch.zeros(n, 1, device=dev)
        self._processed_actions = torch.zeros(n, self._num_joints, device=dev)

        def command(expr):
            out = torch.zeros(self._num_joints, device=dev)
            index_list, name_list, value_list = string_utils.resolve_matching_names_values(expr, self._joint_names)
            if len(index_list) != self._num_joints:
                raise ValueError(f"Could not resolve all joints for the action term. Missing: {set(self._joint_names) - set(name_list)}")
            out[index_list] = torch.tensor(value_list, device=dev)
            return out

        self._open_command, self._close_command = command(cfg.open_command_expr), command(cfg.close_command_expr)
        self._clip = None
        if cfg.clip is not None:
            if not isinstance(cfg.clip, dict):
                raise ValueError(f"Unsupported clip type: {type(cfg.clip)}. Supported types are dict.")
            # as in the reference (:88-94) the clip tensor is as wide as the action (1), so only the first joint's name can
            # carry a range, and that range is applied to every joint of the term
            clip = torch.tensor([[-float("inf"), float("inf")]], device=dev).repeat(n, self.action_dim, 1)
            index_list, _, value_list = string_utils.resolve_matching_names_values(cfg.clip, self._joint_names)
            clip[:, index_list] = torch.tensor(value_list, device=dev)
            self._clip = clip.expand(n, self._num_joints, 2).contiguous()

    @property
    def action_dim(self) -> int:
        return 1

    @property
    def raw_actions(self) -> torch.Tensor:
        return self._raw_actions

    @property
    def processed_actions(self) -> torch.Tensor:
        return self._processed_actions

    def process_actions(self, actions: torch.Tensor):
        self._raw_actions[:] = actions
        if actions.dtype == torch.bool:
            flags = actions.reshape(self.num_envs).contiguous()
            raw_f, raw_b = None, flags.data_ptr()
        else:
            raw_f, raw_b = _abi.ptr(self._raw_actions), None
        _abi.check(_abi.binary_joint_action(raw_f, raw_b, _abi.ptr(self._open_command), _abi.ptr(self._close_command),
                                            _abi.ptr(self._clip), _abi.ptr(self._processed_actions), self.num_envs, self._num_joints,
                                            _abi.stream()), "ilx_binary_joint_action")

    def reset(self, env_ids: Sequence[int] | None = None) -> None:
        self._raw_actions[slice(None) if env_ids is None else env_ids] = 0.0


@dataclass
class BinaryJointPositionActionCfg(BinaryJointActionCfg):
    def __post_init__(self):
        self.class_type = self.class_type or BinaryJointPositionAction


class BinaryJointPositionAction(BinaryJointAction):
    def apply_actions(self):
        self._asset.set_joint_position_target(self._processed_actions, joint_ids=self._joint_ids)


@dataclass
class BinaryJointVelocityActionCfg(BinaryJointActionCfg):
    def __post_init__(self):
        self.class_type = self.class_type or BinaryJointVelocityAction


class BinaryJointVelocityAction(BinaryJointAction):
    def apply_actions(self):
        self._asset.set_joint_velocity_target(self._processed_actions, joint_ids=self._joint_ids)
