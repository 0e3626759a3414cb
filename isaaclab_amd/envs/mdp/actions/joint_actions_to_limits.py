"""Position actions scaled to the joint limits (interface of ``isaaclab/envs/mdp/actions/joint_actions_to_limits.py`` and
the matching records of ``actions_cfg.py``): one launch of ``ilx_joint_action_to_limits`` per ``process_actions``."""

from __future__ import annotations

from collections.abc import Sequence
from dataclasses import dataclass, field

import torch

from .... import _abi
from ....managers.action_manager import ActionTerm, ActionTermCfg
from ....utils import string as string_utils


@dataclass
class JointPositionToLimitsActionCfg(ActionTermCfg):
    joint_names: list[str] = field(default_factory=lambda: [".*"])
    scale: float | dict[str, float] = 1.0
    rescale_to_limits: bool = True

    def __post_init__(self):
        self.class_type = self.class_type or JointPositionToLimitsAction


class JointPositionToLimitsAction(ActionTerm):
    """joint_actions_to_limits.py:23-135."""

    def __init__(self, cfg: JointPositionToLimitsActionCfg, env):
        super().__init__(cfg, env)
        self._joint_ids, self._joint_names = self._asset.find_joints(cfg.joint_names)
        self._num_joints = len(self._joint_ids)
        n, dev = self.num_envs, self.device
        self._joint_ids_dev = torch.tensor(self._joint_ids, dtype=torch.int32, device=dev)
        if self._num_joints == self._asset.num_joints:
            self._joint_ids = slice(None)
        self._raw_actions = torch.zeros(n, self.action_dim, device=dev)
        self._processed_actions = torch.zeros_like(self._raw_actions)
        if isinstance(cfg.scale, (float, int)):
            self._scale = float(cfg.scale)
        elif isinstance(cfg.scale, dict):
            self._scale = torch.ones(n, self.action_dim, device=dev)
            index_list, _, value_list = string_utils.resolve_matching_names_values(cfg.scale, self._joint_names)
            self._scale[:, index_list] = torch.tensor(value_list, device=dev)
        else:
            raise ValueError(f"Unsupported scale type: {type(cfg.scale)}. Supported types are float and dict.")
        self._clip = None
        if cfg.clip is not None:
            if not isinstance(cfg.clip, dict):
                raise ValueError(f"Unsupported clip type: {type(cfg.clip)}. Supported types are dict.")
            self._clip = torch.tensor([[-float("inf"), float("inf")]], device=dev).repeat(n, self.action_dim, 1)
            index_list, _, value_list = string_utils.resolve_matching_names_values(cfg.clip, self._joint_names)
            self._clip[:, index_list] = torch.tensor(value_list, device=dev)
        self._alpha, self._prev_applied_actions = None, None

    @property
    def action_dim(self) -> int:
        return self._num_joints

    @property
    def raw_actions(self) -> torch.Tensor:
        return self._raw_actions

    @property
    def processed_actions(self) -> torch.Tensor:
        return self._processed_actions

    def process_actions(self, actions: torch.Tensor):
        self._raw_actions[:] = actions
        scale_t, scale_s = (_abi.ptr(self._scale), 0.0) if isinstance(self._scale, torch.Tensor) else (None, self._scale)
        if isinstance(self._alpha, torch.Tensor):
            alpha_t, alpha_s = _abi.ptr(self._alpha), 0.0
        else:
            alpha_t, alpha_s = None, 1.0 if self._alpha is None else float(self._alpha)
        limits = self._asset.data.soft_joint_pos_limits
        _abi.check(_abi.joint_action_to_limits(_abi.ptr(self._raw_actions), scale_t, scale_s, _abi.ptr(self._clip),
                                               int(self.cfg.rescale_to_limits), _abi.ptr(limits), _abi.ptr(self._joint_ids_dev),
                                               limits.shape[1], alpha_t, alpha_s, _abi.ptr(self._prev_applied_actions),
                                               _abi.ptr(self._processed_actions), self.num_envs, self.action_dim, _abi.stream()),
                   "ilx_joint_action_to_limits")

    def apply_actions(self):
        self._asset.set_joint_position_target(self.processed_actions, joint_ids=self._joint_ids)

    def reset(self, env_ids: Sequence[int] | None = None) -> None:
        self._raw_actions[slice(None) if env_ids is None else env_ids] = 0.0


@dataclass
class EMAJointPositionToLimitsActionCfg(JointPositionToLimitsActionCfg):
    alpha: float | dict[str, float] = 1.0

    def __post_init__(self):
        self.class_type = self.class_type or EMAJointPositionToLimitsAction


class EMAJointPositionToLimitsAction(JointPositionToLimitsAction):
    """joint_actions_to_limits.py:138-229."""

    def __init__(self, cfg: EMAJointPositionToLimitsActionCfg, env):
        super().__init__(cfg, env)
        if isinstance(cfg.alpha, float):
            if not 0.0 <= cfg.alpha <= 1.0:
                raise ValueError(f"Moving average weight must be in the range [0, 1]. Got {cfg.alpha}.")
            self._alpha = cfg.alpha
        elif isinstance(cfg.alpha, dict):
            self._alpha = torch.ones((env.num_envs, self.action_dim), device=self.device)
            index_list, names_list, value_list = string_utils.resolve_matching_names_values(cfg.alpha, self._joint_names)
            for name, value in zip(names_list, value_list):
                if not 0.0 <= value <= 1.0:
                    raise ValueError(f"Moving average weight must be in the range [0, 1]. Got {value} for joint {name}.")
            self._alpha[:, index_list] = torch.tensor(value_list, device=self.device)
        else:
            raise ValueError(f"Unsupported moving average weight type: {type(cfg.alpha)}. Supported types are float and dict.")
        self._prev_applied_actions = torch.zeros_like(self.processed_actions)

    def reset(self, env_ids: Sequence[int] | None = None) -> None:
        rows = slice(None) if env_ids is None else torch.as_tensor(env_ids, device=self.device)
        self._raw_actions[rows] = 0.0
        self._prev_applied_actions[rows] = self._asset.data.joint_pos[:, self._joint_ids][rows]
