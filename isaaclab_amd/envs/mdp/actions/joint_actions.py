"""Joint action terms (interface of ``isaaclab/envs/mdp/actions/joint_actions.py`` and the matching records of
``actions_cfg.py``): ``processed = raw * scale + offset`` [clamped], one launch of ``ilx_joint_action_process``."""

from __future__ import annotations

from collections.abc import Sequence
from dataclasses import dataclass, field

import torch

from .... import _abi
from ....managers.action_manager import ActionTerm, ActionTermCfg
from ....utils import string as string_utils


@dataclass
class JointActionCfg(ActionTermCfg):
    """actions_cfg.py (JointActionCfg)."""

    joint_names: list[str] = field(default_factory=lambda: [".*"])
    scale: float | dict[str, float] = 1.0
    offset: float | dict[str, float] = 0.0
    preserve_order: bool = False


class JointAction(ActionTerm):
    """joint_actions.py:24-142."""

    def __init__(self, cfg: JointActionCfg, env):
        super().__init__(cfg, env)
        self._joint_ids, self._joint_names = self._asset.find_joints(cfg.joint_names, preserve_order=cfg.preserve_order)
        self._num_joints = len(self._joint_ids)
        if self._num_joints == self._asset.num_joints and not cfg.preserve_order:  # :71-73
            self._joint_ids = slice(None)
        self._raw_actions = torch.zeros(self.num_envs, self.action_dim, device=self.device)
        self._processed_actions = torch.zeros_like(self._raw_actions)
        # scale and offset: one number for every joint stays a number (the kernel takes it as a scalar argument), a
        # {joint-name regex: value} table becomes a per-joint row over `neutral`; clip: a table of (low, high) pairs only
        self._scale = self._per_joint("scale", cfg.scale, 1.0, numbers=True)
        self._offset = self._per_joint("offset", cfg.offset, 0.0, numbers=True)
        self._clip = None if cfg.clip is None else self._per_joint("clip", cfg.clip, (-float("inf"), float("inf")), numbers=False)

    def _per_joint(self, what: str, spec, neutral, numbers: bool):
        """One parameter of the term's cfg as the kernel wants it (joint_actions.py:78-108: same types accepted, same
        errors raised)."""
        if numbers and isinstance(spec, (float, int)):
            return float(spec)
        if not isinstance(spec, dict):
            accepted = "float and dict" if numbers else "dict"
            raise ValueError(f"Unsupported {what} type: {type(spec)}. Supported types are {accepted}.")
        shape = (self.num_envs, self.action_dim) if numbers else (self.num_envs, self.action_dim, len(neutral))
        table = torch.empty(shape, dtype=torch.float32, device=self.device)
        table[...] = torch.tensor(neutral, dtype=torch.float32, device=self.device)
        columns, _, values = string_utils.resolve_matching_names_values(spec, self._joint_names)
        table[:, columns] = torch.tensor(values, dtype=torch.float32, device=self.device)
        return table

    @property
    def action_dim(self) -> int:
        return self._num_joints

    @property
    def raw_actions(self) -> torch.Tensor:
        return self._raw_actions

    @property
    def processed_actions(self) -> torch.Tensor:
        return self._processed_actions

    def _launch(self, raw, action, prev_action):
        def split(v):
            return (_abi.ptr(v.contiguous()), 0.0) if isinstance(v, torch.Tensor) else (None, float(v))

        (scale_t, scale_s), (offset_t, offset_s) = split(self._scale), split(self._offset)
        _abi.check(_abi.joint_action_process(_abi.ptr(raw), _abi.ptr(action), _abi.ptr(prev_action), scale_t, scale_s, offset_t,
                                             offset_s, _abi.ptr(self._clip), _abi.ptr(self._processed_actions), self.num_envs,
                                             self.action_dim, _abi.stream()), "ilx_joint_action_process")

    def process_actions(self, actions: torch.Tensor):
        self._raw_actions[:] = actions
        self._launch(self._raw_actions, None, None)

    def _process_with_manager(self, actions, action, prev_action):
        """The sole term of a manager: the manager's ``prev_action <- action <- raw`` happens in the same launch."""
        self._raw_actions[:] = actions
        self._launch(self._raw_actions, action, prev_action)

    def reset(self, env_ids: Sequence[int] | None = None) -> None:
        self._raw_actions[slice(None) if env_ids is None else env_ids] = 0.0


@dataclass
class JointPositionActionCfg(JointActionCfg):
    use_default_offset: bool = True

    def __post_init__(self):
        self.class_type = self.class_type or JointPositionAction


class JointPositionAction(JointAction):
    """joint_actions.py:145-160."""

    def __init__(self, cfg: JointPositionActionCfg, env):
        super().__init__(cfg, env)
        if cfg.use_default_offset:
            self._offset = self._asset.data.default_joint_pos[:, self._joint_ids].clone()

    def apply_actions(self):
        self._asset.set_joint_position_target(self.processed_actions, joint_ids=self._joint_ids)


@dataclass
class RelativeJointPositionActionCfg(JointActionCfg):
    use_zero_offset: bool = True

    def __post_init__(self):
        self.class_type = self.class_type or RelativeJointPositionAction


class RelativeJointPositionAction(JointAction):
    """joint_actions.py:163-193."""

    def __init__(self, cfg: RelativeJointPositionActionCfg, env):
        super().__init__(cfg, env)
        if cfg.use_zero_offset:
            self._offset = 0.0

    def apply_actions(self):
        current_actions = self.processed_actions + self._asset.data.joint_pos[:, self._joint_ids]
        self._asset.set_joint_position_target(current_actions, joint_ids=self._joint_ids)


@dataclass
class JointVelocityActionCfg(JointActionCfg):
    use_default_offset: bool = True

    def __post_init__(self):
        self.class_type = self.class_type or JointVelocityAction


class JointVelocityAction(JointAction):
    """joint_actions.py:196-211."""

    def __init__(self, cfg: JointVelocityActionCfg, env):
        super().__init__(cfg, env)
        if cfg.use_default_offset:
            self._offset = self._asset.data.default_joint_vel[:, self._joint_ids].clone()

    def apply_actions(self):
        self._asset.set_joint_velocity_target(self.processed_actions, joint_ids=self._joint_ids)


@dataclass
class JointEffortActionCfg(JointActionCfg):
    def __post_init__(self):
        self.class_type = self.class_type or JointEffortAction


class JointEffortAction(JointAction):
    """joint_actions.py:214-225."""

    def apply_actions(self):
        self._asset.set_joint_effort_target(self.processed_actions, joint_ids=self._joint_ids)
