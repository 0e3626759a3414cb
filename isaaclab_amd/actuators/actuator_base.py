"""Base class of the actuator models (interface of ``isaaclab/actuators/actuator_base.py:20-261``)."""

from __future__ import annotations

from abc import ABC, abstractmethod
from collections.abc import Sequence

import torch

from ..utils import string as string_utils
from ..utils.types import ArticulationActions


class ActuatorBase(ABC):
    """Parses per-joint parameters into ``(num_envs, num_joints)`` buffers and defines ``compute`` / ``reset``.

    Same constructor and attributes as the reference (``actuator_base.py:56-116``): ``stiffness``, ``damping``,
    ``armature``, ``friction``, ``effort_limit``, ``velocity_limit``, ``computed_effort``, ``applied_effort``.
    """

    def __init__(self, cfg, joint_names: list[str], joint_ids: slice | Sequence[int], num_envs: int, device: str,
                 stiffness: torch.Tensor | float = 0.0, damping: torch.Tensor | float = 0.0,
                 armature: torch.Tensor | float = 0.0, friction: torch.Tensor | float = 0.0,
                 effort_limit: torch.Tensor | float = torch.inf, velocity_limit: torch.Tensor | float = torch.inf):
        self.cfg = cfg
        self._num_envs = num_envs
        self._device = device
        self._joint_names = joint_names
        self._joint_indices = joint_ids
        self.stiffness = self._parse_joint_parameter(self.cfg.stiffness, stiffness)
        self.damping = self._parse_joint_parameter(self.cfg.damping, damping)
        self.armature = self._parse_joint_parameter(self.cfg.armature, armature)
        self.friction = self._parse_joint_parameter(self.cfg.friction, friction)
        self.effort_limit = self._parse_joint_parameter(self.cfg.effort_limit, effort_limit)
        self.velocity_limit = self._parse_joint_parameter(self.cfg.velocity_limit, velocity_limit)
        self.computed_effort = torch.zeros(self._num_envs, self.num_joints, device=self._device)
        self.applied_effort = torch.zeros_like(self.computed_effort)

    def __str__(self) -> str:
        ids = self.joint_indices
        if ids == slice(None):
            ids = list(range(self.num_joints))
        return (f"<class {self.__class__.__name__}> object:\n\tNumber of joints      : {self.num_joints}\n"
                f"\tJoint names expression: {self.cfg.joint_names_expr}\n\tJoint names           : {self.joint_names}\n"
                f"\tJoint indices         : {ids}\n")

    @property
    def num_joints(self) -> int:
        return len(self._joint_names)

    @property
    def joint_names(self) -> list[str]:
        return self._joint_names

    @property
    def joint_indices(self) -> slice | Sequence[int]:
        return self._joint_indices

    @abstractmethod
    def reset(self, env_ids: Sequence[int]):
        raise NotImplementedError

    @abstractmethod
    def compute(self, control_action: ArticulationActions, joint_pos: torch.Tensor,
                joint_vel: torch.Tensor) -> ArticulationActions:
        raise NotImplementedError

    def _parse_joint_parameter(self, cfg_value, default_value) -> torch.Tensor:
        """One (num_envs, num_joints) tensor per actuator parameter: the cfg entry (a number, or a dict of joint-name
        patterns) wins over the default handed down by the asset (a number or a full tensor).  Argument checks, exception
        types and messages as actuator_base.py:192-250."""
        shape = (self._num_envs, self.num_joints)
        names = self.joint_names
        if cfg_value is None and default_value is None:
            raise ValueError("The parameter value is None and no default value is provided.")
        if cfg_value is None:
            if isinstance(default_value, torch.Tensor):
                if tuple(default_value.shape) != shape:
                    raise ValueError(f"Invalid default value tensor shape.\nGot: {default_value.shape}\nExpected: {shape}")
                return default_value.float().to(self._device)
            if not isinstance(default_value, (float, int)):
                raise TypeError(f"Invalid type for default value: {type(default_value)} for actuator on joints {names}. "
                                "Expected float or Tensor.")
            return torch.full(shape, float(default_value), device=self._device)
        if isinstance(cfg_value, dict):
            param = torch.zeros(shape, device=self._device)
            columns, _, values = string_utils.resolve_matching_names_values(cfg_value, names)
            param[:, columns] = torch.tensor(values, dtype=torch.float, device=self._device)
            return param
        if not isinstance(cfg_value, (float, int)):
            raise TypeError(f"Invalid type for parameter value: {type(cfg_value)} for actuator on joints {names}. "
                            "Expected float or dict.")
        return torch.full(shape, float(cfg_value), device=self._device)
