"""Actuator configuration records (field names follow ``isaaclab/actuators/actuator_cfg.py``)."""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Literal


@dataclass
class ActuatorBaseCfg:
    joint_names_expr: list[str] = field(default_factory=lambda: [".*"])
    effort_limit: dict[str, float] | float | None = None
    velocity_limit: dict[str, float] | float | None = None
    stiffness: dict[str, float] | float | None = None
    damping: dict[str, float] | float | None = None
    armature: dict[str, float] | float | None = None
    friction: dict[str, float] | float | None = None
    class_type: type | None = None


@dataclass
class ImplicitActuatorCfg(ActuatorBaseCfg):
    pass


@dataclass
class IdealPDActuatorCfg(ActuatorBaseCfg):
    pass


@dataclass
class DCMotorCfg(IdealPDActuatorCfg):
    saturation_effort: float | None = None


@dataclass
class DelayedPDActuatorCfg(IdealPDActuatorCfg):
    min_delay: int = 0
    max_delay: int = 0


@dataclass
class RemotizedPDActuatorCfg(DelayedPDActuatorCfg):
    """``joint_parameter_lookup``: rows of (joint angle, transmission ratio, maximum torque), angles ascending."""

    joint_parameter_lookup: list | None = None


@dataclass
class ActuatorNetMLPCfg(DCMotorCfg):
    """``network_file`` may be a path to a ``torch.save``'d state-dict of Linear layers, or ``network`` may carry
    the layers directly as ``{"weights": [...], "biases": [...], "activation": "softsign"}``."""

    network_file: str | None = None
    network: dict | None = None
    pos_scale: float = 1.0
    vel_scale: float = 1.0
    torque_scale: float = 1.0
    input_order: Literal["pos_vel", "vel_pos"] = "pos_vel"
    input_idx: list[int] = field(default_factory=lambda: [0, 1, 2])


@dataclass
class ActuatorNetLSTMCfg(DCMotorCfg):
    """``network_file`` may be a path to a ``torch.save``'d state-dict holding the ``lstm.*`` parameters of a
    ``torch.nn.LSTM`` and the ``<k>.weight`` / ``<k>.bias`` pairs of the Linear head after it, or ``network`` may carry them
    directly as ``{"lstm": {"weight_ih_l0": ..., ...}, "weights": [...], "biases": [...], "activation": "softsign"}``."""

    network_file: str | None = None
    network: dict | None = None
