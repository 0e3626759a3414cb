"""PD-family actuators on the HIP kernel ``ilx_actuator_pd`` (interface of ``isaaclab/actuators/actuator_pd.py``).

One launch per ``compute`` replaces the 6 (ideal PD) to 14 (DC motor) eager kernels of the reference.
"""

from __future__ import annotations

from collections.abc import Sequence

import torch

from .. import _abi
from ..utils.types import ArticulationActions
from .actuator_base import ActuatorBase

_IDEAL, _DC = 0, 1


def _launch_pd(model, act, control_action, joint_pos, joint_vel, vel_limit, sat):
    q_des = _abi.f32c(control_action.joint_positions)
    qd_des = None if control_action.joint_velocities is None else _abi.f32c(control_action.joint_velocities)
    tau_ff = None if control_action.joint_efforts is None else _abi.f32c(control_action.joint_efforts)
    q, qd = _abi.f32c(joint_pos), _abi.f32c(joint_vel)
    n, j = q.shape
    _abi.check(_abi.actuator_pd(
        model, _abi.ptr(q_des), _abi.ptr(qd_des), _abi.ptr(tau_ff), _abi.ptr(q), _abi.ptr(qd),
        _abi.ptr(act.stiffness), _abi.ptr(act.damping), _abi.ptr(act.effort_limit),
        None if vel_limit is None else _abi.ptr(vel_limit), 1, float(sat), _abi.ptr(act.computed_effort),
        _abi.ptr(act.applied_effort), n, j, _abi.stream()), "ilx_actuator_pd")


class ImplicitActuator(ActuatorBase):
    """actuator_pd.py:32-87: targets pass through; the PD torque is only estimated (for reward terms)."""

    def reset(self, *args, **kwargs):
        pass

    def compute(self, control_action, joint_pos, joint_vel):
        _launch_pd(_IDEAL, self, control_action, joint_pos, joint_vel, None, 0.0)
        return control_action


class IdealPDActuator(ActuatorBase):
    """actuator_pd.py:95-145: explicit PD torque with a symmetric effort clip."""

    def reset(self, env_ids: Sequence[int]):
        pass

    def compute(self, control_action, joint_pos, joint_vel):
        _launch_pd(_IDEAL, self, control_action, joint_pos, joint_vel, None, 0.0)
        control_action.joint_efforts = self.applied_effort
        control_action.joint_positions = None
        control_action.joint_velocities = None
        return control_action


class DCMotor(IdealPDActuator):
    """actuator_pd.py:148-233: PD torque clipped by the velocity dependent DC-motor envelope."""

    def __init__(self, cfg, *args, **kwargs):
        super().__init__(cfg, *args, **kwargs)
        self._saturation_effort = self.cfg.saturation_effort if self.cfg.saturation_effort is not None else torch.inf
        self._joint_vel = torch.zeros_like(self.computed_effort)
        if self.cfg.velocity_limit is None:
            raise ValueError("The velocity limit must be provided for the DC motor actuator model.")

    def compute(self, control_action, joint_pos, joint_vel):
        self._joint_vel = joint_vel  # kept for API compatibility; the kernel reads joint_vel directly
        _launch_pd(_DC, self, control_action, joint_pos, joint_vel, self.velocity_limit, self._saturation_effort)
        control_action.joint_efforts = self.applied_effort
        control_action.joint_positions = None
        control_action.joint_velocities = None
        return control_action
