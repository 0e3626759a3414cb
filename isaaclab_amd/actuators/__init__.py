"""Actuator models (drop-ins for ``isaaclab.actuators``)."""

from .actuator_base import ActuatorBase
from .actuator_cfg import (ActuatorBaseCfg, ActuatorNetLSTMCfg, ActuatorNetMLPCfg, DCMotorCfg, DelayedPDActuatorCfg,
                           IdealPDActuatorCfg, ImplicitActuatorCfg, RemotizedPDActuatorCfg)
from .actuator_net import ActuatorNetLSTM, ActuatorNetMLP
from .actuator_pd import DCMotor, DelayBuffer, DelayedPDActuator, IdealPDActuator, ImplicitActuator, RemotizedPDActuator
