"""Actuator models (drop-ins for ``isaaclab.actuators``)."""

from .actuator_base import ActuatorBase
from .actuator_cfg import (ActuatorBaseCfg, ActuatorNetLSTMCfg, ActuatorNetMLPCfg, DCMotorCfg, IdealPDActuatorCfg,
                           ImplicitActuatorCfg)
from .actuator_net import ActuatorNetLSTM, ActuatorNetMLP
from .actuator_pd import DCMotor, IdealPDActuator, ImplicitActuator
