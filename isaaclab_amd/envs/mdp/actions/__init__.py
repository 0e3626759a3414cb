from .task_space_actions import (DifferentialInverseKinematicsAction, DifferentialInverseKinematicsActionCfg, OffsetCfg,  # noqa: F401
                                 OperationalSpaceControllerAction, OperationalSpaceControllerActionCfg)
from .joint_actions import (JointAction, JointActionCfg, JointEffortAction, JointEffortActionCfg, JointPositionAction,  # noqa: F401
                            JointPositionActionCfg, JointVelocityAction, JointVelocityActionCfg, RelativeJointPositionAction,
                            RelativeJointPositionActionCfg)
from .binary_joint_actions import (BinaryJointAction, BinaryJointActionCfg, BinaryJointPositionAction,  # noqa: F401
                                   BinaryJointPositionActionCfg, BinaryJointVelocityAction, BinaryJointVelocityActionCfg)
from .joint_actions_to_limits import (EMAJointPositionToLimitsAction, EMAJointPositionToLimitsActionCfg,  # noqa: F401
                                      JointPositionToLimitsAction, JointPositionToLimitsActionCfg)
from .non_holonomic_actions import NonHolonomicAction, NonHolonomicActionCfg  # noqa: F401
