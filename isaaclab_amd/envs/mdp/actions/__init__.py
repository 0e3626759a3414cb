from .task_space_actions import DifferentialInverseKinematicsAction, DifferentialInverseKinematicsActionCfg, OffsetCfg  # noqa: F401
from .joint_actions import (JointAction, JointActionCfg, JointEffortAction, JointEffortActionCfg, JointPositionAction,  # noqa: F401
                            JointPositionActionCfg, JointVelocityAction, JointVelocityActionCfg, RelativeJointPositionAction,
                            RelativeJointPositionActionCfg)
