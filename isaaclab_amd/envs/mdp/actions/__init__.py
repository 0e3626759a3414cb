from .task_space_actions import DifferentialInverseKinematicsAction, DifferentialInverseKinematicsActionCfg, OffsetCfg  # noqa: F401
