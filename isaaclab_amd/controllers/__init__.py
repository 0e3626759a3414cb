from .differential_ik import DifferentialIKController, DifferentialIKControllerCfg  # noqa: F401
from .operational_space import OperationalSpaceController, OperationalSpaceControllerCfg  # noqa: F401
