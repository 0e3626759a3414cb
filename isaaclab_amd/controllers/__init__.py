from .differential_ik import DifferentialIKController, DifferentialIKControllerCfg  # noqa: F401
