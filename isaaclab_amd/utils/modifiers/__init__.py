"""Observation modifiers with the reference's module layout (``isaaclab/utils/modifiers``)."""

from .modifier_cfg import DigitalFilterCfg, IntegratorCfg, ModifierCfg  # noqa: F401
from .modifier_base import ModifierBase  # noqa: F401
from .modifier import DigitalFilter, Integrator, bias, clip, scale  # noqa: F401
