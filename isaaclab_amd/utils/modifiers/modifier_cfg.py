"""Modifier configuration records (``isaaclab/utils/modifiers/modifier_cfg.py``)."""

from __future__ import annotations

from collections.abc import Callable
from dataclasses import dataclass, field
from typing import Any


@dataclass
class ModifierCfg:
    """modifier_cfg.py:16-33: a function (or ModifierBase subclass) and its keyword parameters."""

    func: Callable | None = None
    params: dict[str, Any] = field(default_factory=dict)


@dataclass
class DigitalFilterCfg(ModifierCfg):
    """modifier_cfg.py:36-64.  ``A``: weights of the past outputs, ``B``: weights of the current and past inputs."""

    A: list[float] | None = None
    B: list[float] | None = None

    def __post_init__(self):
        if self.func is None:
            from .modifier import DigitalFilter

            self.func = DigitalFilter


@dataclass
class IntegratorCfg(ModifierCfg):
    """modifier_cfg.py:67-78."""

    dt: float | None = None

    def __post_init__(self):
        if self.func is None:
            from .modifier import Integrator

            self.func = Integrator
