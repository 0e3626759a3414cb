"""Term configuration records (interface of ``isaaclab/managers/manager_term_cfg.py``)."""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable

from ..utils.noise.noise_cfg import UniformNoiseCfg  # noqa: F401  (isaaclab/utils/noise/noise_cfg.py)


@dataclass
class ManagerTermBaseCfg:
    func: Callable = None
    params: dict = field(default_factory=dict)


@dataclass
class RewardTermCfg(ManagerTermBaseCfg):
    weight: float = 0.0


@dataclass
class TerminationTermCfg(ManagerTermBaseCfg):
    time_out: bool = False


@dataclass
class ObservationTermCfg(ManagerTermBaseCfg):
    modifiers: list | None = None   # ModifierCfg records, applied in order before noise, clip and scale
    noise: UniformNoiseCfg | None = None
    clip: tuple[float, float] | None = None
    scale: float | None = None
    history_length: int = 0
    flatten_history_dim: bool = True


@dataclass
class ObservationGroupCfg:
    """A group is any object whose attributes (or a dict whose values) are :class:`ObservationTermCfg`."""
    concatenate_terms: bool = True
    enable_corruption: bool = False
    history_length: int | None = None      # when set, overrides every term's history_length / flatten_history_dim
    flatten_history_dim: bool = True
