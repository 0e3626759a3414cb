"""Small data containers shared by assets and actuators (mirrors ``isaaclab/utils/types.py:16``)."""

from __future__ import annotations

from collections.abc import Sequence
from dataclasses import dataclass

import torch


@dataclass
class ArticulationActions:
    """Joint-level commands passed between the articulation and its actuator models."""

    joint_positions: torch.Tensor | None = None
    joint_velocities: torch.Tensor | None = None
    joint_efforts: torch.Tensor | None = None
    joint_indices: torch.Tensor | Sequence[int] | slice | None = None
