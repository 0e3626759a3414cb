"""``TimestampedBuffer`` (``isaaclab/utils/buffers/timestamped_buffer.py``)."""

from __future__ import annotations

from dataclasses import dataclass

import torch


@dataclass
class TimestampedBuffer:
    """timestamped_buffer.py:12-30: a tensor and the simulation time it was computed at (lazy refresh of derived data)."""

    data: torch.Tensor = None  # type: ignore
    timestamp: float = -1.0
