"""Ray patterns (interface of ``isaaclab/sensors/ray_caster/patterns/patterns.py`` and ``patterns_cfg.py``)."""

from __future__ import annotations

from dataclasses import dataclass
from typing import Callable

import torch


def grid_pattern(cfg: "GridPatternCfg", device) -> tuple[torch.Tensor, torch.Tensor]:
    """patterns.py:16-58: parallel rays on a regular grid in the sensor frame.

    With ``ordering="xy"`` x varies fastest.  The coordinates come from the same ``torch.arange`` call as the
    reference (on CPU, then moved), so the fp32 values of the grid are identical.
    """
    if cfg.ordering not in ["xy", "yx"]:
        raise ValueError(f"Ordering must be 'xy' or 'yx'. Received: '{cfg.ordering}'.")
    if cfg.resolution <= 0:
        raise ValueError(f"Resolution must be greater than 0. Received: '{cfg.resolution}'.")
    indexing = cfg.ordering if cfg.ordering == "xy" else "ij"
    x = torch.arange(start=-cfg.size[0] / 2, end=cfg.size[0] / 2 + 1.0e-9, step=cfg.resolution)
    y = torch.arange(start=-cfg.size[1] / 2, end=cfg.size[1] / 2 + 1.0e-9, step=cfg.resolution)
    grid_x, grid_y = torch.meshgrid(x, y, indexing=indexing)
    num_rays = grid_x.numel()
    ray_starts = torch.zeros(num_rays, 3)
    ray_starts[:, 0] = grid_x.flatten()
    ray_starts[:, 1] = grid_y.flatten()
    ray_directions = torch.zeros_like(ray_starts)
    ray_directions[..., :] = torch.tensor(list(cfg.direction))
    return ray_starts.to(device), ray_directions.to(device)


@dataclass
class PatternBaseCfg:
    func: Callable = None


@dataclass
class GridPatternCfg(PatternBaseCfg):
    """patterns_cfg.py: resolution [m], size (length, width) [m], ray direction, flattening order."""

    resolution: float = 0.1
    size: tuple[float, float] = (1.0, 1.0)
    direction: tuple[float, float, float] = (0.0, 0.0, -1.0)
    ordering: str = "xy"
    func: Callable = grid_pattern
