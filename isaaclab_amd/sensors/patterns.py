"""Ray patterns (interface of ``isaaclab/sensors/ray_caster/patterns/patterns.py`` and ``patterns_cfg.py``)."""

from __future__ import annotations

import math
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


def _unit_from_angles(elevation: torch.Tensor, azimuth: torch.Tensor) -> torch.Tensor:
    """Unit vectors (z up) from elevation above the horizontal plane and azimuth about z, both in radians."""
    ce = torch.cos(elevation)
    return torch.stack([ce * torch.cos(azimuth), ce * torch.sin(azimuth), torch.sin(elevation)], dim=-1)


def lidar_pattern(cfg: "LidarPatternCfg", device) -> tuple[torch.Tensor, torch.Tensor]:
    """patterns.py:136-179: ``channels`` elevation rings times the azimuth sweep, rays from the sensor origin.

    Flattening order: elevation major, azimuth minor.  A full 360 degree sweep drops its last azimuth (it would
    repeat the first)."""
    lo, hi = cfg.horizontal_fov_range
    elev = torch.deg2rad(torch.linspace(cfg.vertical_fov_range[0], cfg.vertical_fov_range[1], cfg.channels))
    count = math.ceil((hi - lo) / cfg.horizontal_res)
    azim = torch.linspace(lo, hi, count)
    if abs(abs(lo - hi) - 360.0) < 1e-6:
        azim = azim[:-1]
    azim = torch.deg2rad(azim)
    dirs = _unit_from_angles(elev[:, None].expand(-1, azim.numel()), azim[None, :].expand(elev.numel(), -1)).reshape(-1, 3)
    return torch.zeros_like(dirs).to(device), dirs.to(device)


def bpearl_pattern(cfg: "BpearlPatternCfg", device) -> tuple[torch.Tensor, torch.Tensor]:
    """patterns.py:105-133: the RS-Bpearl dome.  The reference builds the polar angle ``pitch + 90 deg`` and negates
    the vector: direction = -(sin p cos y, sin p sin y, cos p) with p = pitch + pi/2.  Flattening order: azimuth
    major, ring minor (``meshgrid(v, h, indexing="xy")``)."""
    azim = torch.arange(-cfg.horizontal_fov / 2, cfg.horizontal_fov / 2, cfg.horizontal_res)
    rings = torch.tensor(list(cfg.vertical_ray_angles))
    polar = torch.deg2rad(rings[None, :].expand(azim.numel(), -1).reshape(-1)) + torch.pi / 2
    yaw = torch.deg2rad(azim[:, None].expand(-1, rings.numel()).reshape(-1))
    sp = torch.sin(polar)
    dirs = -torch.stack([sp * torch.cos(yaw), sp * torch.sin(yaw), torch.cos(polar)], dim=1)
    return torch.zeros_like(dirs).to(device), dirs.to(device)


def pinhole_camera_pattern(cfg: "PinholeCameraPatternCfg", intrinsic_matrices: torch.Tensor, device) -> tuple[torch.Tensor, torch.Tensor]:
    """patterns.py:61-102: one ray through the centre of every pixel, per camera, in the robotics camera frame
    (x forward, y left, z up).  ``intrinsic_matrices`` (N, 3, 3); returns (N, H*W, 3) starts (zeros) and directions,
    pixels in row-major image order (u fastest)."""
    k_inv = torch.inverse(intrinsic_matrices.to("cpu", torch.float32))
    u = torch.arange(cfg.width, dtype=torch.float32) + 0.5
    v = torch.arange(cfg.height, dtype=torch.float32) + 0.5
    uu = u[None, :].expand(cfg.height, -1).reshape(-1)
    vv = v[:, None].expand(-1, cfg.width).reshape(-1)
    pix = torch.stack([uu, vv, torch.ones_like(uu)], dim=0)              # (3, H*W) homogeneous pixel centres
    cam = torch.matmul(k_inv, pix)                                       # (N, 3, H*W): x right, y down, z forward
    robo = torch.stack([cam[:, 2], -cam[:, 0], -cam[:, 1]], dim=1)       # x forward, y left, z up
    dirs = (robo / torch.norm(robo, dim=1, keepdim=True)).permute(0, 2, 1).contiguous()
    return torch.zeros_like(dirs).to(device), dirs.to(device)


@dataclass
class LidarPatternCfg(PatternBaseCfg):
    """patterns_cfg.py: number of elevation channels, elevation range, azimuth range and resolution (degrees)."""

    channels: int = 16
    vertical_fov_range: tuple[float, float] = (-15.0, 15.0)
    horizontal_fov_range: tuple[float, float] = (0.0, 360.0)
    horizontal_res: float = 0.2
    func: Callable = lidar_pattern


# fmt: off
_BPEARL_RING_ANGLES = tuple(89.5 - 2.8125 * i for i in range(32))   # 89.5, 86.6875, ..., 2.3125 degrees
# fmt: on


@dataclass
class BpearlPatternCfg(PatternBaseCfg):
    """patterns_cfg.py: azimuth sweep and resolution (degrees) and the 32 ring angles of the RS-Bpearl."""

    horizontal_fov: float = 360.0
    horizontal_res: float = 10.0
    vertical_ray_angles: tuple = _BPEARL_RING_ANGLES
    func: Callable = bpearl_pattern


@dataclass
class PinholeCameraPatternCfg(PatternBaseCfg):
    """patterns_cfg.py: image size and the USD-style lens parameters the intrinsic matrix is built from."""

    focal_length: float = 24.0
    horizontal_aperture: float = 20.955
    vertical_aperture: float | None = None
    horizontal_aperture_offset: float = 0.0
    vertical_aperture_offset: float = 0.0
    width: int = 0
    height: int = 0
    func: Callable = pinhole_camera_pattern

    @classmethod
    def from_intrinsic_matrix(cls, intrinsic_matrix, width: int, height: int, focal_length: float = 24.0):
        """patterns_cfg.py:122-170: lens parameters from (f_x, 0, c_x, 0, f_y, c_y, 0, 0, 1)."""
        f_x, c_x, f_y, c_y = intrinsic_matrix[0], intrinsic_matrix[2], intrinsic_matrix[4], intrinsic_matrix[5]
        return cls(focal_length=focal_length, horizontal_aperture=width * focal_length / f_x,
                   vertical_aperture=height * focal_length / f_y, horizontal_aperture_offset=(c_x - width / 2) / f_x,
                   vertical_aperture_offset=(c_y - height / 2) / f_y, width=width, height=height)
