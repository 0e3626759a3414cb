"""``find_flat_patches`` (interface of ``isaaclab/terrains/utils.py:135-273``) on the HIP ray caster: every round of the
rejection sampling is three launches (ray set-up, z-parallel cast on the column grid, validity test)."""

from __future__ import annotations

import ctypes
from ctypes import c_float, c_int32, c_int64

import numpy as np
import torch

from .. import _abi
from ..utils import math as math_utils
from ..utils.mesh import Mesh, raycast_mesh

_rays = _abi._sig("ilx_flat_patch_rays", _abi.P, _abi.P, _abi.P, _abi.P, c_int64, c_int32, _abi.P, _abi.P, _abi.P, _abi.P)
_check = _abi._sig("ilx_flat_patch_check", _abi.P, _abi.P, c_int64, c_int32, c_float, c_float, c_float, _abi.P, _abi.P, _abi.P)


def find_flat_patches(wp_mesh: Mesh, num_patches: int, patch_radius: float | list[float], origin, x_range: tuple[float, float],
                      y_range: tuple[float, float], z_range: tuple[float, float], max_height_diff: float, *, _draw=None, _max_rounds: int = 10000) -> torch.Tensor:
    """utils.py:135-273.  ``_draw(count, lo, hi)`` (no counterpart in the reference) supplies the uniform samples of a round
    (x first, then y); by default they come from the device sampler.  ``_max_rounds``: the reference's limit of 10 000 rounds."""
    device = wp_mesh.device
    if isinstance(patch_radius, float):
        patch_radius = [patch_radius]
    if isinstance(origin, np.ndarray):
        origin = torch.from_numpy(origin).to(torch.float)
    elif not isinstance(origin, torch.Tensor):
        origin = torch.tensor(origin, dtype=torch.float)
    ox, oy, oz = (float(v) for v in origin.cpu())
    # the search rectangle is clipped to the mesh's bounding box (:196-207)
    x_range = (max(x_range[0] + ox, float(wp_mesh.bounds_min[0])), min(x_range[1] + ox, float(wp_mesh.bounds_max[0])))
    y_range = (max(y_range[0] + oy, float(wp_mesh.bounds_min[1])), min(y_range[1] + oy, float(wp_mesh.bounds_max[1])))
    z_range = (z_range[0] + oz, z_range[1] + oz)
    # ring of ten points per radius around (0, 0) (:211-220); set-up constants, computed once on the host
    angle = torch.linspace(0, 2 * np.pi, 10)
    ring = torch.cat([torch.stack([r * torch.cos(angle), r * torch.sin(angle), torch.zeros(10)], dim=-1) for r in patch_radius]).to(device)
    q = ring.shape[0]
    if _draw is None:
        def _draw(count, lo, hi):
            return math_utils.sample_uniform(lo, hi, (count,), device)

    points_ids = torch.arange(num_patches, device=device)
    flat_patches = torch.zeros(num_patches, 3, device=device)
    iter_count = 0
    while len(points_ids) > 0 and iter_count < _max_rounds:
        k = len(points_ids)
        pos_x, pos_y = _abi.f32c(_draw(k, *x_range).to(device)), _abi.f32c(_draw(k, *y_range).to(device))
        starts, dirs = torch.empty(k * q, 3, device=device), torch.empty(k * q, 3, device=device)
        _abi.check(_rays(_abi.ptr(pos_x), _abi.ptr(pos_y), _abi.ptr(ring), points_ids.data_ptr(), k, q, _abi.ptr(flat_patches),
                         _abi.ptr(starts), _abi.ptr(dirs), _abi.stream()), "ilx_flat_patch_rays")
        ray_hits = raycast_mesh(starts, dirs, wp_mesh, z_parallel_hint=True)[0]
        not_valid = torch.empty(k, dtype=torch.bool, device=device)
        _abi.check(_check(_abi.ptr(ray_hits), points_ids.data_ptr(), k, q, float(z_range[0]), float(z_range[1]), float(max_height_diff),
                          _abi.ptr(flat_patches), not_valid.data_ptr(), _abi.stream()), "ilx_flat_patch_check")
        points_ids = points_ids[not_valid].contiguous()
        iter_count += 1
    if len(points_ids) > 0:   # :265-273
        raise RuntimeError("Failed to find valid patches! Please check the input parameters."
                           f"\n\tMaximum number of iterations reached: {iter_count}"
                           f"\n\tNumber of invalid patches: {len(points_ids)}"
                           f"\n\tMaximum height difference: {max_height_diff}")
    return flat_patches - origin.to(device)
