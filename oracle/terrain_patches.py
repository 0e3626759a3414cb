"""CPU restatement of ``find_flat_patches`` (test infrastructure only; never imported by the product).

Follows ``isaaclab/terrains/utils.py:135-273`` with the ray cast done by the C oracle (``oracle/raycast.py``); draws come from
``draw(count, lo, hi)``.  Pinned by tests/golden/flat_patches.npz (recorded from the reference function over the same cast:
Warp itself is absent, so the hits themselves are this build's, see the ray-cast oracle).
"""

from __future__ import annotations

import math

import torch

from . import raycast as ORC


def find_flat_patches(cmesh, verts, num_patches, patch_radius, origin, x_range, y_range, z_range, max_height_diff, draw):
    radii = [patch_radius] if isinstance(patch_radius, float) else list(patch_radius)
    origin = torch.tensor(origin, dtype=torch.float32)
    x_range = (max(x_range[0] + origin[0].item(), verts[:, 0].min()), min(x_range[1] + origin[0].item(), verts[:, 0].max()))
    y_range = (max(y_range[0] + origin[1].item(), verts[:, 1].min()), min(y_range[1] + origin[1].item(), verts[:, 1].max()))
    z_range = (z_range[0] + origin[2].item(), z_range[1] + origin[2].item())
    angle = torch.linspace(0, 2 * math.pi, 10)
    ring = torch.cat([torch.stack([r * torch.cos(angle), r * torch.sin(angle), torch.zeros(10)], dim=-1) for r in radii])
    ids, patches, rounds = torch.arange(num_patches), torch.zeros(num_patches, 3), 0
    while len(ids) > 0 and rounds < 10000:
        patches[ids, 0], patches[ids, 1] = draw(len(ids), *x_range), draw(len(ids), *y_range)
        points = patches[ids].unsqueeze(1) + ring
        points[..., 2] = 100.0
        dirs = torch.zeros_like(points)
        dirs[..., 2] = -1.0
        hits, _, _ = ORC.raycast_c(points.view(-1, 3), dirs.view(-1, 3), cmesh, 1e6)
        heights = hits.view(points.shape)[..., 2]
        patches[ids, 2] = heights[..., -1]
        bad = ((heights < z_range[0]) | (heights > z_range[1])).any(dim=1) | ((heights.max(dim=1)[0] - heights.min(dim=1)[0]) > max_height_diff)
        ids, rounds = ids[bad], rounds + 1
    if len(ids) > 0:
        raise RuntimeError("Failed to find valid patches!")
    return patches - origin
