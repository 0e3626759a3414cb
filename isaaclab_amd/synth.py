"""Synthetic stand-ins for what the closed simulator provides: terrain meshes and physics state tensors.

PhysX and the USD terrain importer are outside this package; benchmarks and tests need tensors of the same
shape and statistics instead.  Everything is generated on the CPU from an explicit seed and is deterministic.
"""

from __future__ import annotations

import numpy as np
import torch


def heightfield_mesh(heights: np.ndarray, horizontal_scale: float, origin=(0.0, 0.0)):
    """Triangulate a height grid (nx, ny): vertices on the grid, two triangles per cell (same vertex / face
    counts as the height-field tiles of the reference's terrain generator: 81 x 81 -> 12,800 faces)."""
    nx, ny = heights.shape
    xs = origin[0] + np.arange(nx, dtype=np.float32) * np.float32(horizontal_scale)
    ys = origin[1] + np.arange(ny, dtype=np.float32) * np.float32(horizontal_scale)
    gx, gy = np.meshgrid(xs, ys, indexing="ij")
    verts = np.stack([gx, gy, heights.astype(np.float32)], axis=-1).reshape(-1, 3)
    idx = np.arange(nx * ny, dtype=np.int32).reshape(nx, ny)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, 1:].ravel()
    tris = np.concatenate([np.stack([a, b, c], -1), np.stack([b, d, c], -1)]).astype(np.int32)
    return verts, tris


def rough_terrain_mesh(seed: int = 0, rows: int = 10, cols: int = 20, tile: float = 8.0, hscale: float = 0.1,
                       border: float = 20.0):
    """A rows x cols field of height-field tiles (random uniform noise, pyramid slopes, stairs) with a flat border,
    mirroring the layout statistics of the reference's rough terrain (10 x 20 tiles of 8 m, 0.1 m cells).

    Returns (vertices (V,3) float32, triangles (F,3) int32, height_at) where ``height_at(x, y)`` gives the
    analytic grid height used to place robots (nearest grid vertex).
    """
    rng = np.random.default_rng(seed)
    n = int(round(tile / hscale))  # cells per tile edge
    nx, ny = rows * n + 1, cols * n + 1
    h = np.zeros((nx, ny), dtype=np.float32)
    for r in range(rows):
        for c in range(cols):
            kind = (r + c) % 3
            level = (r + 1) / rows
            sl = (slice(r * n, (r + 1) * n + 1), slice(c * n, (c + 1) * n + 1))
            ii, jj = np.meshgrid(np.arange(n + 1), np.arange(n + 1), indexing="ij")
            if kind == 0:  # uniform noise, step 0.02, range up to 0.10
                amp = 0.02 + 0.08 * level
                t = np.round(rng.uniform(-amp, amp, size=(n + 1, n + 1)) / 0.02) * 0.02
            elif kind == 1:  # pyramid slope up to 0.4
                slope = 0.4 * level
                dist = np.minimum(np.minimum(ii, n - ii), np.minimum(jj, n - jj)) * hscale
                t = slope * dist
            else:  # stairs, step height 0.05..0.23, 0.3 m wide
                step = 0.05 + 0.18 * level
                ring = np.minimum(np.minimum(ii, n - ii), np.minimum(jj, n - jj)) // 3
                t = step * np.minimum(ring, 8)
            # tile edges at zero so that neighbouring tiles join
            t = t.astype(np.float32)
            t[0, :] = t[-1, :] = 0.0
            t[:, 0] = t[:, -1] = 0.0
            h[sl] = np.maximum(h[sl], t)
    verts, tris = heightfield_mesh(h, hscale)
    # flat border ring made of large triangles
    x0, y0, x1, y1 = 0.0, 0.0, rows * tile, cols * tile
    bx0, by0, bx1, by1 = x0 - border, y0 - border, x1 + border, y1 + border
    quads = [((bx0, by0), (bx1, y0)), ((bx0, y1), (bx1, by1)), ((bx0, y0), (x0, y1)), ((x1, y0), (bx1, y1))]
    bv, bt = [], []
    base = verts.shape[0]
    for (qx0, qy0), (qx1, qy1) in quads:
        k = base + len(bv)
        bv += [(qx0, qy0, 0.0), (qx1, qy0, 0.0), (qx0, qy1, 0.0), (qx1, qy1, 0.0)]
        bt += [(k, k + 1, k + 2), (k + 1, k + 3, k + 2)]
    verts = np.concatenate([verts, np.asarray(bv, dtype=np.float32)])
    tris = np.concatenate([tris, np.asarray(bt, dtype=np.int32)])

    def height_at(x, y):
        i = np.clip(np.round(np.asarray(x) / hscale).astype(np.int64), 0, nx - 1)
        j = np.clip(np.round(np.asarray(y) / hscale).astype(np.int64), 0, ny - 1)
        return h[i, j]

    return verts, tris, height_at
