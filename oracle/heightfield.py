"""Oracle for height-field terrains (test infrastructure, CPU only; the product never imports it).

Two things live here:

* ``height_field_to_mesh`` restates ``convert_height_field_to_mesh``
  (``isaaclab/terrains/height_field/utils.py:79-173``): vertex order, the slope correction that moves vertices of
  steep cells, and the face order.  Pinned by ``tests/golden/heightfield.npz`` (outputs of the reference function).
* ``closed_form_hit`` computes, in float64 and from the height grid alone, where a vertical ray meets the mesh of an
  *uncorrected* height field: the cell under the ray, the half of the cell on the ray's side of the ind0-ind3
  diagonal (``utils.py:158-171``), and the plane through that half's three vertices.  It shares nothing with the
  Moller-Trumbore formula of the ray-cast oracle (``oracle/raycast.py``) or of the kernels, so it is an independent
  check of hit height and face index.  (The reference's own intersection, warp's ``mesh_query_ray``, is not
  available: ray hits stay "parity unpinned" against it.)
"""

from __future__ import annotations

import numpy as np


def height_field_to_mesh(height_field: np.ndarray, horizontal_scale: float, vertical_scale: float,
                         slope_threshold: float | None = None):
    """utils.py:119-173.  Returns (vertices (R*C, 3) float32, triangles (2*(R-1)*(C-1), 3) uint32)."""
    rows, cols = height_field.shape
    # utils.py:122-124: x follows the first index, y the second
    y = np.linspace(0, (cols - 1) * horizontal_scale, cols)
    x = np.linspace(0, (rows - 1) * horizontal_scale, rows)
    yy, xx = np.meshgrid(y, x)
    hf = height_field.copy()
    if slope_threshold is not None:
        # utils.py:129-150: a vertex whose neighbour is higher by more than the threshold moves one cell towards it
        thr = slope_threshold * horizontal_scale / vertical_scale
        move_x = np.zeros((rows, cols))
        move_y = np.zeros((rows, cols))
        move_c = np.zeros((rows, cols))
        move_x[:rows - 1, :] += hf[1:, :] - hf[:rows - 1, :] > thr
        move_x[1:, :] -= hf[:rows - 1, :] - hf[1:, :] > thr
        move_y[:, :cols - 1] += hf[:, 1:] - hf[:, :cols - 1] > thr
        move_y[:, 1:] -= hf[:, :cols - 1] - hf[:, 1:] > thr
        move_c[:rows - 1, :cols - 1] += hf[1:, 1:] - hf[:rows - 1, :cols - 1] > thr
        move_c[1:, 1:] -= hf[:rows - 1, :cols - 1] - hf[1:, 1:] > thr
        xx = xx + (move_x + move_c * (move_x == 0)) * horizontal_scale
        yy = yy + (move_y + move_c * (move_y == 0)) * horizontal_scale
    verts = np.zeros((rows * cols, 3), dtype=np.float32)
    verts[:, 0] = xx.flatten()
    verts[:, 1] = yy.flatten()
    verts[:, 2] = hf.flatten() * vertical_scale
    # utils.py:158-171: row by row, per cell (ind0, ind3, ind1) then (ind0, ind2, ind3)
    idx = np.arange(rows * cols, dtype=np.int64).reshape(rows, cols)
    i0, i1, i2, i3 = idx[:-1, :-1], idx[:-1, 1:], idx[1:, :-1], idx[1:, 1:]
    tris = np.stack([np.stack([i0, i3, i1], -1), np.stack([i0, i2, i3], -1)], axis=2).reshape(-1, 3).astype(np.uint32)
    return verts, tris


def closed_form_hit(verts: np.ndarray, rows: int, cols: int, x: np.ndarray, y: np.ndarray, edge_tol: float = 1e-4):
    """Vertical rays at (x, y) over the mesh of an uncorrected (rows, cols) height field with vertices ``verts`` in the
    reference order.  Returns float64 z, int64 face index (-1 outside), and a bool array marking the rays that keep at
    least ``edge_tol`` (as a fraction of the cell) away from every cell border and from the diagonal -- only for those
    is the face index unambiguous."""
    v = verts.astype(np.float64).reshape(rows, cols, 3)
    xs, ys = v[:, 0, 0], v[0, :, 1]
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    i = np.searchsorted(xs, x, side="right") - 1
    j = np.searchsorted(ys, y, side="right") - 1
    inside = (x >= xs[0]) & (x <= xs[-1]) & (y >= ys[0]) & (y <= ys[-1])
    i = np.clip(i, 0, rows - 2)
    j = np.clip(j, 0, cols - 2)
    fx = (x - xs[i]) / (xs[i + 1] - xs[i])
    fy = (y - ys[j]) / (ys[j + 1] - ys[j])
    z00, z01, z10, z11 = v[i, j, 2], v[i, j + 1, 2], v[i + 1, j, 2], v[i + 1, j + 1, 2]
    # the diagonal runs from (i, j) to (i+1, j+1); face 0 = (ind0, ind3, ind1) holds the corner (i, j+1): fy >= fx
    upper = fy >= fx
    z_upper = z00 + fx * (z11 - z01) + fy * (z01 - z00)   # plane through v00, v11, v01
    z_lower = z00 + fx * (z10 - z00) + fy * (z11 - z10)   # plane through v00, v10, v11
    z = np.where(upper, z_upper, z_lower)
    face = 2 * (i * (cols - 1) + j) + np.where(upper, 0, 1)
    clear = inside & (fx > edge_tol) & (fx < 1 - edge_tol) & (fy > edge_tol) & (fy < 1 - edge_tol) & (np.abs(fx - fy) > edge_tol)
    z = np.where(inside, z, np.inf)
    face = np.where(inside, face, -1)
    return z, face, clear
