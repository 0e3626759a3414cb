"""CPU: the height-field oracle against the reference's outputs; the float64 closed form against the ray-cast oracle
(two formulas that share nothing); what the library recognises as height-field patches (host-only probe)."""

import numpy as np
import torch

from conftest import load_golden
from oracle import heightfield as HF
from oracle import raycast as RC

CASES = ("noise", "noise_corrected", "stairs_corrected", "float_heights")


def test_mesh_matches_reference_function():
    g = load_golden("heightfield")
    for tag in CASES:
        hs, vs, thr = (float(v) for v in g[f"{tag}_params"])
        v, t = HF.height_field_to_mesh(g[f"{tag}_heights"].numpy(), hs, vs, None if thr < 0 else thr)
        assert v.dtype == np.float32
        assert np.array_equal(v, g[f"{tag}_vertices"].numpy()), f"{tag}: vertices differ from the reference's"
        assert np.array_equal(t.astype(np.int64), g[f"{tag}_triangles"].numpy()), f"{tag}: faces differ from the reference's"
    # the correction did move vertices in the corrected cases (otherwise they test nothing)
    v0, _ = HF.height_field_to_mesh(g["stairs_corrected_heights"].numpy(), 0.1, 0.005, None)
    assert (v0[:, :2] != g["stairs_corrected_vertices"].numpy()[:, :2]).any()


def hf_case(rows=33, cols=41, hs=0.1, seed=2, origin=(3.7, -12.9)):
    rng = np.random.default_rng(seed)
    heights = rng.integers(-10, 11, size=(rows, cols)).astype(np.int16)
    v, t = HF.height_field_to_mesh(heights, hs, 0.005)
    v = v + np.array([origin[0], origin[1], 0.0], dtype=np.float32)   # a tile placed somewhere on the map
    return v, t.astype(np.int32), rows, cols


def step_case(rows=33, cols=41, hs=0.1):
    """One 0.3 m step along x and one along y: the slope correction moves the vertices of one row and one column."""
    heights = np.zeros((rows, cols), dtype=np.int16)
    heights[20:, :] += 60
    heights[:, 27:] += 60
    v, t = HF.height_field_to_mesh(heights, hs, 0.005, slope_threshold=0.75)
    return v, t.astype(np.int32), heights


def test_closed_form_agrees_with_raycast_oracle():
    """Hit height (1e-5 m) and face index of vertical rays: float64 plane interpolation vs Moller-Trumbore."""
    v, t, rows, cols = hf_case()
    rng = np.random.default_rng(5)
    n = 4000
    x = rng.uniform(v[:, 0].min() - 0.2, v[:, 0].max() + 0.2, n).astype(np.float32)
    y = rng.uniform(v[:, 1].min() - 0.2, v[:, 1].max() + 0.2, n).astype(np.float32)
    starts = torch.from_numpy(np.stack([x, y, np.full(n, 20.0, np.float32)], -1))
    dirs = torch.tensor([[0.0, 0.0, -1.0]]).repeat(n, 1)
    hits, _, face = RC.raycast_brute(starts, dirs, torch.from_numpy(v), torch.from_numpy(t))
    z, f, clear = HF.closed_form_hit(v, rows, cols, x, y)
    hit = face.numpy() >= 0
    assert np.array_equal(hit[clear], f[clear] >= 0) and clear.sum() > 3000
    both = clear & hit
    assert np.array_equal(face.numpy()[both], f[both]), "face index"
    assert np.abs(hits[:, 2].numpy()[both] - z[both]).max() <= 1e-5, "hit height"
    # away from the footprint both say miss
    assert not hit[~np.isfinite(z)].any()


def test_lattice_probe_on_host():
    """What ilx_mesh_create recognises: regular cells of reference-ordered height fields; nothing in other meshes."""
    from isaaclab_amd.utils.mesh import lattice_probe
    from test_oracle_raycast import small_mesh

    v, t, rows, cols = hf_case()
    s = lattice_probe(v, t)
    assert s == {"patches": 1, "faces": t.shape[0], "cells": (rows - 1) * (cols - 1), "simple_cells": (rows - 1) * (cols - 1)}
    # another face order (or diagonal) is not a patch
    sv, st, _ = small_mesh()
    assert lattice_probe(sv.numpy(), st.numpy())["patches"] == 0
    # slope correction: cells with a moved corner, and the cells the stretched faces reach into, are not simple
    vc, tc, _ = step_case()
    sc = lattice_probe(vc, tc)
    assert sc["patches"] == 1 and sc["cells"] == 32 * 40
    assert 32 * 40 - 6 * 40 - 6 * 32 <= sc["simple_cells"] <= 32 * 40 - 2 * 40 - 2 * 32, sc
    # two tiles side by side + a triangle hovering over part of the first: their shared border column and the cells
    # under the triangle drop out
    v2 = v + np.array([0.0, (cols - 1) * 0.1, 0.0], dtype=np.float32)
    hover = np.array([[4.0, -12.0, 3.0], [5.0, -12.0, 3.0], [4.0, -11.0, 3.0]], dtype=np.float32)
    vv = np.concatenate([v, v2, hover])
    tt = np.concatenate([t, t + v.shape[0], np.array([[2 * v.shape[0], 2 * v.shape[0] + 1, 2 * v.shape[0] + 2]], dtype=np.int32)])
    s2 = lattice_probe(vv, tt)
    assert s2["patches"] == 2 and s2["cells"] == 2 * (rows - 1) * (cols - 1)
    lost = s2["cells"] - s2["simple_cells"]
    assert 2 * (rows - 1) + 50 <= lost <= 2 * (rows - 1) + 150, lost
