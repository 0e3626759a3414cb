"""CPU: the flat-patch search oracle against the fixture recorded from the reference function."""

import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, assert_exact
from oracle import raycast as ORC
from oracle import terrain_patches as O

# name: (num_patches, radii, origin, x_range, y_range, z_range, max_height_diff, seed) as the fixture was recorded
CASES = {
    "one_radius": (48, 0.3, (0.0, 0.0, 0.0), (-10.0, 10.0), (-10.0, 10.0), (-1.0, 1.0), 0.05, 31),
    "two_radii": (32, [0.15, 0.45], (0.5, -0.5, 0.25), (-3.0, 3.0), (-3.5, 3.0), (-0.3, 0.6), 0.02, 32),
}


def load_fixture():
    with np.load(os.path.join(GOLDEN, "flat_patches.npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def torch_draw(count, lo, hi):
    """The reference's draw: ``torch.empty(count).uniform_(lo, hi)`` on the global CPU generator."""
    return torch.empty(count).uniform_(lo, hi)


@pytest.mark.parametrize("tag", list(CASES))
def test_find_flat_patches(tag):
    g = load_fixture()
    num, radii, origin, xr, yr, zr, diff, seed = CASES[tag]
    cmesh = ORC.CMesh(g["verts"], g["tris"])
    torch.manual_seed(seed)
    got = O.find_flat_patches(cmesh, g["verts"], num, radii, origin, xr, yr, zr, diff, torch_draw)
    assert_exact(got, torch.from_numpy(g[f"{tag}_patches"]), f"{tag}: patches")
