"""GPU: ``find_flat_patches`` on the HIP ray caster against the fixture recorded from the reference function (its Warp cast
replaced by the C oracle cast: ray-hit parity is against this build's own oracle)."""

import pytest
import torch

from conftest import assert_close
from test_oracle_flat_patches import CASES, load_fixture, torch_draw

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", list(CASES))
def test_find_flat_patches_matches_reference(tag):
    from isaaclab_amd.terrains.utils import find_flat_patches
    from isaaclab_amd.utils.warp import convert_to_warp_mesh

    g = load_fixture()
    num, radii, origin, xr, yr, zr, diff, seed = CASES[tag]
    mesh = convert_to_warp_mesh(g["verts"], g["tris"], "cuda:0")
    torch.manual_seed(seed)
    got = find_flat_patches(mesh, num, radii, origin, xr, yr, zr, diff, _draw=torch_draw)
    # same draws, same accept / reject decisions in every round, heights to the last bit of the interpolation
    assert_close(got, torch.from_numpy(g[f"{tag}_patches"]), atol=1e-6, what=f"{tag}: patches")


def test_find_flat_patches_device_draws_and_failure():
    from isaaclab_amd.terrains.utils import find_flat_patches
    from isaaclab_amd.utils.warp import convert_to_warp_mesh, raycast_mesh

    g = load_fixture()
    mesh = convert_to_warp_mesh(g["verts"], g["tris"], "cuda:0")
    patches = find_flat_patches(mesh, 64, [0.2, 0.4], (0.0, 0.0, 0.0), (-10.0, 10.0), (-10.0, 10.0), (-0.1, 0.3), 0.01)
    assert patches.shape == (64, 3)
    # every patch really is flat: cast its ring again
    angle = torch.linspace(0, 2 * torch.pi, 10, device="cuda:0")
    ring = torch.cat([torch.stack([r * angle.cos(), r * angle.sin(), torch.zeros_like(angle)], -1) for r in (0.2, 0.4)])
    starts = patches[:, None, :] + ring
    starts[..., 2] = 100.0
    dirs = torch.zeros_like(starts)
    dirs[..., 2] = -1.0
    z = raycast_mesh(starts, dirs, mesh)[0][..., 2]
    assert float((z.max(dim=1)[0] - z.min(dim=1)[0]).max()) <= 0.01 and float(z.min()) >= -0.1 and float(z.max()) <= 0.3
    with pytest.raises(RuntimeError, match="Failed to find valid patches"):
        # nothing on this terrain lies 5 m up: every round rejects every patch
        find_flat_patches(mesh, 2, 0.3, (0.0, 0.0, 5.0), (-1.0, 1.0), (-1.0, 1.0), (-0.5, 0.5), 0.05, _max_rounds=3)
