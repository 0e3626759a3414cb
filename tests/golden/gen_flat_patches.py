"""Golden vectors for ``find_flat_patches``, produced by the REFERENCE function (``isaaclab/terrains/utils.py:135-273``) with
its Warp ray cast replaced by this repository's C oracle cast (Warp is not installed).  The fixture keeps the mesh, the
arguments, the seed the draws came from and the patches found.  Writes tests/golden/flat_patches.npz."""

import os
import sys
from types import SimpleNamespace as NS

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import _refload as R  # noqa: E402
from oracle import raycast as ORC  # noqa: E402

R.bootstrap()
# terrain: 8 m x 8 m, flat terraces of different height with bumpy strips between them and a tall wall along one side
g = torch.Generator().manual_seed(606)
nx = 41
xs = np.arange(nx, dtype=np.float32) * np.float32(0.2) - np.float32(4.0)
h = np.zeros((nx, nx), dtype=np.float32)
h[:, 14:20] = (torch.rand(nx, 6, generator=g) * 0.4).numpy()
h[:, 20:] = 0.25
h[25:, :] += 0.5
h[:, -3:] += 2.0
vx, vy = np.meshgrid(xs, xs, indexing="ij")
verts = np.stack([vx.ravel(), vy.ravel(), h.ravel()], -1).astype(np.float32)
idx = np.arange(nx * nx).reshape(nx, nx)
a, b, c, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, 1:].ravel()
tris = np.concatenate([np.stack([a, b, c], -1), np.stack([b, d, c], -1)]).astype(np.int32)
cmesh = ORC.CMesh(verts, tris)
rounds = []


def oracle_cast(ray_starts, ray_directions, mesh=None, max_dist=1e6, **kw):
    hits, _, _ = ORC.raycast_c(ray_starts.reshape(-1, 3), ray_directions.reshape(-1, 3), cmesh, max_dist)
    rounds.append(ray_starts.shape[0])
    return hits.view(ray_starts.shape), None, None, None


sys.modules["warp"].device_to_torch = lambda dev: "cpu"
R.placeholder("isaaclab.utils.warp", raycast_mesh=oracle_cast)
R.placeholder("isaaclab.terrains")
tu = R.load("isaaclab.terrains.utils", f"{R.L}/terrains/utils.py")
wp_mesh = NS(device="cpu", points=NS(numpy=lambda: verts))

out = dict(verts=verts, tris=tris)
CASES = {
    # name: (num_patches, radii, origin, x_range, y_range, z_range, max_height_diff, seed)
    "one_radius": (48, 0.3, (0.0, 0.0, 0.0), (-10.0, 10.0), (-10.0, 10.0), (-1.0, 1.0), 0.05, 31),
    "two_radii": (32, [0.15, 0.45], (0.5, -0.5, 0.25), (-3.0, 3.0), (-3.5, 3.0), (-0.3, 0.6), 0.02, 32),
}
for tag, (num, radii, origin, xr, yr, zr, diff, seed) in CASES.items():
    rounds.clear()
    torch.manual_seed(seed)
    patches = tu.find_flat_patches(wp_mesh, num, radii, origin, xr, yr, zr, diff)
    out[f"{tag}_patches"] = patches.numpy()
    out[f"{tag}_rounds"] = np.array(rounds)
    print(tag, "rays per round:", rounds)
np.savez_compressed(os.path.join(HERE, "flat_patches.npz"), **out)
print("wrote flat_patches.npz with", len(out), "arrays")
