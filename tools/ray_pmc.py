"""GPU probe for counter passes: only the z-parallel cast of the bench's scan pattern under random yaw (diagnostics)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isaaclab_amd import synth
from isaaclab_amd.utils.mesh import convert_to_mesh, raycast_mesh

dev = torch.device("cuda:0")
verts, tris, h = synth.rough_terrain_mesh(seed=3)
mesh = convert_to_mesh(verts, tris, "cuda:0")
g = torch.Generator().manual_seed(0)
n_env, R = 16384, 187
xy = torch.rand(n_env, 2, generator=g) * torch.tensor([78.0, 158.0]) + 1
gx, gy = torch.meshgrid(torch.arange(-0.8, 0.81, 0.1), torch.arange(-0.5, 0.51, 0.1), indexing="xy")
off = torch.stack([gx.flatten(), gy.flatten()], -1)
yaw = torch.rand(n_env, generator=g) * 6.2832
cs, sn = torch.cos(yaw)[:, None], torch.sin(yaw)[:, None]
rot = torch.stack([cs * off[None, :, 0] - sn * off[None, :, 1], sn * off[None, :, 0] + cs * off[None, :, 1]], -1)
starts = torch.cat([(xy[:, None, :] + rot).reshape(-1, 2), torch.full((n_env * R, 1), 20.6)], -1).to(dev).contiguous()
dirs = torch.tensor([0, 0, -1.0]).repeat(n_env * R, 1).to(dev).contiguous()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    raycast_mesh(starts, dirs, mesh, z_parallel_hint=True)
torch.cuda.synchronize()
print("done")
