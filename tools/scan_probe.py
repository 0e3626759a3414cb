"""GPU probe: the height scanner's refresh (ilx_raycaster_update, z-parallel hint) on the benchmark terrain with the
bench's geometry -- 16 384 envs at random positions and yaw, 187 rays each.  Prints the time per refresh; under
rocprofv3 --pmc it is the workload of tools/scan_pmc.sh (diagnostics, not a test)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isaaclab_amd import _abi, synth
from isaaclab_amd.utils.mesh import convert_to_mesh
from oracle import raycast as RC

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
verts, tris, h = synth.rough_terrain_mesh(seed=3)
mesh = convert_to_mesh(verts, tris, "cuda:0")
print("lattice", mesh.lattice_stats())
g = torch.Generator().manual_seed(0)
n, R = 16384, 187
xy = torch.rand(n, 2, generator=g) * torch.tensor([76.0, 156.0]) + 2
pos = torch.cat([xy, torch.full((n, 1), 0.6)], -1).to(dev).contiguous()
yaw = torch.rand(n, generator=g) * 6.2832 - 3.1416
quat = torch.stack([torch.cos(yaw / 2), torch.zeros(n), torch.zeros(n), torch.sin(yaw / 2)], -1).to(dev).contiguous()
s, d = RC.sensor_rays(*RC.grid_pattern((1.6, 1.0), 0.1), (0.0, 0.0, 20.0), (1.0, 0.0, 0.0, 0.0))
s, d = s.to(dev).contiguous(), d.to(dev).contiguous()
hits = torch.empty(n, R, 3, device=dev)


def refresh():
    _abi.check(_abi.raycaster_update(mesh.handle, _abi.ptr(pos), _abi.ptr(quat), _abi.ptr(s), _abi.ptr(d), None, 1, 1e6,
                                     _abi.ptr(hits), None, 0.0, n, R, _abi.RAYS_Z_PARALLEL, _abi.stream()))


if len(sys.argv) > 2 and sys.argv[2] == "count":   # not under --pmc: the counting launch is slow (one atomic per ray)
    mesh.lattice_counter(True)
    refresh()
    torch.cuda.synchronize()
    print("served by the height grid: %d of %d rays" % (mesh.lattice_counter(False), n * R))
refresh()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    refresh()
b.record()
torch.cuda.synchronize()
print("refresh us", a.elapsed_time(b) / reps * 1e3)
