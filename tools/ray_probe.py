"""GPU probe: ray-cast traversal statistics and timing on the benchmark terrain (diagnostics, not a test)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from isaaclab_amd import _abi, synth
from isaaclab_amd.utils.mesh import convert_to_mesh, raycast_mesh

dev = torch.device("cuda:0")
verts, tris, h = synth.rough_terrain_mesh(seed=3)
t0 = time.time(); mesh = convert_to_mesh(verts, tris, "cuda:0"); print("build s", time.time() - t0, "nodes", mesh.num_nodes, "depth", mesh.depth)
print("grid", mesh.grid_stats())
g = torch.Generator().manual_seed(0)
n_env, R = 16384, 187
xy = torch.rand(n_env, 2, generator=g) * torch.tensor([78.0, 158.0]) + 1
gx, gy = torch.meshgrid(torch.arange(-0.8, 0.81, 0.1), torch.arange(-0.5, 0.51, 0.1), indexing="xy")
off = torch.stack([gx.flatten(), gy.flatten()], -1)
starts = torch.cat([(xy[:, None, :] + off[None]).reshape(-1, 2), torch.full((n_env * R, 1), 20.6)], -1).to(dev).contiguous()
dirs = torch.tensor([0, 0, -1.0]).repeat(n_env * R, 1).to(dev).contiguous()
hits = torch.empty_like(starts); cnt = torch.zeros(n_env * R, 2, dtype=torch.int32, device=dev)
_abi.check(_abi.raycast_mesh_stats(mesh.handle, _abi.ptr(starts), _abi.ptr(dirs), n_env * R, 1e6, _abi.ptr(hits), cnt.data_ptr(), _abi.stream()))
torch.cuda.synchronize()
c = cnt.float()
print("nodes/ray mean %.1f max %d ; tris/ray mean %.1f max %d" % (c[:, 0].mean(), c[:, 0].max(), c[:, 1].mean(), c[:, 1].max()))
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
print("terrain cast us", timeit(lambda: raycast_mesh(starts, dirs, mesh)))
print("terrain cast (column grid) us", timeit(lambda: raycast_mesh(starts, dirs, mesh, z_parallel_hint=True)))
a = raycast_mesh(starts, dirs, mesh, return_distance=True, return_face_id=True)
b = raycast_mesh(starts, dirs, mesh, return_distance=True, return_face_id=True, z_parallel_hint=True)
print("identical:", bool((a[1] == b[1]).all()), bool((a[3] == b[3]).all()))
pv = np.array([[-100, -100, 0], [300, -100, 0], [-100, 300, 0], [300, 300, 0]], dtype=np.float32)
plane = convert_to_mesh(pv, np.array([[0, 1, 2], [1, 3, 2]], dtype=np.int32), "cuda:0")
print("plane cast us", timeit(lambda: raycast_mesh(starts, dirs, plane)))
print("plane cast (column grid) us", timeit(lambda: raycast_mesh(starts, dirs, plane, z_parallel_hint=True)))
# the same scan pattern under random sensor yaw, as the bench has it
yaw = torch.rand(n_env, generator=g) * 6.2832
cs, sn = torch.cos(yaw)[:, None], torch.sin(yaw)[:, None]
rot = torch.stack([cs * off[None, :, 0] - sn * off[None, :, 1], sn * off[None, :, 0] + cs * off[None, :, 1]], -1)
starts_r = torch.cat([(xy[:, None, :] + rot).reshape(-1, 2), torch.full((n_env * R, 1), 20.6)], -1).to(dev).contiguous()
print("terrain cast, random yaw (column grid) us", timeit(lambda: raycast_mesh(starts_r, dirs, mesh, z_parallel_hint=True)))
a = raycast_mesh(starts_r, dirs, mesh, return_distance=True, return_face_id=True)
b = raycast_mesh(starts_r, dirs, mesh, return_distance=True, return_face_id=True, z_parallel_hint=True)
print("identical:", bool((a[1] == b[1]).all()), bool((a[3] == b[3]).all()))
# what env ordering is worth: the same rays with the envs visited in Morton order of their 2 m bin
def morton(ix, iy):
    key = torch.zeros_like(ix)
    for b in range(8):
        key |= ((ix >> b) & 1) << (2 * b) | ((iy >> b) & 1) << (2 * b + 1)
    return key
for bin_m in (8.0, 2.0, 0.5):
    order = torch.argsort(morton((xy[:, 0] / bin_m).long(), (xy[:, 1] / bin_m).long()))
    s_sorted = starts_r.view(n_env, R, 3)[order.to(dev)].reshape(-1, 3).contiguous()
    print("terrain cast, random yaw, envs in Morton order of %.1f m bins us" % bin_m,
          timeit(lambda: raycast_mesh(s_sorted, dirs, mesh, z_parallel_hint=True)))
