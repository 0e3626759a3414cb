"""Golden vectors for RayCasterCamera, produced by the REFERENCE class
(``isaaclab/sensors/ray_caster/ray_caster_camera.py`` with the reference's pinhole pattern, ``CameraData`` and
``isaaclab.utils.math``) on a stand-in view.  The Warp ray cast it calls is not available: ``raycast_mesh`` is replaced
by this build's C oracle (hits, depth and normals of the closest triangle), so the fixture pins everything around the
cast -- camera poses and offsets in three conventions, world rays, the two distance images with the three clipping
rules, frame counters, ``set_world_poses`` / ``set_world_poses_from_view`` / ``set_intrinsic_matrices`` -- while the hit
itself stays this build's.  Writes tests/golden/ray_caster_camera.npz."""

import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import _refload as R  # noqa: E402
from oracle import raycast as ORC  # noqa: E402

NS = types.SimpleNamespace
m = R.bootstrap()
physx = sys.modules["omni.physics.tensors.impl.api"]
physx.ArticulationView = type("ArticulationView", (), {})
physx.RigidBodyView = type("RigidBodyView", (), {})
R.placeholder("omni.timeline")
sys.modules["isaacsim.core.prims"].XFormPrim = type("XFormPrim", (), {})
R.placeholder("isaacsim.core.utils.stage", get_stage_up_axis=lambda: "Z")

# terrain the cameras look at: a bumpy 4 m x 4 m height field with a wall, so that there are hits at all ranges and misses
g = torch.Generator().manual_seed(7411)
nx = 21
h = (torch.rand(nx, nx, generator=g) * 0.3).numpy().astype(np.float32)
h[:, -3:] += 1.5
xs = np.arange(nx, dtype=np.float32) * np.float32(0.2) - np.float32(2.0)
vx, vy = np.meshgrid(xs, xs, indexing="ij")
verts = np.stack([vx.ravel(), vy.ravel(), h.ravel()], -1).astype(np.float32)
idx = np.arange(nx * nx).reshape(nx, nx)
a, b, c, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, 1:].ravel()
tris = np.concatenate([np.stack([a, b, c], -1), np.stack([b, d, c], -1)]).astype(np.int32)
cmesh = ORC.CMesh(verts, tris)
calls = []


def oracle_cast(ray_starts, ray_directions, mesh=None, max_dist=1e6, return_distance=False, return_normal=False, return_face_id=False):
    shape = ray_starts.shape
    hits, t, face = ORC.raycast_c(ray_starts.reshape(-1, 3), ray_directions.reshape(-1, 3), cmesh, max_dist)
    inf = float("inf")
    normal = None
    if return_normal:
        v = torch.from_numpy(verts)[torch.from_numpy(tris).long()[face.clamp(min=0).long()]]
        e1, e2 = v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]
        nrm = torch.cross(e1, e2, dim=-1)
        nrm = nrm / nrm.norm(dim=-1, keepdim=True)
        normal = torch.where((face >= 0)[:, None], nrm, torch.full_like(nrm, inf)).view(shape)
    calls.append((ray_starts.clone(), ray_directions.clone(), max_dist))
    return hits.view(shape), (t.view(shape[:-1]) if return_distance else None), normal, None


R.placeholder("isaaclab.utils.warp", raycast_mesh=oracle_cast, convert_to_warp_mesh=lambda *a, **k: None)
R.placeholder("isaaclab.sensors")
R.load("isaaclab.sensors.sensor_base", f"{R.L}/sensors/sensor_base.py")
R.placeholder("isaaclab.sensors.camera")
cd = R.load("isaaclab.sensors.camera.camera_data", f"{R.L}/sensors/camera/camera_data.py")
sys.modules["isaaclab.sensors.camera"].CameraData = cd.CameraData
R.placeholder("isaaclab.sensors.ray_caster")
R.placeholder("isaaclab.sensors.ray_caster.patterns")
pat = R.load("isaaclab.sensors.ray_caster.patterns.patterns", f"{R.L}/sensors/ray_caster/patterns/patterns.py")
R.load("isaaclab.sensors.ray_caster.ray_caster_data", f"{R.L}/sensors/ray_caster/ray_caster_data.py")
R.load("isaaclab.sensors.ray_caster.ray_caster", f"{R.L}/sensors/ray_caster/ray_caster.py")
rcc = R.load("isaaclab.sensors.ray_caster.ray_caster_camera", f"{R.L}/sensors/ray_caster/ray_caster_camera.py")
M = m["math"]

n, H, W = 10, 12, 16
pos = torch.cat([torch.rand(n, 2, generator=g) * 2.0 - 1.0, 1.0 + torch.rand(n, 1, generator=g)], -1)
q = M.normalize(torch.randn(n, 4, generator=g))
q[0] = torch.tensor([1.0, 0.0, 0.0, 0.0])
root_tf = torch.cat([pos, q[:, [1, 2, 3, 0]]], dim=-1)


class AView(physx.ArticulationView):
    count = n

    def get_root_transforms(self):
        return root_tf.clone()


out = dict(verts=torch.from_numpy(verts), tris=torch.from_numpy(tris), view_pos=pos, view_quat=q)
CASES = {
    # (offset convention, offset rot, clipping, data types)
    "ros_max": ("ros", (0.5, -0.5, 0.5, -0.5), "max", ["distance_to_image_plane", "distance_to_camera", "normals"]),
    "gl_zero": ("opengl", (0.9238795, 0.3826834, 0.0, 0.0), "zero", ["distance_to_image_plane", "distance_to_camera"]),
    "world_none": ("world", (0.9659258, 0.0, 0.2588190, 0.0), "none", ["distance_to_camera", "distance_to_image_plane"]),
}
for tag, (conv, rot, clip, types_) in CASES.items():
    s = object.__new__(rcc.RayCasterCamera)
    s._initialize_handle = s._invalidate_initialize_handle = s._debug_vis_handle = None
    pcfg = NS(focal_length=24.0, horizontal_aperture=20.955, vertical_aperture=None, horizontal_aperture_offset=0.0,
              vertical_aperture_offset=0.0, width=W, height=H, func=pat.pinhole_camera_pattern)
    s.cfg = NS(pattern_cfg=pcfg, max_distance=2.5, depth_clipping_behavior=clip, data_types=types_,
               mesh_prim_paths=["/World/ground"], offset=NS(pos=(0.1, -0.05, 0.2), rot=rot, convention=conv))
    s._device = "cpu"
    s._view = AView()
    s._data = cd.CameraData()
    s.meshes = {"/World/ground": None}
    s._initialize_rays_impl()
    out[f"{tag}_offset_quat"], out[f"{tag}_offset_pos"] = s._offset_quat.clone(), s._offset_pos.clone()
    out[f"{tag}_intrinsics"] = s._data.intrinsic_matrices.clone()
    out[f"{tag}_ray_dirs_local"] = s.ray_directions.clone()
    # two refreshes: all cameras, then a subset (frame counters differ afterwards)
    s._update_buffers_impl(torch.arange(n))
    sub = torch.tensor([1, 4, 5, 8])
    s._update_buffers_impl(sub)
    out[f"{tag}_subset"] = sub
    out[f"{tag}_frame"] = s._frame.clone()
    out[f"{tag}_pos_w"], out[f"{tag}_quat_w_world"] = s._data.pos_w.clone(), s._data.quat_w_world.clone()
    out[f"{tag}_quat_w_ros"], out[f"{tag}_quat_w_opengl"] = s._data.quat_w_ros.clone(), s._data.quat_w_opengl.clone()
    out[f"{tag}_ray_starts_w"], out[f"{tag}_ray_dirs_w"] = calls[-2][0], calls[-2][1]
    for name in types_:
        out[f"{tag}_{name}"] = s._data.output[name].clone()
    if tag == "ros_max":
        # re-pose: explicit world poses in the ros convention for a subset, look-at for another, new intrinsics for a third
        ids = torch.tensor([0, 2, 3])
        new_pos = torch.tensor([[0.0, 0.0, 2.0], [0.5, -0.5, 1.5], [-0.8, 0.3, 1.2]])
        new_q = M.normalize(torch.randn(3, 4, generator=g))
        s.set_world_poses(new_pos, new_q, ids, convention="ros")
        ids2 = torch.tensor([6, 7])
        eyes, targets = torch.tensor([[1.0, 1.0, 2.0], [-1.5, 0.0, 1.0]]), torch.tensor([[0.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
        s.set_world_poses_from_view(eyes, targets, ids2)
        ids3 = torch.tensor([8, 9])
        k = torch.tensor([[[18.0, 0.0, 8.5], [0.0, 17.0, 5.5], [0.0, 0.0, 1.0]], [[9.0, 0.0, 8.0], [0.0, 9.0, 6.0], [0.0, 0.0, 1.0]]])
        s.set_intrinsic_matrices(k, focal_length=1.0, env_ids=ids3)
        out["repose_ids"], out["repose_pos"], out["repose_quat_ros"] = ids, new_pos, new_q
        out["lookat_ids"], out["lookat_eyes"], out["lookat_targets"] = ids2, eyes, targets
        out["intr_ids"], out["intr_k"] = ids3, k
        out["repose_offset_pos"], out["repose_offset_quat"] = s._offset_pos.clone(), s._offset_quat.clone()
        out["repose_data_pos_w"], out["repose_data_quat_w_world"] = s._data.pos_w.clone(), s._data.quat_w_world.clone()
        s._update_buffers_impl(torch.arange(n))
        out["repose_pos_w"], out["repose_quat_w_world"] = s._data.pos_w.clone(), s._data.quat_w_world.clone()
        out["repose_ray_starts_w"], out["repose_ray_dirs_w"] = calls[-1][0], calls[-1][1]
        for name in types_:
            out[f"repose_{name}"] = s._data.output[name].clone()
        out["repose_frame"] = s._frame.clone()
    img = s._data.output["distance_to_camera"]
    print(tag, "misses", int(torch.isinf(calls[-1][0]).any()), "depth range", float(img[torch.isfinite(img)].min()),
          float(img[torch.isfinite(img)].max()), "nan", int(torch.isnan(s._data.output["distance_to_image_plane"]).sum()),
          "inf", int(torch.isinf(img).sum()))
np.savez_compressed(os.path.join(HERE, "ray_caster_camera.npz"), **{k: v.numpy() for k, v in out.items()})
print("wrote ray_caster_camera.npz with", len(out), "arrays")
