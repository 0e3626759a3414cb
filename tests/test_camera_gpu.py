"""GPU: RayCasterCamera (ilx_raycaster_camera_refresh through the C ABI) against the fixture recorded from the reference
class.  The reference's Warp cast was replaced by this build's C oracle when the fixture was made, so poses, offsets,
conventions, world rays, clipping rules and frame counters are pinned to the reference, the hit itself to the oracle."""

import pytest
import torch

from conftest import assert_close, assert_exact, load_golden

pytestmark = pytest.mark.gpu

CASES = {
    "ros_max": ("ros", (0.5, -0.5, 0.5, -0.5), "max", ["distance_to_image_plane", "distance_to_camera", "normals"]),
    "gl_zero": ("opengl", (0.9238795, 0.3826834, 0.0, 0.0), "zero", ["distance_to_image_plane", "distance_to_camera"]),
    "world_none": ("world", (0.9659258, 0.0, 0.2588190, 0.0), "none", ["distance_to_camera", "distance_to_image_plane"]),
}


def _camera(g, dev, tag):
    from isaaclab_amd.sensors.patterns import PinholeCameraPatternCfg
    from isaaclab_amd.sensors.ray_caster_camera import CameraOffsetCfg, RayCasterCamera, RayCasterCameraCfg
    from isaaclab_amd.utils.mesh import convert_to_mesh

    conv, rot, clip, types_ = CASES[tag]
    mesh = convert_to_mesh(g["verts"].numpy(), g["tris"].numpy(), str(dev))
    cfg = RayCasterCameraCfg(offset=CameraOffsetCfg(pos=(0.1, -0.05, 0.2), rot=rot, convention=conv), data_types=types_,
                             depth_clipping_behavior=clip, max_distance=2.5,
                             pattern_cfg=PinholeCameraPatternCfg(focal_length=24.0, horizontal_aperture=20.955, width=16, height=12))
    cam = RayCasterCamera(cfg, g["view_pos"].shape[0], mesh, dev)
    cam.set_view_pose(g["view_pos"].to(dev), g["view_quat"].to(dev))
    return cam


def _check_images(cam, g, prefix, types_, what):
    for name in types_:
        got, want = cam._data.output[name].cpu(), g[f"{prefix}_{name}"]
        assert got.shape == want.shape
        # misses: the same pixels are inf / nan; everything else to 1e-5 (depths are O(1) m)
        assert torch.equal(torch.isnan(got), torch.isnan(want)), f"{what} {name}: nan pattern"
        assert torch.equal(torch.isinf(got), torch.isinf(want)), f"{what} {name}: inf pattern"
        ok = torch.isfinite(want)
        assert_close(got[ok], want[ok], rtol=1e-5, atol=2e-6, what=f"{what} {name}")


@pytest.mark.parametrize("tag", list(CASES))
def test_camera_images_match_reference(dev, tag):
    g = load_golden("ray_caster_camera")
    cam = _camera(g, dev, tag)
    types_ = CASES[tag][3]
    assert cam.image_shape == (12, 16) and cam.num_rays == 192
    assert_close(cam._offset_quat, g[f"{tag}_offset_quat"], atol=1e-6, what="offset quaternion in the world convention")
    assert_exact(cam._offset_pos, g[f"{tag}_offset_pos"], "offset position")
    assert_close(cam._data.intrinsic_matrices, g[f"{tag}_intrinsics"], rtol=1e-6, what="intrinsic matrices")
    assert_close(cam.ray_directions, g[f"{tag}_ray_dirs_local"], atol=1e-6, what="pixel ray directions")
    data = cam.data                      # first read refreshes every camera
    cam._is_outdated[g[f"{tag}_subset"].to(dev)] = True
    data = cam.data                      # second refresh: the subset only
    assert not bool(cam._is_outdated.any())
    assert_exact(cam.frame, g[f"{tag}_frame"], "frame counters")
    assert_close(data.pos_w, g[f"{tag}_pos_w"], atol=1e-6, what="camera position")
    assert_close(data.quat_w_world, g[f"{tag}_quat_w_world"], atol=1e-6, what="camera orientation (world convention)")
    assert_close(data.quat_w_ros, g[f"{tag}_quat_w_ros"], atol=2e-6, what="camera orientation (ros convention)")
    assert_close(data.quat_w_opengl, g[f"{tag}_quat_w_opengl"], atol=2e-6, what="camera orientation (opengl convention)")
    _check_images(cam, g, tag, types_, tag)
    # the hit points are where the recorded world rays meet the recorded depth
    if tag == "ros_max":
        hits = cam.ray_hits_w.cpu()
        ok = torch.isfinite(hits).all(-1)
        depth_unclipped = (hits - g[f"{tag}_ray_starts_w"]).norm(dim=-1)
        clipped = g[f"{tag}_distance_to_camera"].reshape(hits.shape[:2])
        assert_close(torch.clamp(depth_unclipped[ok], max=2.5), clipped[ok], rtol=1e-5, atol=1e-5, what="hits vs depth image")


def test_camera_repose_and_intrinsics(dev):
    """set_world_poses (ros convention), set_world_poses_from_view and set_intrinsic_matrices, then a refresh."""
    g = load_golden("ray_caster_camera")
    cam = _camera(g, dev, "ros_max")
    cam.data
    cam._is_outdated[g["ros_max_subset"].to(dev)] = True   # the recorded sequence: all, a subset, re-pose, all
    cam.data
    cam.set_world_poses(g["repose_pos"].to(dev), g["repose_quat_ros"].to(dev), g["repose_ids"].to(dev), convention="ros")
    cam.set_world_poses_from_view(g["lookat_eyes"].to(dev), g["lookat_targets"].to(dev), g["lookat_ids"].to(dev))
    cam.set_intrinsic_matrices(g["intr_k"].to(dev), focal_length=1.0, env_ids=g["intr_ids"].to(dev))
    assert_close(cam._offset_pos, g["repose_offset_pos"], atol=2e-6, what="offsets after re-posing")
    q, want = cam._offset_quat.cpu(), g["repose_offset_quat"]
    sign = torch.sign((q * want).sum(-1, keepdim=True))      # q and -q are the same rotation
    assert_close(q * sign, want, atol=2e-6, what="offset quaternions after re-posing")
    assert_close(cam._data.pos_w, g["repose_data_pos_w"], atol=2e-6, what="pose data after re-posing")
    cam._is_outdated[:] = True
    data = cam.data
    assert_close(data.pos_w, g["repose_pos_w"], atol=2e-6, what="camera position after refresh")
    # the re-posed cameras sit where they were told to
    assert_close(data.pos_w.cpu()[g["repose_ids"]], g["repose_pos"], atol=2e-6, what="explicit positions")
    assert_close(data.pos_w.cpu()[g["lookat_ids"]], g["lookat_eyes"], atol=2e-6, what="look-at eyes")
    assert_exact(cam.frame, g["repose_frame"], "frame counters")
    _check_images(cam, g, "repose", CASES["ros_max"][3], "after re-posing")


def test_camera_rejects_unsupported_types(dev):
    from isaaclab_amd.sensors.ray_caster_camera import RayCasterCamera, RayCasterCameraCfg

    with pytest.raises(ValueError, match="does not support the following sensor types"):
        RayCasterCamera(RayCasterCameraCfg(data_types=["rgb", "distance_to_camera"]), 2, None, dev)
    with pytest.raises(ValueError, match="depth_clipping_behavior"):
        RayCasterCamera(RayCasterCameraCfg(depth_clipping_behavior="clamp"), 2, None, dev)
