"""GPU: homogeneous-pose, interpolation and camera-geometry utilities of ``isaaclab.utils.math`` (HIP ``ilx_pose_op`` /
``ilx_camera_op`` / ``ilx_sample_cylinder``) against the fixture recorded from the reference functions."""

import pytest
import torch

from conftest import assert_close, assert_exact, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def test_pose_matrices(dev):
    from isaaclab_amd.utils import math as M

    g = {k: v.to(dev) for k, v in load_golden("pose_math").items()}
    pose = M.make_pose(g["pos"], g["rot"])
    assert_exact(pose, g["make_pose"], "make_pose")
    pos, rot = M.unmake_pose(pose)
    assert_exact(pos, g["pos"], "unmake_pose position")
    assert_exact(rot, g["rot"], "unmake_pose rotation")
    assert_close(M.pose_inv(pose), g["pose_inv"], atol=1e-6, what="pose_inv")
    assert_close(M.pose_in_A_to_pose_in_B(pose_in_A=pose, pose_A_in_B=g["other"]), g["pose_mul"], atol=2e-6, what="pose_in_A_to_pose_in_B")
    assert_close(M.pose_in_A_to_pose_in_B(pose_in_A=pose, pose_A_in_B=g["other"][:1]), g["pose_mul_single"], atol=2e-6,
                 what="one frame against a batch")
    assert_close(M.transform_poses_from_frame_A_to_frame_B(pose, g["other"][3], g["other"][7]), g["transform_A_B"], atol=5e-6,
                 what="transform_poses_from_frame_A_to_frame_B")
    assert_exact(M.copysign(-2.5, g["signs"]), g["copysign"], "copysign")
    # single matrices and higher-rank batches keep their shape
    assert M.pose_inv(pose[0]).shape == (4, 4) and M.make_pose(g["pos"].view(5, 10, 3), g["rot"].view(5, 10, 3, 3)).shape == (5, 10, 4, 4)
    ident_p, ident_q = torch.zeros(7, 3, device=dev), torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=dev).repeat(7, 1)
    assert M.is_identity_pose(ident_p, ident_q) is True
    ident_q[3, 0] = 1.0 + 5e-6
    assert M.is_identity_pose(ident_p, ident_q) is True           # inside allclose's relative tolerance
    ident_p[5, 1] = 1e-6
    assert M.is_identity_pose(ident_p, ident_q) is False


def test_camera_geometry_and_cylinder(dev):
    from isaaclab_amd.utils import math as M

    g = {k: v.to(dev) for k, v in load_golden("pose_math").items()}
    depth, K = g["depth"], g["K"]
    # the kernel's correctly rounded op sequence (the same as numpy's) differs from torch's CPU result in the last bit of a few pixels
    ulp = dict(rtol=2.5e-7, atol=0.0)   # one unit in the last place
    assert_close(M.orthogonalize_perspective_depth(depth, K), g["ortho"], what="orthogonalize_perspective_depth", **ulp)
    assert_close(M.orthogonalize_perspective_depth(depth[0], K[0]), g["ortho_single"], what="one image", **ulp)
    assert_close(M.orthogonalize_perspective_depth(depth.unsqueeze(-1), K[0]), g["ortho_last1"], what="(N, H, W, 1) with one K", **ulp)
    # the reference inverts K by LU in float; here by cofactors in double: 1e-5 of the point's scale
    assert_close(M.unproject_depth(depth, K), g["unproject"], rtol=1e-5, atol=1e-5, what="unproject_depth")
    assert_close(M.unproject_depth(depth[0], K[0]), g["unproject_single"], rtol=1e-5, atol=1e-5, what="unproject_depth, one image")
    assert_close(M.unproject_depth(depth, K[1]), g["unproject_shared_K"], rtol=1e-5, atol=1e-5, what="unproject_depth, one K")
    assert_close(M.unproject_depth(depth, K, is_ortho=False), g["unproject_persp"], rtol=1e-5, atol=1e-5, what="perspective depth")
    assert_close(M.project_points(g["points"], K), g["project"], rtol=1e-6, atol=1e-5, what="project_points")
    assert_close(M.project_points(g["points"][0], K[0]), g["project_single"], rtol=1e-6, atol=1e-5, what="project_points (P, 3)")
    assert_close(M.project_points(g["points"][:1], K[:1]), g["project_one_batch"], rtol=1e-6, atol=1e-5, what="project_points (1, P, 3)")
    assert_close(M.sample_cylinder(0.7, (-0.5, 1.5), 40, dev, _u=g["cylinder_u"]), g["cylinder"], atol=1e-6, what="sample_cylinder")
    pts = M.sample_cylinder(2.0, (0.0, 1.0), (64, 8), dev)
    assert pts.shape == (64, 8, 3)
    assert_close(pts[..., :2].norm(dim=-1), torch.full((64, 8), 2.0), atol=1e-5, what="radius of generated points")
    assert float(pts[..., 2].min()) >= 0.0 and float(pts[..., 2].max()) <= 1.0
    with pytest.raises(ValueError, match="Expected intrinsics"):
        M.unproject_depth(depth, K[0, 0])


def test_interpolation_helpers(dev):
    from isaaclab_amd.utils import math as M

    g = {k: v.to(dev) for k, v in load_golden("pose_math").items()}
    got = torch.stack([M.quat_slerp(g["slerp_q1"].clone(), g["slerp_q2"].clone(), t) for t in (0.0, 0.25, 0.5, 0.9, 1.0)])
    assert_close(got, g["slerp"], atol=1e-6, what="quat_slerp")
    rot = g["rot"]
    assert_close(M.interpolate_rotations(rot[2], rot[3], 5, axis_angle=True), g["interp_rot_aa"], atol=2e-6, what="axis-angle steps")
    assert_close(M.interpolate_rotations(rot[2], rot[3], 5, axis_angle=False), g["interp_rot_slerp"], atol=2e-6, what="slerp steps")
    assert_close(M.interpolate_rotations(rot[2], g["interp_rot_small_R2"], 4, axis_angle=True), g["interp_rot_small"], atol=2e-6,
                 what="small-angle steps")
    pose = g["make_pose"]
    steps, count = M.interpolate_poses(pose[4], pose[5], num_steps=6)
    assert count == int(g["interp_poses_count"])
    assert_close(steps, g["interp_poses"], atol=2e-6, what="interpolate_poses")
    steps, count = M.interpolate_poses(pose[4], pose[5], step_size=0.4)
    assert count == int(g["interp_poses_by_size_count"])
    assert_close(steps, g["interp_poses_by_size"], atol=2e-6, what="interpolate_poses by step size")
    T = M.generate_random_transformation_matrix(pos_boundary=2.0, device=dev)
    assert T.shape == (4, 4) and float(T[3, 3]) == 1.0 and float(T[:3, 3].abs().max()) <= 2.0
    assert_close(T[:3, :3] @ T[:3, :3].T, torch.eye(3), atol=1e-5, what="random rotation is orthonormal")
