"""CPU: the pose / camera-geometry oracle against the fixture recorded from the reference's math functions."""

from conftest import assert_close, assert_exact, load_golden
from oracle import mathx as O


def test_pose_matrices():
    g = load_golden("pose_math")
    pose = O.make_pose(g["pos"], g["rot"])
    assert_exact(pose, g["make_pose"], "make_pose")
    assert_close(O.pose_inv(pose), g["pose_inv"], atol=1e-6, what="pose_inv")
    assert_close(O.pose_mul(g["other"], pose), g["pose_mul"], atol=2e-6, what="pose_in_A_to_pose_in_B")
    assert_close(O.pose_mul(g["other"][:1], pose), g["pose_mul_single"], atol=2e-6, what="one frame against a batch")
    rel = O.pose_mul(O.pose_inv(g["other"][7][None]), pose)
    assert_close(O.pose_mul(g["other"][3][None], rel), g["transform_A_B"], atol=5e-6, what="transform_poses_from_frame_A_to_frame_B")
    assert_exact(O.copysign(-2.5, g["signs"]), g["copysign"], "copysign")


def test_camera_geometry_and_cylinder():
    g = load_golden("pose_math")
    depth, K = g["depth"], g["K"]
    assert_exact(O.orthogonalize_depth(depth, K), g["ortho"], "orthogonalize_perspective_depth")
    assert_close(O.unproject_depth(depth, K), g["unproject"], rtol=1e-5, atol=1e-5, what="unproject_depth")
    assert_close(O.unproject_depth(depth, K[1:2]), g["unproject_shared_K"], rtol=1e-5, atol=1e-5, what="unproject_depth, one K")
    assert_close(O.unproject_depth(O.orthogonalize_depth(depth, K), K), g["unproject_persp"], rtol=1e-5, atol=1e-5, what="perspective depth")
    assert_close(O.project_points(g["points"], K), g["project"], rtol=1e-6, atol=1e-5, what="project_points")
    assert_close(O.sample_cylinder(0.7, (-0.5, 1.5), g["cylinder_u"]), g["cylinder"], atol=1e-6, what="sample_cylinder")
