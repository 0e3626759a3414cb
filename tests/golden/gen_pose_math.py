"""Golden vectors for the homogeneous-pose, interpolation and camera-geometry utilities, produced by the REFERENCE functions
(``isaaclab/utils/math.py``).  Writes tests/golden/pose_math.npz."""

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refload as R  # noqa: E402

M = R.bootstrap()["math"]
out = {}
g = torch.Generator().manual_seed(9091)
n = 50
quat = M.normalize(torch.randn(n, 4, generator=g))
rot = M.matrix_from_quat(quat)
pos = torch.randn(n, 3, generator=g) * 2.0
out["pos"], out["rot"] = pos, rot
pose = M.make_pose(pos, rot)
out["make_pose"] = pose
out["pose_inv"] = M.pose_inv(pose)
other = M.make_pose(torch.randn(n, 3, generator=g), M.matrix_from_quat(M.normalize(torch.randn(n, 4, generator=g))))
out["other"] = other
out["pose_mul"] = M.pose_in_A_to_pose_in_B(pose_in_A=pose, pose_A_in_B=other)
out["pose_mul_single"] = M.pose_in_A_to_pose_in_B(pose_in_A=pose, pose_A_in_B=other[:1])
out["transform_A_B"] = M.transform_poses_from_frame_A_to_frame_B(pose, other[3], other[7])
signs = torch.randn(n, generator=g)
signs[::7] = 0.0
out["signs"], out["copysign"] = signs, M.copysign(-2.5, signs)

# camera geometry
h, w = 6, 9
depth = torch.rand(4, h, w, generator=g) * 5.0 + 0.2
K = torch.tensor([[[20.0, 0.0, 4.3], [0.0, 18.0, 2.6], [0.0, 0.0, 1.0]]]).repeat(4, 1, 1)
K[1, 0, 0], K[2, 1, 2], K[3, 0, 1] = 25.0, 3.1, 0.4      # different focal length / centre / a skew term
out["depth"], out["K"] = depth, K
out["ortho"] = M.orthogonalize_perspective_depth(depth, K)
out["ortho_single"] = M.orthogonalize_perspective_depth(depth[0], K[0])
out["ortho_last1"] = M.orthogonalize_perspective_depth(depth.unsqueeze(-1), K[0])
out["unproject"] = M.unproject_depth(depth, K)
out["unproject_single"] = M.unproject_depth(depth[0], K[0])
out["unproject_shared_K"] = M.unproject_depth(depth, K[1])
out["unproject_persp"] = M.unproject_depth(depth, K, is_ortho=False)
pts = torch.randn(4, 30, 3, generator=g) + torch.tensor([0.0, 0.0, 4.0])
out["points"] = pts
out["project"] = M.project_points(pts, K)
out["project_single"] = M.project_points(pts[0], K[0])
out["project_one_batch"] = M.project_points(pts[:1], K[:1])

# cylinder: replay the draws
torch.manual_seed(77)
u0 = torch.rand(40)
u1 = torch.zeros(40).uniform_(0.0, 1.0)
torch.manual_seed(77)
out["cylinder"] = M.sample_cylinder(0.7, (-0.5, 1.5), 40, "cpu")
out["cylinder_u"] = torch.stack([u0, u1])

# interpolation helpers (single poses)
q1, q2 = quat[0].clone(), quat[1].clone()
out["slerp_q1"], out["slerp_q2"] = q1.clone(), q2.clone()
out["slerp"] = torch.stack([M.quat_slerp(q1.clone(), q2.clone(), t).clone() for t in (0.0, 0.25, 0.5, 0.9, 1.0)])
out["interp_rot_aa"] = M.interpolate_rotations(rot[2], rot[3], 5, axis_angle=True)
out["interp_rot_slerp"] = M.interpolate_rotations(rot[2], rot[3], 5, axis_angle=False)
small = M.matrix_from_quat(M.normalize(quat[2:3] + 0.005))[0]
out["interp_rot_small"] = M.interpolate_rotations(rot[2], small, 4, axis_angle=True)
out["interp_rot_small_R2"] = small
steps, count = M.interpolate_poses(pose[4], pose[5], num_steps=6)
out["interp_poses"], out["interp_poses_count"] = steps, torch.tensor(count)
steps, count = M.interpolate_poses(pose[4], pose[5], step_size=0.4)
out["interp_poses_by_size"], out["interp_poses_by_size_count"] = steps, torch.tensor(count)
np.savez_compressed(os.path.join(HERE, "pose_math.npz"), **{k: v.numpy() for k, v in out.items()})
print("wrote pose_math.npz with", len(out), "arrays")
