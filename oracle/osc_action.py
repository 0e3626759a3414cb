"""CPU restatement of the state assembly of OperationalSpaceControllerAction (test infrastructure only; never imported by the
product).  Follows ``isaaclab/envs/mdp/actions/task_space_actions.py:400-410, 568-662``.  Pinned by
tests/golden/osc_action.npz, recorded from the reference class.
"""

from __future__ import annotations

import torch

from . import mathx as M


def state(root_state_w, body_state_w, ee_body_idx, jacobians, jac_body_idx, jac_joint_ids, offset=None, ee_force_w=None):
    """Returns (ee_pose_b (n,7), ee_vel_b (n,6), ee_force_b (n,3) | None, jacobian_b (n,6,dof))."""
    n = root_state_w.shape[0]
    root_p, root_q, root_v = root_state_w[:, :3], root_state_w[:, 3:7], root_state_w[:, 7:]
    ee = body_state_w[:, ee_body_idx]
    pb, qb = M.subtract_frame_transforms(root_p, root_q, ee[:, :3], ee[:, 3:7])
    pose_p, pose_q = pb, qb
    if offset is not None:
        off_p, off_q = offset[0].expand(n, 3), offset[1].expand(n, 4)
        pose_p, pose_q = M.combine_frame_transforms(pb, qb, off_p, off_q)
    rel = ee[:, 7:] - root_v
    lin, ang = M.quat_rotate_inverse(root_q, rel[:, :3]), M.quat_rotate_inverse(root_q, rel[:, 3:])
    if offset is not None:
        lin = lin + torch.cross(ang, M.quat_rotate(qb, off_p), dim=-1)
    force = None if ee_force_w is None else M.quat_rotate_inverse(root_q, ee_force_w)
    jac = jacobians[:, jac_body_idx][:, :, jac_joint_ids].clone()
    R = M.matrix_from_quat(M.quat_inv(root_q))
    jac[:, :3], jac[:, 3:] = torch.bmm(R, jac[:, :3]), torch.bmm(R, jac[:, 3:])
    if offset is not None:
        jac[:, :3] += torch.bmm(-M.skew_symmetric_matrix(off_p), jac[:, 3:])
        jac[:, 3:] = torch.bmm(M.matrix_from_quat(off_q), jac[:, 3:])
    return torch.cat([pose_p, pose_q], dim=1), torch.cat([lin, ang], dim=1), force, jac
