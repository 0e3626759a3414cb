"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's differential inverse kinematics.

Controller: ``isaaclab/controllers/differential_ik.py`` (set_command :97-147, compute :148-174,
_compute_delta_joint_pos :180-240).  Action term: ``isaaclab/envs/mdp/actions/task_space_actions.py``
(process_actions :155-167, apply_actions :169-181, _compute_frame_pose :190-209, jacobian_b :140-149,
_compute_frame_jacobian :211-229).  Plain torch CPU ops in the reference's order; the small solves call the same
``torch.linalg`` routines the reference calls.  Pinned by ``tests/golden/ik.npz`` (outputs of the reference classes).
"""

from __future__ import annotations

import torch

from . import mathx as M


def set_command(command, ee_pos, ee_quat, command_type: str, use_relative_mode: bool):
    """differential_ik.py:97-147 -> (ee_pos_des, ee_quat_des)."""
    if command_type == "position":
        if use_relative_mode:
            return ee_pos + command, ee_quat.clone()
        return command.clone(), ee_quat.clone()
    if use_relative_mode:
        return M.apply_delta_pose(ee_pos, ee_quat, command)
    return command[:, 0:3].clone(), command[:, 3:7].clone()


def delta_joint_pos(delta_pose, jacobian, method: str, params: dict):
    """differential_ik.py:180-240."""
    if method == "pinv":
        return (params["k_val"] * torch.linalg.pinv(jacobian) @ delta_pose.unsqueeze(-1)).squeeze(-1)
    if method == "svd":
        U, S, Vh = torch.linalg.svd(jacobian)
        S_inv = torch.where(S > params["min_singular_value"], 1.0 / S, torch.zeros_like(S))
        pinv = Vh.transpose(1, 2)[:, :, :S.shape[1]] @ torch.diag_embed(S_inv) @ U.transpose(1, 2)
        return (params["k_val"] * pinv @ delta_pose.unsqueeze(-1)).squeeze(-1)
    if method == "trans":
        return (params["k_val"] * jacobian.transpose(1, 2) @ delta_pose.unsqueeze(-1)).squeeze(-1)
    if method == "dls":
        jt = jacobian.transpose(1, 2)
        lam = (params["lambda_val"] ** 2) * torch.eye(jacobian.shape[1])
        return (jt @ torch.inverse(jacobian @ jt + lam) @ delta_pose.unsqueeze(-1)).squeeze(-1)
    raise ValueError(method)


def compute(ee_pos, ee_quat, ee_pos_des, ee_quat_des, jacobian, joint_pos, command_type: str, method: str, params: dict):
    """differential_ik.py:148-174."""
    if "position" in command_type:
        return joint_pos + delta_joint_pos(ee_pos_des - ee_pos, jacobian[:, 0:3], method, params)
    pos_err, aa_err = M.compute_pose_error(ee_pos, ee_quat, ee_pos_des, ee_quat_des)
    return joint_pos + delta_joint_pos(torch.cat((pos_err, aa_err), dim=1), jacobian, method, params)


def compute_f64(ee_pos, ee_quat, ee_pos_des, ee_quat_des, jacobian, joint_pos, command_type: str, method: str, params: dict):
    """The same update with the small solve carried out in float64 (pose error stays float32 like the reference):
    the value both fp32 implementations approximate; used to put the fp32 deviations in perspective."""
    if "position" in command_type:
        err, jac = (ee_pos_des - ee_pos), jacobian[:, 0:3]
    else:
        pos_err, aa_err = M.compute_pose_error(ee_pos, ee_quat, ee_pos_des, ee_quat_des)
        err, jac = torch.cat((pos_err, aa_err), dim=1), jacobian
    p = dict(params)
    if method == "pinv":  # torch.linalg.pinv's default cut-off for float32 input
        U, S, Vh = torch.linalg.svd(jac.double(), full_matrices=False)
        rtol = max(jac.shape[1], jac.shape[2]) * torch.finfo(torch.float32).eps
        S_inv = torch.where(S > rtol * S[:, :1], 1.0 / S, torch.zeros_like(S))
        dq = p["k_val"] * (Vh.transpose(1, 2) @ torch.diag_embed(S_inv) @ U.transpose(1, 2) @ err.double().unsqueeze(-1)).squeeze(-1)
    elif method == "svd":
        U, S, Vh = torch.linalg.svd(jac.double(), full_matrices=False)
        S_inv = torch.where(S > p["min_singular_value"], 1.0 / S, torch.zeros_like(S))
        dq = p["k_val"] * (Vh.transpose(1, 2) @ torch.diag_embed(S_inv) @ U.transpose(1, 2) @ err.double().unsqueeze(-1)).squeeze(-1)
    else:
        dq = delta_joint_pos(err.double(), jac.double(), method, p)
    return (joint_pos.double() + dq).float()


def frame_pose(root_pos_w, root_quat_w, ee_pos_w, ee_quat_w, offset_pos=None, offset_rot=None):
    """task_space_actions.py:190-209."""
    pos_b, quat_b = M.subtract_frame_transforms(root_pos_w, root_quat_w, ee_pos_w, ee_quat_w)
    if offset_pos is not None:
        n = root_pos_w.shape[0]
        pos_b, quat_b = M.combine_frame_transforms(pos_b, quat_b, offset_pos.expand(n, 3), offset_rot.expand(n, 4))
    return pos_b, quat_b


def frame_jacobian(jacobian_w, root_quat_w, offset_pos=None, offset_rot=None):
    """task_space_actions.py:140-149 (rotate into the root frame) + :211-229 (frame offset)."""
    jac = jacobian_w.clone()
    rot = M.matrix_from_quat(M.quat_inv(root_quat_w))
    jac[:, :3, :] = torch.bmm(rot, jac[:, :3, :])
    jac[:, 3:, :] = torch.bmm(rot, jac[:, 3:, :])
    if offset_pos is not None:
        n = jac.shape[0]
        jac[:, 0:3, :] += torch.bmm(-M.skew_symmetric_matrix(offset_pos.expand(n, 3)), jac[:, 3:, :])
        jac[:, 3:, :] = torch.bmm(M.matrix_from_quat(offset_rot.expand(n, 4)), jac[:, 3:, :])
    return jac


def process_actions(raw, scale, clip, root_pos_w, root_quat_w, ee_pos_w, ee_quat_w, offset_pos, offset_rot,
                    command_type: str, use_relative_mode: bool):
    """task_space_actions.py:155-167 -> (processed, ee_pos_des, ee_quat_des)."""
    processed = raw * scale
    if clip is not None:
        processed = torch.clamp(processed, min=clip[..., 0], max=clip[..., 1])
    pos_b, quat_b = frame_pose(root_pos_w, root_quat_w, ee_pos_w, ee_quat_w, offset_pos, offset_rot)
    des = set_command(processed, pos_b, quat_b, command_type, use_relative_mode)
    return processed, des[0], des[1]


def apply_actions(root_pos_w, root_quat_w, ee_pos_w, ee_quat_w, jacobian_w, joint_pos, ee_pos_des, ee_quat_des, offset_pos,
                  offset_rot, command_type: str, method: str, params: dict, f64: bool = False):
    """task_space_actions.py:169-181 -> joint_pos_des."""
    pos_b, quat_b = frame_pose(root_pos_w, root_quat_w, ee_pos_w, ee_quat_w, offset_pos, offset_rot)
    if quat_b.norm() == 0:
        return joint_pos.clone()
    jac = frame_jacobian(jacobian_w, root_quat_w, offset_pos, offset_rot)
    fn = compute_f64 if f64 else compute
    return fn(pos_b, quat_b, ee_pos_des, ee_quat_des, jac, joint_pos, command_type, method, params)
