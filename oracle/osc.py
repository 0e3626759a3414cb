"""CPU restatement of the operational-space controller.  Test infrastructure only: the product never imports this module.

Follows ``isaaclab/controllers/operational_space.py``: ``set_command`` :173-343, ``compute`` :345-548.  State is a plain
dict; linear algebra runs in float64 (the reference runs it in float32, which is what bounds the agreement with the
fixture on these inputs to about 1e-4 relative).
"""

import torch

from . import mathx as M

_WIDTH = {"pose_rel": 6, "pose_abs": 7, "wrench_abs": 6}


def _diag_b(R, d):
    """R diag(d) R^T per 3x3 block (:299-327)."""
    out = torch.zeros(d.shape[0], 6, 6, dtype=torch.float64)
    for o in (0, 3):
        out[:, o:o + 3, o:o + 3] = R @ torch.diag_embed(d[:, o:o + 3]) @ R.mT
    return out


def init(cfg: dict, n: int) -> dict:
    """:34-146."""
    six = lambda v: torch.as_tensor(v, dtype=torch.float32).expand(6).clone() if torch.as_tensor(v).dim() == 0 else torch.as_tensor(v, dtype=torch.float32)  # noqa: E731
    axes_m, axes_f = six(cfg["motion_control_axes_task"]), six(cfg["contact_wrench_control_axes_task"])
    p = (axes_m * six(cfg["motion_stiffness_task"])).repeat(n, 1)
    s = dict(cfg=cfg, n=n, axes_m=axes_m, axes_f=axes_f, p_task=p, d_task=2 * p.sqrt() * six(cfg["motion_damping_ratio_task"]),
             pose_b=None, wrench_b=None, os_mass=torch.zeros(n, 6, 6, dtype=torch.float64))
    w = cfg.get("contact_wrench_stiffness_task")
    s["w_task"] = None if w is None else (axes_f * six(w)).repeat(n, 1)
    s["null_p"] = torch.tensor(cfg["nullspace_stiffness"], dtype=torch.float32)
    s["null_d"] = 2 * torch.sqrt(s["null_p"]) * torch.tensor(cfg["nullspace_damping_ratio"], dtype=torch.float32)
    return s


def set_command(s: dict, command, ee_pose=None, task_pose=None):
    """:173-343."""
    cfg, n = s["cfg"], s["n"]
    target_dim = sum(_WIDTH[t] for t in cfg["target_types"])
    mode = cfg["impedance_mode"]
    if mode != "fixed":
        lo, hi = cfg["motion_stiffness_limits_task"]
        s["p_task"] = s["axes_m"] * command[:, target_dim:target_dim + 6].clamp(lo, hi)
        ratio = torch.as_tensor(cfg["motion_damping_ratio_task"], dtype=torch.float32)
        if mode == "variable":
            lo, hi = cfg["motion_damping_ratio_limits_task"]
            ratio = command[:, target_dim + 6:target_dim + 12].clamp(lo, hi)
        s["d_task"] = 2 * s["p_task"].sqrt() * ratio
    if task_pose is None:
        task_pose = torch.tensor([[0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0]]).repeat(n, 1)
    tp, tq = task_pose[:, :3], task_pose[:, 3:]
    at = 0
    pose_task = wrench_task = None
    for t in cfg["target_types"]:
        target = command[:, at:at + _WIDTH[t]]
        at += _WIDTH[t]
        if t == "pose_rel":
            cp, cq = M.subtract_frame_transforms(tp, tq, ee_pose[:, :3], ee_pose[:, 3:])
            dp, dq = M.apply_delta_pose(cp, cq, target)
            pose_task = torch.cat([dp, dq], -1)
        elif t == "pose_abs":
            pose_task = target.clone()
        else:
            wrench_task = target.clone()
    R = M.matrix_from_quat(tq).double()
    s["p_b"], s["d_b"] = _diag_b(R, s["p_task"].double()), _diag_b(R, s["d_task"].double())
    s["w_b"] = None if s["w_task"] is None else _diag_b(R, s["w_task"].double())
    s["sel_m_b"], s["sel_f_b"] = _diag_b(R, s["axes_m"].double().repeat(n, 1)), _diag_b(R, s["axes_f"].double().repeat(n, 1))
    if pose_task is not None:
        bp, bq = M.combine_frame_transforms(tp, tq, pose_task[:, :3], pose_task[:, 3:])
        s["pose_task"], s["pose_b"] = pose_task, torch.cat([bp, bq], -1)
    if wrench_task is not None:
        f = (R @ wrench_task[:, :3].double().unsqueeze(-1)).squeeze(-1)
        mo = (R @ wrench_task[:, 3:].double().unsqueeze(-1)).squeeze(-1) + torch.cross(tp.double(), f, dim=-1)
        s["wrench_b"] = torch.cat([f, mo], -1)


def compute(s: dict, jac, ee_pose=None, ee_vel=None, ee_force=None, mass=None, gravity=None, q=None, qd=None, q_null=None):
    """:345-548."""
    cfg, n = s["cfg"], s["n"]
    J = jac.double()
    dof = J.shape[2]
    tau = torch.zeros(n, dof, dtype=torch.float64)
    minv = None
    if s["pose_b"] is not None:
        pe, re = M.compute_pose_error(ee_pose[:, :3], ee_pose[:, 3:], s["pose_b"][:, :3], s["pose_b"][:, 3:])
        err = torch.cat([pe, re], -1).double().unsqueeze(-1)
        acc = s["p_b"] @ err + s["d_b"] @ (-ee_vel.double()).unsqueeze(-1)
        if cfg["inertial_dynamics_decoupling"]:
            minv = torch.inverse(mass.double())
            if cfg["partial_inertial_dynamics_decoupling"]:
                s["os_mass"][:, 0:3, 0:3] = torch.inverse(J[:, 0:3] @ minv @ J[:, 0:3].mT)
                s["os_mass"][:, 3:6, 3:6] = torch.inverse(J[:, 3:6] @ minv @ J[:, 3:6].mT)
            else:
                s["os_mass"] = torch.inverse(J @ minv @ J.mT)
            force = s["os_mass"] @ acc
        else:
            force = acc
        tau += (J.mT @ s["sel_m_b"] @ force).squeeze(-1)
    if s["wrench_b"] is not None:
        w = s["wrench_b"]
        cmd = w.unsqueeze(-1)
        if s["w_b"] is not None:
            measured = torch.cat([ee_force.double(), w[:, 3:]], -1)
            cmd = cmd + s["w_b"] @ (w - measured).unsqueeze(-1)
        tau += (J.mT @ s["sel_f_b"] @ cmd).squeeze(-1)
    if cfg["gravity_compensation"]:
        tau += gravity.double()
    if cfg["nullspace_control"] == "position":
        if cfg["inertial_dynamics_decoupling"] and not cfg["partial_inertial_dynamics_decoupling"]:
            jpt = s["os_mass"] @ J @ minv
        else:
            jpt = torch.linalg.pinv(J).mT
        null_t = torch.eye(dof, dtype=torch.float64) - J.mT @ jpt
        target = torch.zeros_like(q) if q_null is None else q_null
        acc_n = (s["null_p"] * (target - q) + s["null_d"] * (-qd)).double().unsqueeze(-1)
        tau += (null_t @ mass.double() @ acc_n).squeeze(-1)
    return tau.float()
