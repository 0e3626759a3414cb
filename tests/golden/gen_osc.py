"""Golden vectors for the operational-space controller, produced by the REFERENCE class
(``isaaclab/controllers/operational_space.py``) on seeded, well-conditioned inputs (7-DoF arm: random Jacobians with
singular values >= 0.3, mass matrices A A^T + 2 I).  Writes tests/golden/osc.npz."""

import os
import sys
from types import SimpleNamespace as NS

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refload as R  # noqa: E402

m = R.bootstrap()
osc = R.load("isaaclab.controllers.operational_space", f"{R.L}/controllers/operational_space.py")
M = m["math"]

BASE = dict(target_types=["pose_abs"], motion_control_axes_task=(1, 1, 1, 1, 1, 1), contact_wrench_control_axes_task=(0, 0, 0, 0, 0, 0),
            inertial_dynamics_decoupling=False, partial_inertial_dynamics_decoupling=False, gravity_compensation=False,
            impedance_mode="fixed", motion_stiffness_task=(100.0, 100.0, 100.0, 100.0, 100.0, 100.0),
            motion_damping_ratio_task=(1.0, 1.0, 1.0, 1.0, 1.0, 1.0), motion_stiffness_limits_task=(0, 1000),
            motion_damping_ratio_limits_task=(0, 100), contact_wrench_stiffness_task=None, nullspace_control="none",
            nullspace_stiffness=10.0, nullspace_damping_ratio=1.0)
CASES = {
    # the reference's own test configurations (test_operational_space.py) in spirit: one per code path
    "abs_fixed": dict(),
    "rel_decoupled_gravity": dict(target_types=["pose_rel"], inertial_dynamics_decoupling=True, gravity_compensation=True,
                                  motion_stiffness_task=(150.0, 120.0, 90.0, 60.0, 50.0, 40.0),
                                  motion_damping_ratio_task=(1.0, 0.9, 0.8, 1.1, 1.2, 0.7)),
    "abs_partial_nullspace": dict(inertial_dynamics_decoupling=True, partial_inertial_dynamics_decoupling=True,
                                  nullspace_control="position", nullspace_stiffness=12.0, nullspace_damping_ratio=0.8),
    "abs_full_nullspace": dict(inertial_dynamics_decoupling=True, nullspace_control="position"),
    "variable_kp_taskframe": dict(impedance_mode="variable_kp", motion_stiffness_limits_task=(10.0, 300.0)),
    "variable_taskframe": dict(target_types=["pose_rel"], impedance_mode="variable", motion_damping_ratio_limits_task=(0.2, 1.5),
                               motion_stiffness_limits_task=(10.0, 300.0)),
    "hybrid_closed_loop": dict(target_types=["pose_abs", "wrench_abs"], motion_control_axes_task=(1, 1, 0, 1, 1, 1),
                               contact_wrench_control_axes_task=(0, 0, 1, 0, 0, 0), contact_wrench_stiffness_task=(0.0, 0.0, 0.1, 0.0, 0.0, 0.0),
                               gravity_compensation=True),
    "wrench_open_loop": dict(target_types=["wrench_abs"], motion_control_axes_task=(0, 0, 0, 0, 0, 0),
                             contact_wrench_control_axes_task=(1, 1, 1, 1, 1, 1)),
}

n, dof = 96, 7
g = torch.Generator().manual_seed(6620)


def rand_quat(k):
    return M.normalize(torch.randn(k, 4, generator=g))


def well_conditioned_jacobian():
    a = torch.randn(n, 6, dof, generator=g, dtype=torch.float64)
    u, s, vh = torch.linalg.svd(a, full_matrices=False)
    s = 0.3 + 1.2 * torch.rand(n, 6, generator=g, dtype=torch.float64)
    return (u @ torch.diag_embed(s) @ vh).float()


out = {}
for tag, over in CASES.items():
    cfg = NS(**{**BASE, **over})
    ctl = osc.OperationalSpaceController(cfg, n, "cpu")
    steps = []
    for k in range(2):
        ee_pose = torch.cat([torch.randn(n, 3, generator=g) * 0.4, rand_quat(n)], -1)
        ee_vel = torch.randn(n, 6, generator=g) * 0.3
        ee_force = torch.randn(n, 3, generator=g) * 5.0
        task_pose = torch.cat([torch.randn(n, 3, generator=g) * 0.2, rand_quat(n)], -1) if "taskframe" in tag or "hybrid" in tag else None
        jac = well_conditioned_jacobian()
        a = torch.randn(n, dof, dof, generator=g, dtype=torch.float64) * 0.4
        mass = (a @ a.mT + 2.0 * torch.eye(dof, dtype=torch.float64)).float()
        grav = torch.randn(n, dof, generator=g) * 3.0
        q, qd = torch.randn(n, dof, generator=g), torch.randn(n, dof, generator=g) * 0.5
        q_null = torch.randn(n, dof, generator=g) * 0.3 if k == 1 else None
        parts = []
        for t in cfg.target_types:
            if t == "pose_abs":
                parts.append(torch.cat([torch.randn(n, 3, generator=g) * 0.4, rand_quat(n)], -1))
            elif t == "pose_rel":
                d = torch.randn(n, 6, generator=g) * 0.1
                d[:4, 3:] = 0.0      # zero rotation deltas take apply_delta_pose's small-angle branch
                parts.append(d)
            else:
                parts.append(torch.randn(n, 6, generator=g) * 4.0)
        if cfg.impedance_mode in ("variable_kp", "variable"):
            parts.append(torch.rand(n, 6, generator=g) * 400.0 - 20.0)    # some outside the limits on both sides
        if cfg.impedance_mode == "variable":
            parts.append(torch.rand(n, 6, generator=g) * 2.0)
        command = torch.cat(parts, -1)
        ctl.set_command(command.clone(), current_ee_pose_b=ee_pose, current_task_frame_pose_b=task_pose)
        tau = ctl.compute(jac, current_ee_pose_b=ee_pose, current_ee_vel_b=ee_vel, current_ee_force_b=ee_force, mass_matrix=mass,
                          gravity=grav, current_joint_pos=q, current_joint_vel=qd, nullspace_joint_pos_target=q_null)
        rec = dict(command=command, ee_pose=ee_pose, ee_vel=ee_vel, ee_force=ee_force, jac=jac, mass=mass, grav=grav, q=q, qd=qd,
                   tau=tau.clone(), p_gains_b=ctl._motion_p_gains_b.clone(), d_gains_b=ctl._motion_d_gains_b.clone(),
                   p_gains_task=ctl._motion_p_gains_task.clone(), d_gains_task=ctl._motion_d_gains_task.clone(),
                   sel_motion_b=ctl._selection_matrix_motion_b.clone(), sel_force_b=ctl._selection_matrix_force_b.clone(),
                   os_mass=ctl._os_mass_matrix_b.clone())
        if task_pose is not None:
            rec["task_pose"] = task_pose
        if q_null is not None:
            rec["q_null"] = q_null
        if ctl.desired_ee_pose_b is not None:
            rec["desired_pose_b"], rec["desired_pose_task"] = ctl.desired_ee_pose_b.clone(), ctl.desired_ee_pose_task.clone()
        if ctl.desired_ee_wrench_b is not None:
            rec["desired_wrench_b"] = ctl.desired_ee_wrench_b.clone()
        if ctl._contact_wrench_p_gains_b is not None:
            rec["wrench_gains_b"] = ctl._contact_wrench_p_gains_b.clone()
        steps.append(rec)
    for k, rec in enumerate(steps):
        for key, v in rec.items():
            out[f"{tag}_{k}_{key}"] = v
    print(tag, "action_dim", ctl.action_dim, "|tau| mean", float(steps[-1]["tau"].abs().mean()))
np.savez_compressed(os.path.join(HERE, "osc.npz"), **{k: v.numpy() for k, v in out.items()})
print("wrote osc.npz with", len(out), "arrays")
