"""Golden vectors for OperationalSpaceControllerAction, produced by the REFERENCE class
(``isaaclab/envs/mdp/actions/task_space_actions.py:232-700`` on the reference ``OperationalSpaceController``) with a recorded
articulation: its own ``__init__``, ``process_actions`` and ``apply_actions`` run; the contact sensor and the task-frame
transformer it would create on the stage are recording stand-ins.  Writes tests/golden/osc_action.npz."""

import os
import sys
from types import SimpleNamespace as NS

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refload as R  # noqa: E402

m = R.bootstrap()
M, L = m["math"], R.L
n = 64
STATE = {}


class FakeContactSensor:
    is_initialized = True

    def __init__(self, cfg):
        self.data = NS(net_forces_w=None)

    def update(self, dt):
        self.data.net_forces_w = STATE["ee_force_w"].unsqueeze(1)

    def reset(self, env_ids=None):
        pass


class FakeFrameTransformer:
    is_initialized = True

    def __init__(self, cfg):
        self.data = NS()

    def update(self, dt):
        self.data.target_pos_w, self.data.target_quat_w = STATE["task_pos_w"].unsqueeze(1), STATE["task_quat_w"].unsqueeze(1)

    def reset(self, env_ids=None):
        pass


class FrameCfg(NS):
    FrameCfg = NS


R.placeholder("isaaclab.sensors", ContactSensor=FakeContactSensor, ContactSensorCfg=NS, FrameTransformer=FakeFrameTransformer, FrameTransformerCfg=FrameCfg)
R.load("isaaclab.controllers.differential_ik_cfg", f"{L}/controllers/differential_ik_cfg.py")
sys.modules["isaaclab.controllers"].DifferentialIKControllerCfg = sys.modules["isaaclab.controllers.differential_ik_cfg"].DifferentialIKControllerCfg
osc = R.load("isaaclab.controllers.operational_space", f"{L}/controllers/operational_space.py")
R.placeholder("isaaclab.assets.articulation")
R.load("isaaclab.managers.action_manager", f"{L}/managers/action_manager.py")
R.placeholder("isaaclab.envs.mdp.actions")
tsa = R.load("isaaclab.envs.mdp.actions.task_space_actions", f"{L}/envs/mdp/actions/task_space_actions.py")
tsa.OperationalSpaceControllerAction._first_RigidObject_child_path = lambda self: "/World/envs/env_.*/Robot/link_base"

BASE = dict(target_types=["pose_abs"], motion_control_axes_task=(1, 1, 1, 1, 1, 1), contact_wrench_control_axes_task=(0, 0, 0, 0, 0, 0),
            inertial_dynamics_decoupling=False, partial_inertial_dynamics_decoupling=False, gravity_compensation=False,
            impedance_mode="fixed", motion_stiffness_task=(100.0, 100.0, 100.0, 100.0, 100.0, 100.0),
            motion_damping_ratio_task=(1.0, 1.0, 1.0, 1.0, 1.0, 1.0), motion_stiffness_limits_task=(0, 1000),
            motion_damping_ratio_limits_task=(0, 100), contact_wrench_stiffness_task=None, nullspace_control="none",
            nullspace_stiffness=10.0, nullspace_damping_ratio=1.0)
ROT = tuple(M.normalize(torch.tensor([[0.9, 0.1, 0.3, -0.2]]))[0].tolist())
# name: (fixed base, joint regex, controller overrides, action cfg overrides)
CASES = {
    "abs_offset_nullspace": (True, ["joint.*"], dict(inertial_dynamics_decoupling=True, gravity_compensation=True, nullspace_control="position"),
                             dict(body_offset=NS(pos=(0.02, -0.03, 0.107), rot=ROT), nullspace_joint_pos_target="center", position_scale=0.8)),
    "rel_variable_kp": (True, ["joint.*"], dict(target_types=["pose_rel"], impedance_mode="variable_kp", motion_stiffness_limits_task=(10.0, 300.0),
                                                inertial_dynamics_decoupling=True, partial_inertial_dynamics_decoupling=True),
                        dict(position_scale=0.5, orientation_scale=0.3, stiffness_scale=100.0)),
    "hybrid_taskframe_floating": (False, ["joint[1-7]"], dict(target_types=["pose_abs", "wrench_abs"], impedance_mode="variable",
                                                               motion_control_axes_task=(1, 1, 0, 1, 1, 1), contact_wrench_control_axes_task=(0, 0, 1, 0, 0, 0),
                                                               contact_wrench_stiffness_task=(0.0, 0.0, 0.1, 0.0, 0.0, 0.0), gravity_compensation=True,
                                                               motion_damping_ratio_limits_task=(0.2, 1.5), motion_stiffness_limits_task=(10.0, 300.0)),
                                  dict(body_offset=NS(pos=(0.0, 0.0, 0.1), rot=(1.0, 0.0, 0.0, 0.0)), task_frame_rel_path="Table/frame", wrench_scale=2.0,
                                       stiffness_scale=120.0, damping_ratio_scale=1.2)),
}


class Asset:
    def __init__(self, fixed, g):
        self.is_fixed_base = fixed
        self.num_joints = 9                         # seven arm joints and two fingers
        self.joint_names = [f"joint{i}" for i in range(1, 8)] + ["finger_l", "finger_r"]
        self.body_names = ["link_base"] + [f"link{i}" for i in range(1, 8)] + ["hand"]
        self.cfg = NS(prim_path="/World/envs/env_.*/Robot")
        nb, nj = len(self.body_names), self.num_joints
        d = self.data = NS()
        d.soft_joint_pos_limits = torch.stack([-2.0 - torch.rand(n, nj, generator=g), 1.5 + torch.rand(n, nj, generator=g)], dim=-1)
        d.default_joint_pos = torch.randn(n, nj, generator=g) * 0.3
        self.fixed = dict(soft_joint_pos_limits=d.soft_joint_pos_limits, default_joint_pos=d.default_joint_pos)
        self.root_physx_view = NS(get_jacobians=lambda: STATE["jacobians"].clone(), get_generalized_mass_matrices=lambda: STATE["mass"].clone(),
                                  get_gravity_compensation_forces=lambda: STATE["gravity"].clone())
        self.efforts = None
        self.nb, self.nj = nb, nj

    def find_joints(self, names, preserve_order=False):
        return m["string"].resolve_matching_names(names, self.joint_names, preserve_order)

    def find_bodies(self, names, preserve_order=False):
        return m["string"].resolve_matching_names(names, self.body_names, preserve_order)

    def set_joint_effort_target(self, target, joint_ids=None):
        self.efforts = target.clone()

    def new_state(self, g):
        nb, nj, d = self.nb, self.nj, self.data
        root = torch.cat([torch.randn(n, 3, generator=g), M.normalize(torch.randn(n, 4, generator=g)), torch.randn(n, 6, generator=g) * 0.2], dim=1)
        body = torch.cat([root[:, None, :3] + torch.randn(n, nb, 3, generator=g) * 0.4, M.normalize(torch.randn(n, nb, 4, generator=g)),
                          torch.randn(n, nb, 6, generator=g) * 0.3], dim=2)
        d.root_pos_w, d.root_quat_w, d.root_vel_w = root[:, :3], root[:, 3:7], root[:, 7:]
        d.body_pos_w, d.body_quat_w, d.body_vel_w = body[..., :3], body[..., 3:7], body[..., 7:]
        d.joint_pos, d.joint_vel = torch.randn(n, nj, generator=g) * 0.5, torch.randn(n, nj, generator=g) * 0.3
        cols = nj if self.is_fixed_base else nj + 6
        jb = nb - 1 if self.is_fixed_base else nb
        # well-conditioned Jacobian block for the seven arm joints of every body
        U, _ = torch.linalg.qr(torch.randn(n, jb, 6, 6, generator=g))
        V, _ = torch.linalg.qr(torch.randn(n, jb, cols, cols, generator=g))
        S = 0.3 + torch.rand(n, jb, 6, generator=g)
        STATE["jacobians"] = U @ torch.diag_embed(S) @ V[..., :6].transpose(-1, -2)
        A = torch.randn(n, nj, nj, generator=g) * 0.3
        STATE["mass"] = A @ A.transpose(1, 2) + 2.0 * torch.eye(nj)
        STATE["gravity"] = torch.randn(n, nj, generator=g)
        STATE["ee_force_w"] = torch.randn(n, 3, generator=g) * 5.0
        STATE["task_pos_w"] = root[:, :3] + torch.randn(n, 3, generator=g) * 0.5
        STATE["task_quat_w"] = M.normalize(torch.randn(n, 4, generator=g))
        return dict(root_state_w=root, body_state_w=body, joint_pos=d.joint_pos, joint_vel=d.joint_vel, **{k: v.clone() for k, v in STATE.items()})


out = {}
for tag, (fixed, joints, ctrl_over, act_over) in CASES.items():
    g = torch.Generator().manual_seed(sum(map(ord, tag)))
    asset = Asset(fixed, g)
    for k, v in asset.fixed.items():
        out[f"{tag}_{k}"] = v
    env = NS(num_envs=n, device="cpu", scene={"robot": asset}, sim=NS(get_physics_dt=lambda: 0.01))
    ctrl = NS(class_type=osc.OperationalSpaceController, **{**BASE, **ctrl_over})
    cfg = NS(asset_name="robot", joint_names=joints, body_name="hand", body_offset=None, task_frame_rel_path=None, controller_cfg=ctrl,
             position_scale=1.0, orientation_scale=1.0, wrench_scale=1.0, stiffness_scale=1.0, damping_ratio_scale=1.0,
             nullspace_joint_pos_target="none", debug_vis=False, clip=None)
    for k, v in act_over.items():
        setattr(cfg, k, v)
    asset.new_state(g)                     # the constructor reads nothing of it, but the views must answer
    term = tsa.OperationalSpaceControllerAction(cfg, env)
    steps = []
    for step in range(2):
        rec = asset.new_state(g)
        raw = torch.randn(n, term.action_dim, generator=g)
        if term._pose_abs_idx is not None:   # an absolute target near the current pose, orientation normalised
            i = term._pose_abs_idx
            raw[:, i:i + 3] = torch.randn(n, 3, generator=g) * 0.3
            raw[:, i + 3:i + 7] = M.normalize(torch.randn(n, 4, generator=g))
        if term._stiffness_idx is not None:
            raw[:, term._stiffness_idx:term._stiffness_idx + 6] = torch.rand(n, 6, generator=g) * 4.0 - 0.5
        if term._damping_ratio_idx is not None:
            raw[:, term._damping_ratio_idx:term._damping_ratio_idx + 6] = torch.rand(n, 6, generator=g) * 2.0 - 0.2
        term.process_actions(raw)
        term.apply_actions()
        rec.update(raw=raw, processed=term.processed_actions.clone(), ee_pose_b=term._ee_pose_b.clone(), ee_vel_b=term._ee_vel_b.clone(),
                   ee_force_b=term._ee_force_b.clone(), jacobian_b=term._jacobian_b.clone(), mass_matrix=term._mass_matrix.clone(),
                   gravity_dof=term._gravity.clone(), joint_efforts=asset.efforts.clone())
        if term._task_frame_pose_b is not None:
            rec["task_frame_pose_b"] = term._task_frame_pose_b.clone()
        steps.append(rec)
    for key in steps[0]:
        out[f"{tag}_{key}"] = torch.stack([s[key] for s in steps])
    print(tag, "action_dim", term.action_dim, "dof", term._num_DoF, "efforts |max|", float(asset.efforts.abs().max()))
np.savez_compressed(os.path.join(HERE, "osc_action.npz"), **{k: v.numpy() for k, v in out.items()})
print("wrote osc_action.npz with", len(out), "arrays")
