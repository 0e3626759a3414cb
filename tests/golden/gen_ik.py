"""Golden vectors for the DifferentialIK controller and action term, produced by the REFERENCE classes
(``DifferentialIKController``, ``DifferentialInverseKinematicsAction``) on a fake fixed-base articulation.
Runs only where /root/reference is mounted; writes tests/golden/ik.npz."""

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

from isaaclab_amd import synth  # noqa: E402

m = R.bootstrap()
L = R.L
R.load("isaaclab.controllers.differential_ik_cfg", f"{L}/controllers/differential_ik_cfg.py")
sys.modules["isaaclab.controllers"].DifferentialIKControllerCfg = sys.modules["isaaclab.controllers.differential_ik_cfg"].DifferentialIKControllerCfg
R.placeholder("isaaclab.assets.articulation")
am = R.load("isaaclab.managers.action_manager", f"{L}/managers/action_manager.py")
R.placeholder("isaaclab.envs.mdp.actions")
tsa = R.load("isaaclab.envs.mdp.actions.task_space_actions", f"{L}/envs/mdp/actions/task_space_actions.py")
IKCfg = sys.modules["isaaclab.controllers.differential_ik_cfg"].DifferentialIKControllerCfg
IK = m["differential_ik"].DifferentialIKController

out = {}


def put(k, v):
    out[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)


class FakeView:
    def __init__(self, jac):
        self.jac = jac

    def get_jacobians(self):
        return self.jac.clone()


class FakeAsset:
    """Fixed-base arm: bodies = [base, link1..link6], joints = joint1..joint6."""
    is_fixed_base = True
    num_joints = 6
    joint_names = [f"joint{i}" for i in range(1, 7)]
    body_names = ["link_base"] + [f"link{i}" for i in range(1, 7)]

    def __init__(self, s):
        n = s["joint_pos"].shape[0]
        self.data = types.SimpleNamespace()
        self.data.joint_pos = s["joint_pos"]
        self.data.root_pos_w, self.data.root_quat_w = s["root_pos_w"], s["root_quat_w"]
        self.data.body_pos_w = torch.zeros(n, 7, 3)
        self.data.body_quat_w = torch.zeros(n, 7, 4)
        self.data.body_pos_w[:, 6], self.data.body_quat_w[:, 6] = s["ee_pos_w"], s["ee_quat_w"]
        jac = torch.zeros(n, 6, 6, 6)   # (n, bodies - 1, 6, joints)
        jac[:, 5] = s["jacobian_w"]
        self.root_physx_view = FakeView(jac)
        self.target = None

    def find_joints(self, names, preserve_order=False):
        return m["string"].resolve_matching_names(names, self.joint_names, preserve_order)

    def find_bodies(self, names, preserve_order=False):
        return m["string"].resolve_matching_names(names, self.body_names, preserve_order)

    def set_joint_position_target(self, target, joint_ids=None):
        self.target = target.clone()


def run_action_term(tag, s, command_type, relative, method, params, offset):
    n = s["joint_pos"].shape[0]
    asset = FakeAsset(s)
    env = types.SimpleNamespace(num_envs=n, device="cpu", scene={"robot": asset})
    cfg = types.SimpleNamespace(asset_name="robot", joint_names=["joint.*"], body_name="link6", scale=0.5, clip=None,
                                debug_vis=False, controller=IKCfg(command_type=command_type, use_relative_mode=relative,
                                                                  ik_method=method, ik_params=params),
                                body_offset=types.SimpleNamespace(pos=offset[0], rot=offset[1]) if offset else None)
    term = tsa.DifferentialInverseKinematicsAction(cfg, env)
    dim = term.action_dim
    g = torch.Generator().manual_seed(len(tag))
    if dim == 7:   # absolute pose: a reachable target near the current pose, in the root frame
        pos_b, quat_b = term._compute_frame_pose()
        delta = torch.cat([0.05 * torch.randn(n, 3, generator=g), 0.2 * torch.randn(n, 3, generator=g)], -1)
        tp, tq = m["math"].apply_delta_pose(pos_b, quat_b, delta)
        raw = torch.cat([tp, tq], -1) / 0.5
    elif dim == 6:
        raw = s["raw_action_pose_rel"].clone()
    else:
        raw = torch.rand(n, 3, generator=g) * 2 - 1
    term.process_actions(raw)
    term.apply_actions()
    put(f"{tag}_raw", raw)
    put(f"{tag}_processed", term.processed_actions)
    put(f"{tag}_ee_pos_des", term._ik_controller.ee_pos_des)
    put(f"{tag}_ee_quat_des", term._ik_controller.ee_quat_des)
    pos_b, quat_b = term._compute_frame_pose()
    put(f"{tag}_ee_pos_b", pos_b)
    put(f"{tag}_ee_quat_b", quat_b)
    put(f"{tag}_jacobian_b", term._compute_frame_jacobian())
    put(f"{tag}_joint_pos_des", asset.target)


n = 256
s = synth.reach_ik_state(n, 2001)
for k, v in s.items():
    put(f"state_{k}", v)
OFFSET = ((0.0, 0.0, 0.107), (1.0, 0.0, 0.0, 0.0))              # ik_rel_env_cfg.py:29-36 (Franka hand offset)
OFFSET_ROT = ((0.02, -0.03, 0.107), (0.9238795, 0.0, 0.3826834, 0.0))
run_action_term("rel_dls", s, "pose", True, "dls", None, OFFSET)
run_action_term("abs_dls", s, "pose", False, "dls", None, OFFSET)
run_action_term("rel_dls_rotoff", s, "pose", True, "dls", {"lambda_val": 0.05}, OFFSET_ROT)
run_action_term("rel_pinv", s, "pose", True, "pinv", None, None)
run_action_term("rel_svd", s, "pose", True, "svd", None, OFFSET)
run_action_term("rel_trans", s, "pose", True, "trans", {"k_val": 0.7}, OFFSET)
run_action_term("pos_rel_dls", s, "position", True, "dls", None, OFFSET)
run_action_term("pos_abs_pinv", s, "position", False, "pinv", None, None)

# controller alone on 6x7 (Franka-shaped) Jacobians with sigma_min >= 0.1, incl. the small-angle branch of apply_delta_pose
g = torch.Generator().manual_seed(7)
n7 = 256
U, _ = torch.linalg.qr(torch.randn(n7, 6, 6, generator=g))
V, _ = torch.linalg.qr(torch.randn(n7, 7, 7, generator=g))
S = 0.1 + 1.4 * torch.rand(n7, 6, generator=g)
jac7 = U @ torch.diag_embed(S) @ V[:, :, :6].transpose(1, 2)
q7 = torch.randn(n7, 7, generator=g)
ee_pos = torch.randn(n7, 3, generator=g) * 0.3
ee_quat = m["math"].normalize(torch.randn(n7, 4, generator=g))
cmd = torch.cat([0.1 * torch.randn(n7, 3, generator=g), 0.3 * torch.randn(n7, 3, generator=g)], -1)
cmd[:8, 3:] = 0.0              # zero rotation: NaN axis masked by the angle test (math.py:872-910)
cmd[8:16, 3:] *= 1e-6
for k, v in (("jac", jac7), ("q", q7), ("ee_pos", ee_pos), ("ee_quat", ee_quat), ("cmd", cmd)):
    put(f"franka_{k}", v)
for method in ("dls", "pinv", "svd", "trans"):
    ctrl = IK(IKCfg(command_type="pose", use_relative_mode=True, ik_method=method), n7, "cpu")
    ctrl.set_command(cmd, ee_pos, ee_quat)
    put(f"franka_{method}_ee_pos_des", ctrl.ee_pos_des)
    put(f"franka_{method}_ee_quat_des", ctrl.ee_quat_des)
    put(f"franka_{method}_q_des", ctrl.compute(ee_pos, ee_quat, jac7, q7))

np.savez_compressed(os.path.join(HERE, "ik.npz"), **out)
print("wrote ik.npz with", len(out), "arrays")
