"""Golden vectors for the limit-scaled, moving-average, binary and non-holonomic action terms, produced by the REFERENCE
classes (``isaaclab/envs/mdp/actions/{joint_actions_to_limits,binary_joint_actions,non_holonomic_actions}.py``) on a
stand-in articulation.  Writes tests/golden/actions2.npz."""

import os
import sys
from types import SimpleNamespace as NS

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refload as R  # noqa: E402

m = R.bootstrap()
am = R.load("isaaclab.managers.action_manager", f"{R.L}/managers/action_manager.py")
sys.modules["isaaclab.managers"].ActionTerm = am.ActionTerm
R.placeholder("isaaclab.envs.mdp.actions")
R.placeholder("isaaclab.envs.mdp.actions.actions_cfg")
tl = R.load("isaaclab.envs.mdp.actions.joint_actions_to_limits", f"{R.L}/envs/mdp/actions/joint_actions_to_limits.py")
bj = R.load("isaaclab.envs.mdp.actions.binary_joint_actions", f"{R.L}/envs/mdp/actions/binary_joint_actions.py")
nh = R.load("isaaclab.envs.mdp.actions.non_holonomic_actions", f"{R.L}/envs/mdp/actions/non_holonomic_actions.py")
ActionTermCfg = m["manager_term_cfg"].ActionTermCfg
S, M = m["string"], m["math"]

JOINTS = ["base_x", "base_y", "base_yaw", "arm_1", "arm_2", "arm_3", "finger_l", "finger_r"]
BODIES = ["world", "base", "arm", "hand"]
n = 80
g = torch.Generator().manual_seed(8202)


class Asset:
    def __init__(self):
        self.joint_names, self.num_joints, self.body_names = JOINTS, len(JOINTS), BODIES
        j = self.num_joints
        lo = -1.0 - torch.rand(n, j, generator=g)
        hi = 0.5 + torch.rand(n, j, generator=g)
        q = M.normalize(torch.randn(n, len(BODIES), 4, generator=g))
        self.data = NS(soft_joint_pos_limits=torch.stack([lo, hi], -1), joint_pos=torch.rand(n, j, generator=g) - 0.5,
                       body_quat_w=q, joint_pos_target=torch.zeros(n, j), joint_vel_target=torch.zeros(n, j))

    def find_joints(self, names, preserve_order=False):
        return S.resolve_matching_names(names, self.joint_names, preserve_order)

    def find_bodies(self, names, preserve_order=False):
        return S.resolve_matching_names(names, self.body_names, preserve_order)

    def set_joint_position_target(self, t, joint_ids=None):
        self.data.joint_pos_target[:, joint_ids] = t

    def set_joint_velocity_target(self, t, joint_ids=None):
        self.data.joint_vel_target[:, joint_ids] = t


def cfg_of(cls, **kw):
    cfg = ActionTermCfg(class_type=cls, asset_name="robot", debug_vis=False, clip=kw.pop("clip", None))
    for k, v in kw.items():
        object.__setattr__(cfg, k, v)
    return cfg


asset = Asset()
env = NS(num_envs=n, device="cpu", scene={"robot": asset})
out = dict(soft_joint_pos_limits=asset.data.soft_joint_pos_limits.clone(), joint_pos=asset.data.joint_pos.clone(),
           body_quat_w=asset.data.body_quat_w.clone())


def run(tag, term, steps=3, width=None, boolean=False):
    raws, procs = [], []
    for k in range(steps):
        raw = torch.rand(n, width or term.action_dim, generator=g) * 4 - 2
        if boolean and k == 1:
            raw = raw > 0
        term.process_actions(raw)
        term.apply_actions()
        raws.append(raw.float())
        procs.append(term.processed_actions.clone())
    out[f"{tag}_raw"], out[f"{tag}_processed"] = torch.stack(raws), torch.stack(procs)
    out[f"{tag}_pos_target"], out[f"{tag}_vel_target"] = asset.data.joint_pos_target.clone(), asset.data.joint_vel_target.clone()
    print(tag, "dim", term.action_dim)


run("limits_arm", tl.JointPositionToLimitsAction(cfg_of(tl.JointPositionToLimitsAction, joint_names=["arm_.*"], scale=0.8,
                                                        rescale_to_limits=True), env))
run("limits_raw", tl.JointPositionToLimitsAction(cfg_of(tl.JointPositionToLimitsAction, joint_names=[".*"], rescale_to_limits=False,
                                                        scale={"arm_.*": 0.5, "finger_.*": 2.0}, clip={"arm_.*": (-0.4, 0.6)}), env))
ema = tl.EMAJointPositionToLimitsAction(cfg_of(tl.EMAJointPositionToLimitsAction, joint_names=["arm_.*", "finger_.*"], scale=1.0,
                                               rescale_to_limits=True, alpha={"arm_.*": 0.3, "finger_.*": 0.9}), env)
ema.reset(None)
out["ema_prev_after_reset"] = ema._prev_applied_actions.clone()
run("ema", ema, steps=4)
# reset of a subset: in the reference this only works for a term over all joints (its index shapes clash otherwise)
ema_all = tl.EMAJointPositionToLimitsAction(cfg_of(tl.EMAJointPositionToLimitsAction, joint_names=[".*"], scale=1.0,
                                                   rescale_to_limits=True, alpha=0.5), env)
run("ema_all", ema_all, steps=2)
ema_all.reset(torch.tensor([3, 9, 40]))
out["ema_reset_ids"], out["ema_prev_after_partial_reset"] = torch.tensor([3, 9, 40]), ema_all._prev_applied_actions.clone()
out["ema_raw_after_partial_reset"] = ema_all.raw_actions.clone()
ema1 = tl.EMAJointPositionToLimitsAction(cfg_of(tl.EMAJointPositionToLimitsAction, joint_names=["arm_.*"], scale=1.0,
                                                rescale_to_limits=False, alpha=0.25), env)
run("ema_scalar", ema1, steps=3)
run("binary_pos", bj.BinaryJointPositionAction(cfg_of(bj.BinaryJointPositionAction, joint_names=["finger_.*"],
                                                      open_command_expr={"finger_.*": 0.04}, close_command_expr={"finger_l": 0.0, "finger_r": 0.005},
                                                      clip={"finger_l": (0.01, 0.03)}), env), boolean=True)
run("binary_vel", bj.BinaryJointVelocityAction(cfg_of(bj.BinaryJointVelocityAction, joint_names=["finger_.*"],
                                                      open_command_expr={"finger_.*": 1.5}, close_command_expr={"finger_.*": -1.5}), env))
run("non_holonomic", nh.NonHolonomicAction(cfg_of(nh.NonHolonomicAction, body_name="base", x_joint_name="base_x", y_joint_name="base_y",
                                                  yaw_joint_name="base_yaw", scale=(1.2, 0.7), offset=(0.1, -0.05),
                                                  clip={"base_x": (-1.0, 1.0)}), env))
np.savez_compressed(os.path.join(HERE, "actions2.npz"), **{k: v.numpy() for k, v in out.items()})
print("wrote actions2.npz with", len(out), "arrays")
