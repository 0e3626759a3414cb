"""Golden vectors for the joint action terms and the ActionManager, produced by the REFERENCE classes
(``isaaclab/envs/mdp/actions/joint_actions.py`` under ``isaaclab/managers/action_manager.py``) on a stand-in
articulation.  Writes tests/golden/actions.npz."""

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
sys.modules["isaaclab.managers"].ActionManager = am.ActionManager
R.placeholder("isaaclab.envs.mdp.actions")
R.placeholder("isaaclab.envs.mdp.actions.actions_cfg")
ja = R.load("isaaclab.envs.mdp.actions.joint_actions", f"{R.L}/envs/mdp/actions/joint_actions.py")
ActionTermCfg = m["manager_term_cfg"].ActionTermCfg
S = m["string"]

JOINTS = ["LF_HAA", "LH_HAA", "RF_HAA", "RH_HAA", "LF_HFE", "LH_HFE", "RF_HFE", "RH_HFE", "LF_KFE", "LH_KFE", "RF_KFE", "RH_KFE"]


class Asset:
    def __init__(self, n, g):
        self.joint_names, self.num_joints = JOINTS, len(JOINTS)
        j = self.num_joints
        self.data = NS(default_joint_pos=torch.randn(1, j, generator=g).repeat(n, 1) * 0.5,
                       default_joint_vel=torch.randn(1, j, generator=g).repeat(n, 1) * 0.1,
                       joint_pos=torch.randn(n, j, generator=g), joint_pos_target=torch.zeros(n, j),
                       joint_vel_target=torch.zeros(n, j), joint_effort_target=torch.zeros(n, j))

    def find_joints(self, names, preserve_order=False):
        return S.resolve_matching_names(names, self.joint_names, preserve_order)

    def set_joint_position_target(self, t, joint_ids=None):
        self.data.joint_pos_target[:, joint_ids] = t

    def set_joint_velocity_target(self, t, joint_ids=None):
        self.data.joint_vel_target[:, joint_ids] = t

    def set_joint_effort_target(self, t, joint_ids=None):
        self.data.joint_effort_target[:, joint_ids] = t


def term_cfg(cls, **kw):
    base = dict(class_type=cls, asset_name="robot", debug_vis=False, clip=None, joint_names=[".*"], scale=1.0, offset=0.0,
                preserve_order=False, use_default_offset=True, use_zero_offset=True)
    base.update(kw)
    cfg = ActionTermCfg(**{k: base.pop(k) for k in ("class_type", "asset_name", "debug_vis", "clip")})
    for k, v in base.items():
        object.__setattr__(cfg, k, v)
    return cfg


CASES = {
    # velocity_env_cfg.py:112: JointPositionAction, all joints, scale 0.5, default offset
    "anymal": dict(joint_pos=dict(cls="JointPositionAction", scale=0.5)),
    # two terms side by side: per-joint scales and a clip on the first, relative positions on the second
    "split": dict(legs=dict(cls="JointEffortAction", joint_names=[".*_HAA", ".*_KFE"], scale={".*_HAA": 2.0, ".*_KFE": -3.0},
                            offset={".*_KFE": 0.25}, clip={".*_HAA": (-1.0, 1.5)}),
                  hips=dict(cls="RelativeJointPositionAction", joint_names=[".*_HFE"], scale=0.25)),
    # preserve_order and a velocity term with its default offset
    "ordered": dict(vel=dict(cls="JointVelocityAction", joint_names=["RH_KFE", "LF_HAA", "RF_HFE"], preserve_order=True, scale=1.5)),
}

out = {}
g = torch.Generator().manual_seed(8101)
n, steps = 96, 3
for tag, terms in CASES.items():
    asset = Asset(n, g)
    env = NS(num_envs=n, device="cpu", scene={"robot": asset}, sim=NS(has_gui=lambda: False))
    cfg = NS(**{name: term_cfg(getattr(ja, kw["cls"]), **{k: v for k, v in kw.items() if k != "cls"}) for name, kw in terms.items()})
    mgr = am.ActionManager(cfg, env)
    out[f"{tag}_default_joint_pos"], out[f"{tag}_default_joint_vel"] = asset.data.default_joint_pos.clone(), asset.data.default_joint_vel.clone()
    out[f"{tag}_joint_pos"] = asset.data.joint_pos.clone()
    rec = {k: [] for k in ("raw", "action", "prev_action", "pos_target", "vel_target", "effort_target")}
    proc = {name: [] for name in terms}
    for k in range(steps):
        raw = torch.rand(n, mgr.total_action_dim, generator=g) * 4 - 2
        mgr.process_action(raw)
        mgr.apply_action()
        rec["raw"].append(raw)
        rec["action"].append(mgr.action.clone())
        rec["prev_action"].append(mgr.prev_action.clone())
        rec["pos_target"].append(asset.data.joint_pos_target.clone())
        rec["vel_target"].append(asset.data.joint_vel_target.clone())
        rec["effort_target"].append(asset.data.joint_effort_target.clone())
        for name in terms:
            proc[name].append(mgr.get_term(name).processed_actions.clone())
    for k, v in rec.items():
        out[f"{tag}_{k}"] = torch.stack(v)
    for name, v in proc.items():
        out[f"{tag}_processed_{name}"] = torch.stack(v)
    ids = torch.tensor([0, 5, 17, 95])
    mgr.reset(ids)
    out[f"{tag}_reset_ids"] = ids
    out[f"{tag}_reset_action"], out[f"{tag}_reset_prev_action"] = mgr.action.clone(), mgr.prev_action.clone()
    for name in terms:
        out[f"{tag}_reset_raw_{name}"] = mgr.get_term(name).raw_actions.clone()
    print(tag, "total action dim", mgr.total_action_dim, "terms", mgr.active_terms)
np.savez_compressed(os.path.join(HERE, "actions.npz"), **{k: v.numpy() for k, v in out.items()})
print("wrote actions.npz with", len(out), "arrays")
