"""GPU: DifferentialIK controller and action term (HIP through the C ABI) against the reference's outputs
(tests/golden/ik.npz) and against the oracle with the 6x6 solve in float64."""

import pytest
import torch

from conftest import assert_close, assert_exact, load_golden
from oracle import ik as IK
from test_oracle_ik import CASES, FRANKA_PARAMS, offsets

pytestmark = pytest.mark.gpu


def test_controller_franka_shape(dev):
    from isaaclab_amd.controllers import DifferentialIKController, DifferentialIKControllerCfg

    g = load_golden("ik")
    n = g["franka_q"].shape[0]
    for method, params in FRANKA_PARAMS.items():
        ctrl = DifferentialIKController(DifferentialIKControllerCfg("pose", True, method), n, dev)
        assert ctrl.action_dim == 6
        ctrl.set_command(g["franka_cmd"].to(dev), g["franka_ee_pos"].to(dev), g["franka_ee_quat"].to(dev))
        assert_close(ctrl.ee_pos_des, g[f"franka_{method}_ee_pos_des"], rtol=1e-6, atol=1e-7, what="desired position")
        assert_close(ctrl.ee_quat_des, g[f"franka_{method}_ee_quat_des"], rtol=1e-6, atol=2e-7, what="desired orientation")
        q = ctrl.compute(g["franka_ee_pos"].to(dev), g["franka_ee_quat"].to(dev), g["franka_jac"].to(dev), g["franka_q"].to(dev))
        q64 = IK.compute_f64(g["franka_ee_pos"], g["franka_ee_quat"], ctrl.ee_pos_des.cpu(), ctrl.ee_quat_des.cpu(),
                             g["franka_jac"], g["franka_q"], "pose", method, params)
        assert_close(q, q64, rtol=1e-5, atol=2e-6, what=f"{method}: kernel vs float64 oracle")
        # well-conditioned fixtures (sigma_min >= 0.1): the fp32 reference itself is within 1e-5 of the float64 solve
        assert_close(q, g[f"franka_{method}_q_des"], rtol=1e-5, atol=1.2e-5, what=f"{method}: kernel vs reference")
    # position commands, absolute pose commands, argument checks
    ctrl = DifferentialIKController(DifferentialIKControllerCfg("position", False, "dls"), n, dev)
    assert ctrl.action_dim == 3
    with pytest.raises(ValueError):
        ctrl.set_command(g["franka_cmd"][:, :3].to(dev))
    ctrl.set_command(g["franka_cmd"][:, :3].to(dev), ee_quat=g["franka_ee_quat"].to(dev))
    assert_exact(ctrl.ee_pos_des, g["franka_cmd"][:, :3], "absolute position command")
    assert_exact(ctrl.ee_quat_des, g["franka_ee_quat"], "orientation passes through")
    q = ctrl.compute(g["franka_ee_pos"].to(dev), g["franka_ee_quat"].to(dev), g["franka_jac"].to(dev), g["franka_q"].to(dev))
    q64 = IK.compute_f64(g["franka_ee_pos"], g["franka_ee_quat"], ctrl.ee_pos_des.cpu(), ctrl.ee_quat_des.cpu(), g["franka_jac"],
                         g["franka_q"], "position", "dls", {"lambda_val": 0.01})
    assert_close(q, q64, rtol=1e-5, atol=2e-6, what="position dls")
    ctrl = DifferentialIKController(DifferentialIKControllerCfg("pose", False, "trans"), n, dev)
    assert ctrl.action_dim == 7
    pose = torch.cat([g["franka_ee_pos"] + 0.01, g["franka_ee_quat"]], -1)
    ctrl.set_command(pose.to(dev))
    assert_exact(ctrl.ee_pos_des, pose[:, :3], "absolute pose command")
    with pytest.raises(ValueError):
        DifferentialIKControllerCfg("pose", False, "newton")


class FakeArm:
    """Fixed-base 6-DOF arm around an ArticulationData: bodies = base + 6 links, physics Jacobian (N, 6, 6, 6)."""

    is_fixed_base = True
    num_joints = 6
    joint_names = [f"joint{i}" for i in range(1, 7)]
    body_names = ["link_base"] + [f"link{i}" for i in range(1, 7)]

    def __init__(self, s, dev):
        from isaaclab_amd.assets.articulation_data import ArticulationData

        n = s["joint_pos"].shape[0]
        self.data = ArticulationData(n, 6, 7, dev, joint_names=self.joint_names, body_names=self.body_names)
        root = torch.zeros(n, 13)
        root[:, :3], root[:, 3:7] = s["root_pos_w"], s["root_quat_w"]
        body = torch.zeros(n, 7, 13)
        body[:, 6, :3], body[:, 6, 3:7] = s["ee_pos_w"], s["ee_quat_w"]
        self.data.set_sim_state(root_state_w=root.to(dev), joint_pos=s["joint_pos"].to(dev), body_state_w=body.to(dev))
        jac = torch.zeros(n, 6, 6, 6)
        jac[:, 5] = s["jacobian_w"]
        self._jac = jac.to(dev)

    def find_joints(self, names, preserve_order=False):
        from isaaclab_amd.utils.string import resolve_matching_names
        return resolve_matching_names(names, self.joint_names, preserve_order)

    def find_bodies(self, names, preserve_order=False):
        from isaaclab_amd.utils.string import resolve_matching_names
        return resolve_matching_names(names, self.body_names, preserve_order)

    def get_jacobians(self):
        return self._jac

    def set_joint_position_target(self, target, joint_ids=None):
        self.data.joint_pos_target[:, joint_ids if joint_ids is not None else slice(None)] = target


def test_action_term_matches_reference(dev):
    import types

    from isaaclab_amd.controllers import DifferentialIKControllerCfg
    from isaaclab_amd.envs.mdp.actions import DifferentialInverseKinematicsAction, DifferentialInverseKinematicsActionCfg, OffsetCfg

    g = load_golden("ik")
    s = {k[6:]: v for k, v in g.items() if k.startswith("state_")}
    n = s["joint_pos"].shape[0]
    sigma_min = torch.linalg.svdvals(s["jacobian_w"].double())[:, -1]
    for tag, (ctype, rel, method, params, offset) in CASES.items():
        arm = FakeArm(s, dev)
        env = types.SimpleNamespace(num_envs=n, device=dev, scene={"robot": arm})
        cfg = DifferentialInverseKinematicsActionCfg(
            joint_names=["joint.*"], body_name="link6", scale=0.5, body_offset=OffsetCfg(*offset) if offset else None,
            controller=DifferentialIKControllerCfg(ctype, rel, method, dict(params)))
        term = DifferentialInverseKinematicsAction(cfg, env)
        assert term.action_dim == g[f"{tag}_raw"].shape[1]
        term.process_actions(g[f"{tag}_raw"].to(dev))
        assert_exact(term.raw_actions, g[f"{tag}_raw"], f"{tag} raw actions")
        assert_exact(term.processed_actions, g[f"{tag}_processed"], f"{tag} processed actions")
        c = term._ik_controller
        assert_close(c.ee_pos_des, g[f"{tag}_ee_pos_des"], rtol=1e-6, atol=2e-6, what=f"{tag} desired position")
        assert_close(c.ee_quat_des, g[f"{tag}_ee_quat_des"], rtol=1e-6, atol=5e-7, what=f"{tag} desired orientation")
        term.apply_actions()
        got = arm.data.joint_pos_target.cpu()
        op, orot = offsets(offset)
        q64 = IK.apply_actions(s["root_pos_w"], s["root_quat_w"], s["ee_pos_w"], s["ee_quat_w"], s["jacobian_w"], s["joint_pos"],
                               c.ee_pos_des.cpu(), c.ee_quat_des.cpu(), op, orot, ctype, method, params, f64=True)
        dq = (q64 - s["joint_pos"]).abs().amax(dim=1, keepdim=True)
        # relative to the size of the update: near-singular poses produce updates of tens of radians
        err = ((got - q64).abs() / (1.0 + dq)).max().item()
        assert err < 2e-5, f"{tag}: kernel vs float64 oracle, scaled error {err:.2e}"
        # against the fp32 reference itself where the Jacobian is well conditioned
        good = sigma_min > 0.05
        assert good.float().mean() > 0.5
        assert_close(got[good], g[f"{tag}_joint_pos_des"][good], rtol=3e-5, atol=3e-5, what=f"{tag}: joint targets vs reference")


def test_action_term_subset_of_joints_and_zero_pose_guard(dev):
    import types

    from isaaclab_amd.controllers import DifferentialIKControllerCfg
    from isaaclab_amd.envs.mdp.actions import DifferentialInverseKinematicsAction, DifferentialInverseKinematicsActionCfg

    g = load_golden("ik")
    s = {k[6:]: v for k, v in g.items() if k.startswith("state_")}
    n = s["joint_pos"].shape[0]
    arm = FakeArm(s, dev)
    env = types.SimpleNamespace(num_envs=n, device=dev, scene={"robot": arm})
    cfg = DifferentialInverseKinematicsActionCfg(joint_names=["joint[1-4]"], body_name="link6", scale=0.5,
                                                 controller=DifferentialIKControllerCfg("pose", True, "dls"))
    term = DifferentialInverseKinematicsAction(cfg, env)
    arm.data.joint_pos_target.fill_(7.0)
    term.process_actions(s["raw_action_pose_rel"].to(dev))
    term.apply_actions()
    got = arm.data.joint_pos_target.cpu()
    assert (got[:, 4:] == 7.0).all(), "joints outside the term keep their targets"
    pdes, qdes = term._ik_controller.ee_pos_des.cpu(), term._ik_controller.ee_quat_des.cpu()
    ref = IK.apply_actions(s["root_pos_w"], s["root_quat_w"], s["ee_pos_w"], s["ee_quat_w"], s["jacobian_w"][:, :, :4],
                           s["joint_pos"][:, :4], pdes, qdes, None, None, "pose", "dls", {"lambda_val": 0.01}, f64=True)
    dq = (ref - s["joint_pos"][:, :4]).abs().amax(dim=1, keepdim=True)
    assert ((got[:, :4] - ref).abs() / (1.0 + dq)).max() < 2e-5
    # all-zero orientations (simulator not stepped yet): targets = current joint positions (task_space_actions.py:173-178)
    arm2 = FakeArm(s, dev)
    arm2.data.body_state_w.zero_()
    term2 = DifferentialInverseKinematicsAction(cfg, types.SimpleNamespace(num_envs=n, device=dev, scene={"robot": arm2}))
    term2.apply_actions()
    assert_exact(arm2.data.joint_pos_target[:, :4], s["joint_pos"][:, :4], "zero-pose guard")
