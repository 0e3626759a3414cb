"""GPU: ActionManager / joint action terms and CommandManager / command terms through their reference-shaped classes,
against the fixtures recorded from the reference's classes."""

import math
from types import SimpleNamespace as NS

import pytest
import torch

from conftest import assert_close, assert_exact, load_golden

pytestmark = pytest.mark.gpu

JOINTS = ["LF_HAA", "LH_HAA", "RF_HAA", "RH_HAA", "LF_HFE", "LH_HFE", "RF_HFE", "RH_HFE", "LF_KFE", "LH_KFE", "RF_KFE", "RH_KFE"]


def _asset(g, tag, dev):
    from isaaclab_amd.assets.articulation_data import ArticulationData
    from isaaclab_amd.envs.manager_based_step import _Asset

    n = g[f"{tag}_joint_pos"].shape[0]
    data = ArticulationData(n, len(JOINTS), 1, dev, joint_names=JOINTS, body_names=["base"])
    data.default_joint_pos[:] = g[f"{tag}_default_joint_pos"].to(dev)
    data.default_joint_vel[:] = g[f"{tag}_default_joint_vel"].to(dev)
    data.joint_pos = g[f"{tag}_joint_pos"].to(dev).contiguous()
    return _Asset(data), n


def _run_actions(dev, tag, cfg, term_names):
    from isaaclab_amd.managers import ActionManager

    g = load_golden("actions")
    asset, n = _asset(g, tag, dev)
    env = NS(num_envs=n, device=dev, scene={"robot": asset})
    mgr = ActionManager(cfg, env)
    assert mgr.active_terms == term_names
    assert mgr.total_action_dim == g[f"{tag}_raw"].shape[2]
    for k in range(g[f"{tag}_raw"].shape[0]):
        mgr.process_action(g[f"{tag}_raw"][k].to(dev))
        mgr.apply_action()
        assert_exact(mgr.action, g[f"{tag}_action"][k], f"{tag} action {k}")
        assert_exact(mgr.prev_action, g[f"{tag}_prev_action"][k], f"{tag} prev_action {k}")
        for name in term_names:
            # one multiply-add per element: the reference rounds the product first, the kernel may contract it
            assert_close(mgr.get_term(name).processed_actions, g[f"{tag}_processed_{name}"][k], rtol=1e-6, atol=1e-7,
                         what=f"{tag} processed {name} {k}")
        assert_close(asset.data.joint_pos_target, g[f"{tag}_pos_target"][k], rtol=1e-6, atol=1e-7, what=f"{tag} pos target {k}")
        assert_close(asset.data.joint_vel_target, g[f"{tag}_vel_target"][k], rtol=1e-6, atol=1e-7, what=f"{tag} vel target {k}")
        assert_close(asset.data.joint_effort_target, g[f"{tag}_effort_target"][k], rtol=1e-6, atol=1e-7, what=f"{tag} effort target {k}")
    assert mgr.reset(g[f"{tag}_reset_ids"].to(dev)) == {}
    assert_exact(mgr.action, g[f"{tag}_reset_action"], f"{tag} reset action")
    assert_exact(mgr.prev_action, g[f"{tag}_reset_prev_action"], f"{tag} reset prev_action")
    for name in term_names:
        assert_exact(mgr.get_term(name).raw_actions, g[f"{tag}_reset_raw_{name}"], f"{tag} reset raw {name}")
    return mgr


def test_joint_position_action_manager(dev):
    from isaaclab_amd.envs.mdp.actions import JointPositionActionCfg

    mgr = _run_actions(dev, "anymal", NS(joint_pos=JointPositionActionCfg(joint_names=[".*"], scale=0.5)), ["joint_pos"])
    assert mgr.get_term("joint_pos")._joint_ids == slice(None)
    with pytest.raises(ValueError, match="Invalid action shape, expected: 12, received: 11"):
        mgr.process_action(torch.zeros(96, 11, device=dev))


def test_two_action_terms(dev):
    from isaaclab_amd.envs.mdp.actions import JointEffortActionCfg, RelativeJointPositionActionCfg

    cfg = dict(legs=JointEffortActionCfg(joint_names=[".*_HAA", ".*_KFE"], scale={".*_HAA": 2.0, ".*_KFE": -3.0},
                                         offset={".*_KFE": 0.25}, clip={".*_HAA": (-1.0, 1.5)}),
               hips=RelativeJointPositionActionCfg(joint_names=[".*_HFE"], scale=0.25), unused=None)
    _run_actions(dev, "split", cfg, ["legs", "hips"])


def test_ordered_velocity_action(dev):
    from isaaclab_amd.envs.mdp.actions import JointVelocityActionCfg

    mgr = _run_actions(dev, "ordered", NS(vel=JointVelocityActionCfg(joint_names=["RH_KFE", "LF_HAA", "RF_HFE"], preserve_order=True,
                                                                   scale=1.5)), ["vel"])
    assert mgr.get_term("vel")._joint_ids == [11, 0, 6]


def test_action_manager_rejects_foreign_cfg(dev):
    from isaaclab_amd.managers import ActionManager

    with pytest.raises(TypeError, match="is not of type ActionTermCfg"):
        ActionManager(dict(bad=3.0), NS(num_envs=4, device=dev, scene={}))


def test_velocity_command_term(dev):
    from isaaclab_amd.envs.mdp.commands import UniformVelocityCommandCfg
    from isaaclab_amd.managers import CommandManager
    from isaaclab_amd.tasks import locomotion_velocity as LV
    from oracle import mathx as M

    g = load_golden("velocity_command")
    for tag in ("rough", "mixed"):
        lo, hi, stand, head, stiff = g[f"{tag}_cfg"].tolist()
        p = lambda key: g[f"{tag}_pre_{key}"]  # noqa: E731
        n = p("time_left").shape[0]
        yaw = p("heading_w")
        quat = torch.stack([torch.cos(yaw / 2), torch.zeros(n), torch.zeros(n), torch.sin(yaw / 2)], -1)
        rs = torch.cat([torch.zeros(n, 3), quat, M.quat_rotate(quat, p("root_lin_vel_b")), M.quat_rotate(quat, p("root_ang_vel_b"))], -1)
        robot = NS(data=NS(root_state_w=rs.to(dev).contiguous()))
        R = UniformVelocityCommandCfg.Ranges
        cfg = UniformVelocityCommandCfg(resampling_time_range=(lo, hi), rel_standing_envs=stand, rel_heading_envs=head,
                                        heading_command=True, heading_control_stiffness=stiff,
                                        ranges=R(lin_vel_x=(-1.0, 1.0), lin_vel_y=(-1.0, 1.0), ang_vel_z=(-1.0, 1.0),
                                                 heading=(-math.pi, math.pi)))
        env = NS(num_envs=n, device=dev, step_dt=LV.STEP_DT, scene={"robot": robot})
        mgr = CommandManager(NS(base_velocity=cfg, nothing=None), env)
        assert mgr.active_terms == ["base_velocity"]
        term = mgr.get_term("base_velocity")
        term.vel_command_b.copy_(p("vel_command_b"))
        term.heading_target.copy_(p("heading_target"))
        term.is_heading_env.copy_(p("is_heading_env"))
        term.is_standing_env.copy_(p("is_standing_env"))
        term.time_left.copy_(p("time_left"))
        term.command_counter.copy_(p("command_counter"))
        term.metrics["error_vel_xy"].copy_(p("metric_xy"))
        term.metrics["error_vel_yaw"].copy_(p("metric_yaw"))
        for k in range(g[f"{tag}_samples"].shape[0]):
            term._samples = g[f"{tag}_samples"][k].to(dev).contiguous()
            mgr.compute(LV.STEP_DT)
            assert_exact(term.time_left, g[f"{tag}_time_left"][k], f"{tag} time_left step {k}")
            assert_exact(term.command_counter, g[f"{tag}_command_counter"][k], f"{tag} counter step {k}")
            assert_exact(term.is_standing_env, g[f"{tag}_is_standing_env"][k], f"{tag} is_standing_env step {k}")
            assert_exact(term.heading_target, g[f"{tag}_heading_target"][k], f"{tag} heading_target step {k}")
            assert_close(mgr.get_command("base_velocity"), g[f"{tag}_vel_command_b"][k], atol=2e-5, what=f"{tag} command step {k}")
            assert_close(term.metrics["error_vel_xy"], g[f"{tag}_metric_xy"][k], atol=1e-6, what=f"{tag} metric xy step {k}")
            assert_close(term.metrics["error_vel_yaw"], g[f"{tag}_metric_yaw"][k], atol=1e-6, what=f"{tag} metric yaw step {k}")
        ids = g[f"{tag}_reset_ids"]
        term._reset_samples = g[f"{tag}_reset_samples"].to(dev).contiguous()
        extras = mgr.reset(ids.to(dev))
        want = g[f"{tag}_reset_extras"].tolist()
        assert extras["Metrics/base_velocity/error_vel_xy"] == pytest.approx(want[0], rel=1e-5, abs=1e-6)
        assert extras["Metrics/base_velocity/error_vel_yaw"] == pytest.approx(want[1], rel=1e-5, abs=1e-6)
        assert_exact(term.command_counter, g[f"{tag}_reset_command_counter"], f"{tag} reset counter")
        assert_exact(term.time_left, g[f"{tag}_reset_time_left"], f"{tag} reset time_left")
        assert_exact(term.command.cpu()[ids], g[f"{tag}_reset_vel_command_b"][ids], f"{tag} reset command")
        assert (term.metrics["error_vel_xy"].cpu()[ids] == 0).all()


def test_velocity_command_needs_heading_range(dev):
    from isaaclab_amd.envs.mdp.commands import UniformVelocityCommand, UniformVelocityCommandCfg

    cfg = UniformVelocityCommandCfg(resampling_time_range=(1.0, 2.0), heading_command=True)
    with pytest.raises(ValueError, match="ranges.heading"):
        UniformVelocityCommand(cfg, NS(num_envs=4, device=dev, step_dt=0.02, scene={"robot": NS(data=None)}))


def test_pose_command_term(dev):
    from isaaclab_amd.envs.mdp.commands import UniformPoseCommandCfg
    from isaaclab_amd.managers import CommandManager

    g = load_golden("pose_command")
    for tag in ("reach", "unique"):
        v = g[f"{tag}_cfg"].tolist()
        n = g[f"{tag}_pre_time_left"].shape[0]
        bodies = [f"link{i}" for i in range(8)] + ["ee", "finger_l", "finger_r"]
        robot = NS(data=NS(root_state_w=g[f"{tag}_root_state_w"].to(dev).contiguous(), body_state_w=g[f"{tag}_body_state_w"].to(dev).contiguous()),
                   num_bodies=len(bodies), find_bodies=lambda name: ([bodies.index(name)], [name]))
        R = UniformPoseCommandCfg.Ranges
        cfg = UniformPoseCommandCfg(resampling_time_range=(v[0], v[1]), body_name="ee", make_quat_unique=bool(v[14]),
                                    ranges=R(pos_x=(v[2], v[3]), pos_y=(v[4], v[5]), pos_z=(v[6], v[7]), roll=(v[8], v[9]),
                                             pitch=(v[10], v[11]), yaw=(v[12], v[13])))
        mgr = CommandManager(dict(ee_pose=cfg), NS(num_envs=n, device=dev, step_dt=1 / 30, scene={"robot": robot}))
        term = mgr.get_term("ee_pose")
        assert term.body_idx == 8
        assert_exact(term.command[:, 3], torch.ones(n), "identity orientation before the first resampling")
        term.pose_command_b.copy_(g[f"{tag}_pre_pose_command_b"])
        term.time_left.copy_(g[f"{tag}_pre_time_left"])
        term.command_counter.copy_(g[f"{tag}_pre_command_counter"])
        for k in range(3):
            term._samples = g[f"{tag}_samples"][k].to(dev).contiguous()
            mgr.compute(1 / 30)
            assert_exact(term.time_left, g[f"{tag}_time_left"][k], f"{tag} time_left step {k}")
            assert_exact(term.command_counter, g[f"{tag}_command_counter"][k], f"{tag} counter step {k}")
            assert_close(mgr.get_command("ee_pose"), g[f"{tag}_pose_command_b"][k], atol=1e-6, what=f"{tag} command step {k}")
            assert_close(term.metrics["position_error"], g[f"{tag}_position_error"][k], atol=1e-6, what=f"{tag} position error {k}")
            assert_close(term.metrics["orientation_error"], g[f"{tag}_orientation_error"][k], atol=3e-6, what=f"{tag} orientation error {k}")
        # the world-frame command of the last step's metrics belongs to the command before that step's resampling;
        # check the property against the envs that kept their command
        kept = ~g[f"{tag}_resampled"][2]
        assert_close(term.pose_command_w.cpu()[kept], g[f"{tag}_pose_command_w"][2][kept], atol=2e-6, what=f"{tag} pose_command_w")
        ids = g[f"{tag}_reset_ids"]
        term._reset_samples = g[f"{tag}_reset_samples"].to(dev).contiguous()
        extras = mgr.reset(ids.tolist())
        want = g[f"{tag}_reset_extras"].tolist()
        assert extras["Metrics/ee_pose/position_error"] == pytest.approx(want[0], rel=1e-5, abs=1e-6)
        assert extras["Metrics/ee_pose/orientation_error"] == pytest.approx(want[1], rel=1e-5, abs=1e-6)
        assert_exact(term.time_left, g[f"{tag}_reset_time_left"], f"{tag} reset time_left")
        assert_close(term.command, g[f"{tag}_reset_pose_command_b"], atol=1e-6, what=f"{tag} reset command")


# ---- limit-scaled, moving-average, binary and non-holonomic action terms (fixture: the reference classes)
JOINTS2 = ["base_x", "base_y", "base_yaw", "arm_1", "arm_2", "arm_3", "finger_l", "finger_r"]
BODIES2 = ["world", "base", "arm", "hand"]


def _asset2(g, dev):
    from isaaclab_amd.assets.articulation_data import ArticulationData
    from isaaclab_amd.envs.manager_based_step import _Asset

    n = g["joint_pos"].shape[0]
    data = ArticulationData(n, len(JOINTS2), len(BODIES2), dev, joint_names=JOINTS2, body_names=BODIES2)
    data.soft_joint_pos_limits[:] = g["soft_joint_pos_limits"].to(dev)
    data.joint_pos = g["joint_pos"].to(dev).contiguous()
    data.body_state_w[:, :, 3:7] = g["body_quat_w"].to(dev)
    return _Asset(data), n


def _drive(term, g, tag, dev, boolean_step=None, tol=1e-6):
    for k in range(g[f"{tag}_raw"].shape[0]):
        raw = g[f"{tag}_raw"][k].to(dev)
        if boolean_step == k:
            raw = raw > 0.5
        term.process_actions(raw)
        term.apply_actions()
        assert_close(term.processed_actions, g[f"{tag}_processed"][k], rtol=tol, atol=tol, what=f"{tag} processed {k}")


def test_position_to_limits_actions(dev):
    from isaaclab_amd.envs.mdp.actions import (EMAJointPositionToLimitsActionCfg, JointPositionToLimitsActionCfg)

    g = load_golden("actions2")
    asset, n = _asset2(g, dev)
    env = NS(num_envs=n, device=dev, scene={"robot": asset})
    cfg = JointPositionToLimitsActionCfg(joint_names=["arm_.*"], scale=0.8, rescale_to_limits=True)
    term = cfg.class_type(cfg, env)
    _drive(term, g, "limits_arm", dev)
    assert_close(asset.data.joint_pos_target[:, 3:6], g["limits_arm_processed"][-1], atol=1e-6, what="targets of the arm joints")
    cfg = JointPositionToLimitsActionCfg(joint_names=[".*"], rescale_to_limits=False, scale={"arm_.*": 0.5, "finger_.*": 2.0},
                                         clip={"arm_.*": (-0.4, 0.6)})
    term = cfg.class_type(cfg, env)
    assert term._joint_ids == slice(None)
    _drive(term, g, "limits_raw", dev)
    cfg = EMAJointPositionToLimitsActionCfg(joint_names=["arm_.*", "finger_.*"], scale=1.0, rescale_to_limits=True,
                                            alpha={"arm_.*": 0.3, "finger_.*": 0.9})
    ema = cfg.class_type(cfg, env)
    ema.reset(None)
    assert_close(ema._prev_applied_actions, g["ema_prev_after_reset"], atol=0, what="moving average starts from the joint positions")
    _drive(ema, g, "ema", dev, tol=2e-6)
    cfg = EMAJointPositionToLimitsActionCfg(joint_names=[".*"], scale=1.0, rescale_to_limits=True, alpha=0.5)
    ema_all = cfg.class_type(cfg, env)
    _drive(ema_all, g, "ema_all", dev, tol=2e-6)
    ema_all.reset(g["ema_reset_ids"].to(dev))
    assert_close(ema_all._prev_applied_actions, g["ema_prev_after_partial_reset"], atol=2e-6, what="partial reset of the average")
    assert_exact(ema_all.raw_actions, g["ema_raw_after_partial_reset"], "raw actions after the partial reset")
    cfg = EMAJointPositionToLimitsActionCfg(joint_names=["arm_.*"], scale=1.0, rescale_to_limits=False, alpha=0.25)
    _drive(cfg.class_type(cfg, env), g, "ema_scalar", dev, tol=2e-6)
    with pytest.raises(ValueError, match="Moving average weight must be in the range"):
        bad = EMAJointPositionToLimitsActionCfg(joint_names=["arm_.*"], alpha=1.5)
        bad.class_type(bad, env)


def test_binary_and_non_holonomic_actions(dev):
    from isaaclab_amd.envs.mdp.actions import BinaryJointPositionActionCfg, BinaryJointVelocityActionCfg, NonHolonomicActionCfg

    g = load_golden("actions2")
    asset, n = _asset2(g, dev)
    env = NS(num_envs=n, device=dev, scene={"robot": asset})
    cfg = BinaryJointPositionActionCfg(joint_names=["finger_.*"], open_command_expr={"finger_.*": 0.04},
                                       close_command_expr={"finger_l": 0.0, "finger_r": 0.005}, clip={"finger_l": (0.01, 0.03)})
    term = cfg.class_type(cfg, env)
    assert term.action_dim == 1
    _drive(term, g, "binary_pos", dev, boolean_step=1, tol=0)
    assert_exact(asset.data.joint_pos_target[:, 6:8], g["binary_pos_processed"][-1], "finger position targets")
    cfg = BinaryJointVelocityActionCfg(joint_names=["finger_.*"], open_command_expr={"finger_.*": 1.5}, close_command_expr={"finger_.*": -1.5})
    _drive(cfg.class_type(cfg, env), g, "binary_vel", dev, tol=0)
    assert_exact(asset.data.joint_vel_target[:, 6:8], g["binary_vel_processed"][-1], "finger velocity targets")
    with pytest.raises(ValueError, match="Could not resolve all joints"):
        bad = BinaryJointPositionActionCfg(joint_names=["finger_.*"], open_command_expr={"finger_l": 0.04}, close_command_expr={"finger_.*": 0.0})
        bad.class_type(bad, env)
    cfg = NonHolonomicActionCfg(body_name="base", x_joint_name="base_x", y_joint_name="base_y", yaw_joint_name="base_yaw",
                                scale=(1.2, 0.7), offset=(0.1, -0.05), clip={"base_x": (-1.0, 1.0)})
    term = cfg.class_type(cfg, env)
    _drive(term, g, "non_holonomic", dev, tol=1e-6)
    assert_close(asset.data.joint_vel_target[:, :3], g["non_holonomic_vel_target"][:, :3], atol=2e-6, what="dummy base joint velocities")


@pytest.mark.parametrize("tag", ["uniform", "uniform_simple", "terrain", "terrain_simple"])
def test_pose2d_command_terms(dev, tag):
    from isaaclab_amd.envs.mdp.commands import TerrainBasedPose2dCommandCfg, UniformPose2dCommandCfg
    from isaaclab_amd.managers import CommandManager

    g = load_golden("pose2d_command")
    n = g[f"{tag}_pre_time_left"].shape[0]
    data = NS(root_state_w=torch.zeros(n, 13, device=dev), default_root_state=g[f"{tag}_default_root_state"].to(dev).contiguous())

    class Scene(dict):
        pass

    scene = Scene(robot=NS(data=data))
    scene.env_origins = g[f"{tag}_env_origins"].to(dev)
    terrain = tag.startswith("terrain")
    if terrain:
        scene["terrain"] = NS(flat_patches={"target": g[f"{tag}_valid_targets"].to(dev)},
                              terrain_levels=g[f"{tag}_terrain_levels"].to(dev), terrain_types=g[f"{tag}_terrain_types"].to(dev))
    Cfg = TerrainBasedPose2dCommandCfg if terrain else UniformPose2dCommandCfg
    cfg = Cfg(resampling_time_range=(0.03, 0.09), simple_heading="simple" in tag,
              ranges=Cfg.Ranges(pos_x=(-3.0, 3.0), pos_y=(-2.0, 2.5), heading=(-3.0, 3.1)))
    mgr = CommandManager(dict(pose_command=cfg), NS(num_envs=n, device=dev, step_dt=0.02, scene=scene))
    term = mgr.get_term("pose_command")
    for k in ("pos_command_w", "heading_command_w", "time_left", "command_counter"):
        getattr(term, k).copy_(g[f"{tag}_pre_{k}"])
    for k in range(g[f"{tag}_samples"].shape[0]):
        data.root_state_w.copy_(g[f"{tag}_root_state_w"][k])
        term._samples = g[f"{tag}_samples"][k].to(dev).contiguous()
        term._patch_ids = g[f"{tag}_picks"][k].to(dev).contiguous()
        mgr.compute(0.02)
        assert_exact(term.time_left, g[f"{tag}_time_left"][k], f"{tag} time_left {k}")
        assert_exact(term.command_counter, g[f"{tag}_command_counter"][k], f"{tag} counter {k}")
        assert_exact(term.pos_command_w, g[f"{tag}_pos_command_w"][k], f"{tag} pos_command_w {k}")
        assert_close(term.heading_command_w, g[f"{tag}_heading_command_w"][k], atol=2e-6, what=f"{tag} heading_command_w {k}")
        # the base-frame target is a rotation of an O(10 m) vector: 1e-6 of that magnitude on every component
        assert_close(mgr.get_command("pose_command"), g[f"{tag}_command"][k], atol=1e-5, what=f"{tag} command {k}")
        assert_close(term.metrics["error_pos_2d"], g[f"{tag}_error_pos_2d"][k], atol=1e-6, what=f"{tag} error_pos_2d {k}")
        assert_close(term.metrics["error_heading"], g[f"{tag}_error_heading"][k], atol=2e-6, what=f"{tag} error_heading {k}")
    ids = g[f"{tag}_reset_ids"]
    term._reset_samples = g[f"{tag}_reset_samples"].to(dev).contiguous()
    term._reset_patch_ids = g[f"{tag}_reset_picks"].to(dev).contiguous()
    extras = mgr.reset(ids.tolist())
    want = g[f"{tag}_reset_extras"].tolist()
    for i, name in enumerate(("error_pos", "error_heading", "error_pos_2d")):
        assert extras[f"Metrics/pose_command/{name}"] == pytest.approx(want[i], rel=1e-5, abs=2e-6)
    assert_exact(term.time_left, g[f"{tag}_reset_time_left"], f"{tag} reset time_left")
    assert_exact(term.command_counter, g[f"{tag}_reset_command_counter"], f"{tag} reset counter")
    assert_exact(term.pos_command_w, g[f"{tag}_reset_pos_command_w"], f"{tag} reset pos_command_w")
    assert_close(term.heading_command_w, g[f"{tag}_reset_heading_command_w"], atol=2e-6, what=f"{tag} reset heading")
    assert_close(term.metrics["error_pos_2d"], g[f"{tag}_reset_error_pos_2d"], atol=1e-6, what=f"{tag} reset metric")
    # generated draws stay in range
    term._samples = term._patch_ids = None
    term.time_left.fill_(0.0)
    mgr.compute(0.02)
    assert float(term.time_left.min()) >= 0.03 and float(term.time_left.max()) <= 0.09


def test_normal_velocity_and_null_command_terms(dev):
    from isaaclab_amd.envs.mdp.commands import NormalVelocityCommandCfg, NullCommandCfg
    from isaaclab_amd.managers import CommandManager

    g = load_golden("normal_velocity_command")
    n = g["pre_time_left"].shape[0]
    data = NS(root_state_w=torch.zeros(n, 13, device=dev))
    cfg = NormalVelocityCommandCfg(resampling_time_range=(0.03, 0.09), rel_standing_envs=0.15,
                                   ranges=NormalVelocityCommandCfg.Ranges(mean_vel=(1.0, 0.4, 0.6), std_vel=(0.5, 0.2, 0.3),
                                                                          zero_prob=(0.1, 0.3, 0.2)))
    mgr = CommandManager(dict(base_velocity=cfg, nothing=NullCommandCfg()), NS(num_envs=n, device=dev, step_dt=0.02, scene={"robot": NS(data=data)}))
    term = mgr.get_term("base_velocity")
    term.vel_command_b.copy_(g["pre_vel_command_b"])
    term.is_standing_env.copy_(g["pre_is_standing_env"])
    term.is_zero_vel_x_env.copy_(g["pre_is_zero_vel_x_env"])
    term.is_zero_vel_y_env.copy_(g["pre_is_zero_vel_y_env"])
    term.is_zero_vel_yaw_env.copy_(g["pre_is_zero_vel_yaw_env"])
    term.time_left.copy_(g["pre_time_left"])
    term.command_counter.copy_(g["pre_command_counter"])
    term.metrics["error_vel_xy"].copy_(g["pre_error_vel_xy"])
    term.metrics["error_vel_yaw"].copy_(g["pre_error_vel_yaw"])

    def check(prefix, what):
        assert_exact(term.time_left, g[f"{prefix}time_left"], f"time_left {what}")
        assert_exact(term.command_counter, g[f"{prefix}command_counter"], f"counter {what}")
        assert_exact(mgr.get_command("base_velocity"), g[f"{prefix}vel_command_b"], f"command {what}")
        assert_exact(term.is_standing_env, g[f"{prefix}is_standing_env"], f"standing flags {what}")
        flags = torch.stack([term.is_zero_vel_x_env, term.is_zero_vel_y_env, term.is_zero_vel_yaw_env], dim=1)
        assert_exact(flags, g[f"{prefix}is_zero"], f"zero flags {what}")
        assert_close(term.metrics["error_vel_xy"], g[f"{prefix}error_vel_xy"], atol=1e-6, what=f"error_vel_xy {what}")
        assert_close(term.metrics["error_vel_yaw"], g[f"{prefix}error_vel_yaw"], atol=1e-6, what=f"error_vel_yaw {what}")

    for k in range(g["samples"].shape[0]):
        data.root_state_w.copy_(g["root_state_w"][k])
        term._samples = g["samples"][k].to(dev).contiguous()
        mgr.compute(0.02)
        sub = {key: g[key][k] for key in ("time_left", "command_counter", "vel_command_b", "is_standing_env", "is_zero", "error_vel_xy", "error_vel_yaw")}
        g.update({f"step{k}_{key}": v for key, v in sub.items()})
        check(f"step{k}_", f"step {k}")
    term._reset_samples = g["reset_samples"].to(dev).contiguous()
    extras = mgr.reset(g["reset_ids"].tolist())
    want = g["reset_extras"].tolist()
    assert extras["Metrics/base_velocity/error_vel_xy"] == pytest.approx(want[0], rel=1e-5)
    assert extras["Metrics/base_velocity/error_vel_yaw"] == pytest.approx(want[1], rel=1e-5)
    check("reset_", "after reset")
    # generated draws: time left in range, the three components spread around +-mean
    term._samples = None
    term.time_left.fill_(0.0)
    mgr.compute(0.02)
    assert float(term.time_left.min()) >= 0.03 and float(term.time_left.max()) <= 0.09
    moving = ~term.is_standing_env & ~term.is_zero_vel_x_env
    assert abs(float(term.vel_command_b[moving, 0].abs().mean()) - 1.0) < 0.25
    assert 0.2 < float((term.vel_command_b[moving, 0] > 0).float().mean()) < 0.8
    with pytest.raises(RuntimeError, match="does not generate any commands"):
        mgr.get_command("nothing")
