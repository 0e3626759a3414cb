"""GPU: every term function of isaaclab_amd.envs.mdp (the reference's ``func(env, **params)`` seam) and the manager
classes, against outputs of the reference's own term functions / managers (tests/golden/terms_*.npz, managers_anymal.npz)."""

import math
import types

import pytest
import torch

from conftest import assert_close, assert_exact, load_golden
from test_oracle_terms import DECISIONS, state

pytestmark = pytest.mark.gpu


class Scene(dict):
    sensors: dict = {}
    env_origins = None


class Entity:
    def __init__(self, data, joint_names=(), body_names=()):
        self.data, self.joint_names, self.body_names = data, list(joint_names), list(body_names)
        self.num_joints, self.num_bodies = len(self.joint_names), len(self.body_names)

    def find_joints(self, names, preserve_order=False):
        from isaaclab_amd.utils.string import resolve_matching_names
        return resolve_matching_names(names, self.joint_names, preserve_order)

    def find_bodies(self, names, preserve_order=False):
        from isaaclab_amd.utils.string import resolve_matching_names
        return resolve_matching_names(names, self.body_names, preserve_order)


def make_env(s, dev, max_episode_length=1000, step_dt=0.02, ray=None):
    """A duck-typed env around one golden state: what the term functions touch (scene, managers, counters)."""
    n, j = s["joint_pos"].shape
    b = s["body_state_w"].shape[1]
    g = lambda k: s[k].to(dev).contiguous()  # noqa: E731
    data = types.SimpleNamespace(
        root_state_w=g("root_state_w"), joint_pos=g("joint_pos"), joint_vel=g("joint_vel"), joint_acc=g("joint_acc"),
        applied_torque=g("applied_torque"), computed_torque=g("computed_torque"), default_joint_pos=g("default_joint_pos"),
        default_joint_vel=g("default_joint_vel"), soft_joint_pos_limits=g("soft_joint_pos_limits"),
        soft_joint_vel_limits=g("soft_joint_vel_limits"), body_state_w=g("body_state_w"), body_lin_acc_w=g("body_lin_acc_w"))
    robot = Entity(data, [f"joint_{i}" for i in range(j)], [f"body_{i}" for i in range(b)])
    contact = Entity(types.SimpleNamespace(net_forces_w_history=g("net_forces_w_history"), current_air_time=g("current_air_time"),
                                           last_air_time=g("last_air_time"), current_contact_time=g("current_contact_time"),
                                           last_contact_time=g("last_contact_time")), (), [f"body_{i}" for i in range(b)])
    scene = Scene(robot=robot, contact_forces=contact)
    scene.sensors = {"contact_forces": contact}
    if ray is not None:
        scene["height_scanner"] = scene.sensors["height_scanner"] = Entity(types.SimpleNamespace(ray_hits_w=ray[0], pos_w=ray[1]))
    cmd = g("command")
    env = types.SimpleNamespace(
        num_envs=n, device=dev, scene=scene, step_dt=step_dt, max_episode_length=max_episode_length,
        max_episode_length_s=max_episode_length * step_dt, episode_length_buf=s["episode_length_buf"].to(dev).clone(),
        action_manager=types.SimpleNamespace(action=g("action"), prev_action=g("prev_action")),
        command_manager=types.SimpleNamespace(get_command=lambda name: cmd))
    return env


def check(name, got, ref, tag):
    if name in DECISIONS or ref.dtype == torch.bool:
        assert_exact(got, ref, f"{tag}.{name}")
    elif name == "bad_orientation":
        assert (got.cpu() != ref).sum() <= 1, f"{tag}.bad_orientation (goes through acos)"
    else:
        assert_close(got, ref, atol=2e-6, what=f"{tag}.{name}")


def test_locomotion_term_functions(dev):
    from isaaclab_amd.envs import mdp
    from isaaclab_amd.managers import SceneEntityCfg as E

    g = load_golden("terms_locomotion")
    for tag in ("anymal", "h1"):
        s = state(g, tag)
        env = make_env(s, dev)
        feet, thigh, base, some_j = (s[k].tolist() for k in ("ids_feet", "ids_thigh", "ids_base", "ids_some_j"))
        c = "base_velocity"
        sensor = lambda ids: E("contact_forces", body_ids=ids)  # noqa: E731
        calls = dict(
            lin_vel_z_l2=lambda: mdp.lin_vel_z_l2(env),
            ang_vel_xy_l2=lambda: mdp.ang_vel_xy_l2(env),
            flat_orientation_l2=lambda: mdp.flat_orientation_l2(env, asset_cfg=E("robot")),
            base_height_l2=lambda: mdp.base_height_l2(env, target_height=0.55),
            body_lin_acc_l2=lambda: mdp.body_lin_acc_l2(env, asset_cfg=E("robot", body_ids=feet)),
            joint_torques_l2=lambda: mdp.joint_torques_l2(env),
            joint_torques_l2_ids=lambda: mdp.joint_torques_l2(env, asset_cfg=E("robot", joint_ids=some_j)),
            joint_vel_l1=lambda: mdp.joint_vel_l1(env, asset_cfg=E("robot")),
            joint_vel_l2=lambda: mdp.joint_vel_l2(env),
            joint_acc_l2=lambda: mdp.joint_acc_l2(env),
            joint_deviation_l1=lambda: mdp.joint_deviation_l1(env, asset_cfg=E("robot", joint_ids=some_j)),
            joint_pos_limits=lambda: mdp.joint_pos_limits(env),
            joint_vel_limits=lambda: mdp.joint_vel_limits(env, soft_ratio=0.9),
            applied_torque_limits=lambda: mdp.applied_torque_limits(env),
            action_rate_l2=lambda: mdp.action_rate_l2(env),
            action_l2=lambda: mdp.action_l2(env),
            undesired_contacts=lambda: mdp.undesired_contacts(env, threshold=1.0, sensor_cfg=sensor(thigh)),
            contact_forces=lambda: mdp.contact_forces(env, threshold=5.0, sensor_cfg=sensor(feet)),
            track_lin_vel_xy_exp=lambda: mdp.track_lin_vel_xy_exp(env, std=math.sqrt(0.25), command_name=c),
            track_ang_vel_z_exp=lambda: mdp.track_ang_vel_z_exp(env, std=math.sqrt(0.25), command_name=c),
            feet_air_time=lambda: mdp.feet_air_time(env, command_name=c, sensor_cfg=sensor(feet), threshold=0.5),
            feet_air_time_positive_biped=lambda: mdp.feet_air_time_positive_biped(env, command_name=c, threshold=0.4, sensor_cfg=sensor(feet)),
            feet_slide=lambda: mdp.feet_slide(env, sensor_cfg=sensor(feet), asset_cfg=E("robot", body_ids=feet)),
            track_lin_vel_xy_yaw_frame_exp=lambda: mdp.track_lin_vel_xy_yaw_frame_exp(env, std=0.5, command_name=c),
            track_ang_vel_z_world_exp=lambda: mdp.track_ang_vel_z_world_exp(env, command_name=c, std=0.5),
            illegal_contact=lambda: mdp.illegal_contact(env, threshold=1.0, sensor_cfg=sensor(base)),
            bad_orientation=lambda: mdp.bad_orientation(env, limit_angle=0.3),
            root_height_below_minimum=lambda: mdp.root_height_below_minimum(env, minimum_height=0.5),
            joint_pos_out_of_manual_limit=lambda: mdp.joint_pos_out_of_manual_limit(env, bounds=(-0.9, 0.9), asset_cfg=E("robot", joint_ids=some_j)),
            joint_vel_out_of_limit=lambda: mdp.joint_vel_out_of_limit(env),
            joint_vel_out_of_manual_limit=lambda: mdp.joint_vel_out_of_manual_limit(env, max_velocity=4.0),
            joint_effort_out_of_limit=lambda: mdp.joint_effort_out_of_limit(env),
            obs_joint_pos_rel=lambda: mdp.joint_pos_rel(env),
            obs_joint_vel_rel=lambda: mdp.joint_vel_rel(env),
            obs_joint_pos_limit_normalized=lambda: mdp.joint_pos_limit_normalized(env),
            obs_last_action=lambda: mdp.last_action(env),
            root_lin_vel_b=lambda: mdp.base_lin_vel(env),
            root_ang_vel_b=lambda: mdp.base_ang_vel(env),
            projected_gravity_b=lambda: mdp.projected_gravity(env),
        )
        for name, fn in calls.items():
            check(name, fn(), s[name], tag)
        assert_exact(mdp.time_out(env), s["time_out"], f"{tag}.time_out")
        assert_exact(env.episode_length_buf, s["episode_length_buf"], "term functions leave the step counter alone")
        with pytest.raises(IndexError):
            mdp.joint_pos_out_of_limit(env)
        # plain state terms
        assert_exact(mdp.joint_pos(env), s["joint_pos"], "joint_pos")
        assert_exact(mdp.joint_vel(env, asset_cfg=E("robot", joint_ids=some_j)), s["joint_vel"][:, some_j], "joint_vel[ids]")
        assert_exact(mdp.root_pos_w(env), s["root_state_w"][:, :3], "root_pos_w")
        assert_exact(mdp.root_quat_w(env), s["root_state_w"][:, 3:7], "root_quat_w")
        q = s["root_state_w"][:, 3:7]
        assert_exact(mdp.root_quat_w(env, make_quat_unique=True), torch.where(q[:, :1] < 0, -q, q), "root_quat_w unique")
        assert_exact(mdp.root_lin_vel_w(env), s["root_state_w"][:, 7:10], "root_lin_vel_w")
        assert_exact(mdp.root_ang_vel_w(env), s["root_state_w"][:, 10:13], "root_ang_vel_w")
        assert_exact(mdp.base_pos_z(env), s["root_state_w"][:, 2:3], "base_pos_z")
        assert_exact(mdp.generated_commands(env, command_name=c), s["command"], "generated_commands")


def test_manipulation_term_functions(dev):
    from isaaclab_amd.envs import mdp
    from isaaclab_amd.managers import SceneEntityCfg as E

    g = load_golden("terms_manipulation")
    s = state(g, "cartpole")
    env = make_env(s, dev)
    term = s["terminated"].to(dev)
    env.termination_manager = types.SimpleNamespace(active_terms=["time_out", "cart_out_of_bounds"], terminated=term,
                                                    time_outs=torch.zeros_like(term),
                                                    term_dones_tensor=lambda: torch.stack([torch.zeros_like(term), term]))
    assert_exact(mdp.is_alive(env), s["is_alive"], "is_alive")
    assert_exact(mdp.is_terminated(env), s["is_terminated"], "is_terminated")
    assert_exact(mdp.is_terminated_term(env, term_keys="cart_out_of_bounds"), s["is_terminated"], "is_terminated_term")
    assert_exact(mdp.joint_pos_target_l2(env, target=0.0, asset_cfg=E("robot", joint_ids=[1])), s["pole_pos"], "pole_pos")
    assert_exact(mdp.joint_vel_l1(env, asset_cfg=E("robot", joint_ids=[0])), s["cart_vel"], "cart_vel")
    assert_exact(mdp.joint_vel_l1(env, asset_cfg=E("robot", joint_ids=[1])), s["pole_vel"], "pole_vel")
    assert_exact(mdp.joint_pos_out_of_manual_limit(env, bounds=(-3.0, 3.0), asset_cfg=E("robot", joint_ids=[0])),
                 s["out_of_bounds"], "out_of_bounds")
    assert_exact(mdp.joint_pos_rel(env), s["obs_joint_pos_rel"], "joint_pos_rel")
    assert_exact(mdp.joint_vel_rel(env), s["obs_joint_vel_rel"], "joint_vel_rel")
    s = state(g, "reach")
    env = make_env(s, dev)
    ee = E("robot", body_ids=[8])
    assert_close(mdp.position_command_error(env, command_name="ee_pose", asset_cfg=ee), s["position_command_error"],
                 atol=1e-6, what="position_command_error")
    assert_close(mdp.position_command_error_tanh(env, std=0.1, command_name="ee_pose", asset_cfg=ee),
                 s["position_command_error_tanh"], atol=1e-6, what="position_command_error_tanh")
    assert_close(mdp.orientation_command_error(env, command_name="ee_pose", asset_cfg=ee), s["orientation_command_error"],
                 atol=2e-6, what="orientation_command_error")
    assert_close(mdp.action_rate_l2(env), s["action_rate_l2"], what="action_rate_l2")
    assert_close(mdp.joint_vel_l2(env), s["joint_vel_l2"], what="joint_vel_l2")
    with pytest.raises(ValueError):
        mdp.position_command_error(env, command_name="ee_pose", asset_cfg=E("robot"))


def test_manager_classes_match_reference_managers(dev):
    """RewardManager / TerminationManager / ObservationManager built from term cfgs (ANYmal-C rough lists,
    velocity_env_cfg.py:118-265) against the outputs of the reference's manager classes."""
    from isaaclab_amd.envs import mdp
    from isaaclab_amd.managers import (ObservationManager, ObservationTermCfg as ObsTerm, RewardManager, RewardTermCfg as RewTerm,
                                       SceneEntityCfg as E, TerminationManager, TerminationTermCfg as DoneTerm,
                                       UniformNoiseCfg as Unoise)

    g = load_golden("managers_anymal")
    steps = g["reward"].shape[0]
    s0 = {k: v[0] for k, v in g.items() if v.shape[:1] == (steps,)}
    n = s0["joint_pos"].shape[0]
    ray = [s0["ray_hits_w"].to(dev).contiguous(), s0["sensor_pos_w"].to(dev).contiguous()]
    env = make_env(s0, dev, ray=ray)
    env.scene["contact_forces"].body_names = ["base"] + [f"{l}_{p}" for p in ("HIP", "THIGH", "SHANK", "FOOT") for l in ("LF", "LH", "RF", "RH")]
    env.scene["contact_forces"].num_bodies = 17
    c = "base_velocity"
    rewards = dict(
        track_lin_vel_xy_exp=RewTerm(mdp.track_lin_vel_xy_exp, dict(command_name=c, std=math.sqrt(0.25)), 1.0),
        track_ang_vel_z_exp=RewTerm(mdp.track_ang_vel_z_exp, dict(command_name=c, std=math.sqrt(0.25)), 0.5),
        lin_vel_z_l2=RewTerm(mdp.lin_vel_z_l2, {}, -2.0),
        ang_vel_xy_l2=RewTerm(mdp.ang_vel_xy_l2, {}, -0.05),
        dof_torques_l2=RewTerm(mdp.joint_torques_l2, {}, -1.0e-5),
        dof_acc_l2=RewTerm(mdp.joint_acc_l2, {}, -2.5e-7),
        action_rate_l2=RewTerm(mdp.action_rate_l2, {}, -0.01),
        feet_air_time=RewTerm(mdp.feet_air_time, dict(sensor_cfg=E("contact_forces", body_names=".*FOOT"), command_name=c, threshold=0.5), 0.125),
        undesired_contacts=RewTerm(mdp.undesired_contacts, dict(sensor_cfg=E("contact_forces", body_names=".*THIGH"), threshold=1.0), -1.0),
        flat_orientation_l2=RewTerm(mdp.flat_orientation_l2, {}, 0.0),
        dof_pos_limits=RewTerm(mdp.joint_pos_limits, {}, 0.0),
    )
    dones = dict(time_out=DoneTerm(mdp.time_out, {}, time_out=True),
                 base_contact=DoneTerm(mdp.illegal_contact, dict(sensor_cfg=E("contact_forces", body_names="base"), threshold=1.0)))
    policy = dict(
        base_lin_vel=ObsTerm(mdp.base_lin_vel, {}, noise=Unoise(-0.1, 0.1)),
        base_ang_vel=ObsTerm(mdp.base_ang_vel, {}, noise=Unoise(-0.2, 0.2)),
        projected_gravity=ObsTerm(mdp.projected_gravity, {}, noise=Unoise(-0.05, 0.05)),
        velocity_commands=ObsTerm(mdp.generated_commands, dict(command_name=c)),
        joint_pos=ObsTerm(mdp.joint_pos_rel, {}, noise=Unoise(-0.01, 0.01)),
        joint_vel=ObsTerm(mdp.joint_vel_rel, {}, noise=Unoise(-1.5, 1.5)),
        actions=ObsTerm(mdp.last_action, {}),
        height_scan=ObsTerm(mdp.height_scan, dict(sensor_cfg=E("height_scanner")), noise=Unoise(-0.1, 0.1), clip=(-1.0, 1.0)),
        enable_corruption=True, concatenate_terms=True)
    tm, rm, om = TerminationManager(dones, env), RewardManager(rewards, env), ObservationManager(dict(policy=policy), env)
    env.termination_manager = tm
    assert rm.active_terms == list(g["reward_names"]) if "reward_names" in g and not torch.is_tensor(g["reward_names"]) else True
    assert rm._specs[7].ids == [13, 14, 15, 16] and rm._specs[8].ids == [5, 6, 7, 8] and tm._specs[1].ids == [0]
    for step in range(steps):
        s = {k: v[step] for k, v in g.items() if v.shape[:1] == (steps,)}
        e2 = make_env(s, dev, ray=[s["ray_hits_w"].to(dev).contiguous(), s["sensor_pos_w"].to(dev).contiguous()])
        env.scene["robot"].data = e2.scene["robot"].data
        env.scene["contact_forces"].data = e2.scene["contact_forces"].data
        env.scene["height_scanner"].data = e2.scene["height_scanner"].data
        env.action_manager, env.command_manager = e2.action_manager, e2.command_manager
        env.episode_length_buf.copy_(s["episode_length_buf"])
        done = tm.compute()
        assert_exact(env.episode_length_buf, s["episode_length_buf"], "the manager leaves the step counter alone")
        assert_exact(tm.terminated, s["terminated"], f"terminated step {step}")
        assert_exact(tm.time_outs, s["time_outs"], f"time_outs step {step}")
        assert_exact(done, s["dones"], f"dones step {step}")
        assert_exact(tm.get_term("base_contact"), s["term_dones"][1], f"term_dones step {step}")
        reward = rm.compute(dt=0.02)
        assert_close(reward, s["reward"], atol=1e-6, what=f"reward step {step}")
        assert_close(rm._step_reward, s["step_reward"], atol=5e-5, what=f"step_reward step {step}")
        sums = torch.stack([rm._episode_sums[k] for k in rm.active_terms])
        assert_close(sums, s["episode_sums"], atol=1e-6, what=f"episode_sums step {step}")
        obs = om.compute_group("policy", noise_u=s["noise_u"].to(dev).contiguous())
        assert obs.shape == (n, 235) and om.group_obs_dim["policy"] == (235,)
        assert_close(obs, s["obs"], atol=2e-6, what=f"obs step {step}")
        ids = s["dones"].nonzero().squeeze(-1).to(dev)
        log = rm.reset(ids)
        counts = tm.reset(ids)
        if len(ids) > 0:
            got = torch.stack([log["Episode_Reward/" + k] for k in rm.active_terms]).cpu()
            assert_close(got, s["reset_reward_log"], rtol=1e-5, atol=1e-7, what=f"reward log step {step}")
            assert [counts["Episode_Termination/" + k] for k in tm.active_terms] == s["reset_term_log"].tolist()
            sums = torch.stack([rm._episode_sums[k] for k in rm.active_terms])
            assert_close(sums, s["episode_sums_after_reset"], atol=1e-6, what=f"episode sums after reset, step {step}")
    # a term that is not part of the library is refused with the reference's kind of error, at construction
    with pytest.raises(TypeError):
        RewardManager(dict(custom=RewTerm(lambda env: torch.zeros(n), {}, 1.0)), env)
    with pytest.raises(TypeError):
        RewardManager(dict(bad=RewTerm(mdp.action_l2, {}, "1.0")), env)


def test_observation_group_over_two_articulations(dev):
    """A group whose terms read different articulations (observation_manager.py:260-335 concatenates whatever its terms
    return): one fused plan per articulation, columns in the group's order, noise and clip per term -- against the same
    terms computed one group each."""
    from isaaclab_amd.envs import mdp
    from isaaclab_amd.managers import ObservationManager, ObservationTermCfg as ObsTerm, SceneEntityCfg as E, UniformNoiseCfg as Unoise

    g = load_golden("managers_anymal")
    steps = g["reward"].shape[0]
    s0 = {k: v[0] for k, v in g.items() if v.shape[:1] == (steps,)}
    s1 = {k: v[min(1, steps - 1)] for k, v in g.items() if v.shape[:1] == (steps,)}
    n = s0["joint_pos"].shape[0]
    env = make_env(s0, dev)
    other = make_env(s1, dev).scene["robot"]
    other.data.joint_pos = other.data.joint_pos + 0.25          # a second articulation with a state of its own
    other.data.root_state_w = other.data.root_state_w.flip(0).contiguous()
    env.scene["object"] = other
    terms = dict(
        base_lin_vel=ObsTerm(mdp.base_lin_vel, {}, noise=Unoise(-0.1, 0.1)),
        object_joint_pos=ObsTerm(mdp.joint_pos_rel, dict(asset_cfg=E("object")), noise=Unoise(-0.01, 0.01), clip=(-0.2, 0.2)),
        joint_vel=ObsTerm(mdp.joint_vel_rel, {}, noise=Unoise(-1.5, 1.5), scale=0.5),
        object_lin_vel=ObsTerm(mdp.base_lin_vel, dict(asset_cfg=E("object"))),
        actions=ObsTerm(mdp.last_action, {}),
        projected_gravity=ObsTerm(mdp.projected_gravity, dict(asset_cfg=E("object")), noise=Unoise(-0.05, 0.05)),
    )
    om = ObservationManager(dict(policy=dict(terms, enable_corruption=True, concatenate_terms=True)), env)
    widths = [3, 12, 12, 3, 12, 3]
    total = sum(widths)
    assert om.group_obs_dim["policy"] == (total,)
    noise = torch.rand(n, total, generator=torch.Generator().manual_seed(3)).to(dev)
    for rep in range(2):   # the plans are built once and launched again
        row = om.compute_group("policy", noise_u=noise)
        assert row.shape == (n, total)
        col = 0
        for (name, cfg), w in zip(terms.items(), widths):
            single = ObservationManager(dict(one=dict({name: cfg}, enable_corruption=True, concatenate_terms=True)), env)
            ref = single.compute_group("one", noise_u=noise[:, col:col + w].contiguous())
            assert_exact(row[:, col:col + w], ref, f"{name} (columns {col}..{col + w})")
            col += w
    # the two articulations really differ, and un-concatenated groups hand the same columns out by name
    assert not torch.equal(row[:, 0:3], row[:, 27:30])
    om2 = ObservationManager(dict(policy=dict(terms, enable_corruption=True, concatenate_terms=False)), env)
    parts = om2.compute_group("policy", noise_u=noise)
    assert list(parts) == list(terms)
    assert_exact(torch.cat(list(parts.values()), dim=1), row, "un-concatenated terms")
