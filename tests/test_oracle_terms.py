"""CPU: oracle term functions, command terms and manager arithmetic against reference-generated fixtures."""

import math

import torch

from conftest import assert_close, assert_exact, load_golden
from oracle import command as C
from oracle import managers as MG
from oracle import terms as T

SQ = math.sqrt(0.25)


def state(g, tag):
    p = tag + "_"
    return {k[len(p):]: v for k, v in g.items() if k.startswith(p)}


def locomotion_terms(s):
    feet, thigh, base, some_j = (s[k].tolist() for k in ("ids_feet", "ids_thigh", "ids_base", "ids_some_j"))
    hist = s["net_forces_w_history"]
    cmd = s["command"]
    bs = s["body_state_w"]
    rs = s["root_state_w"]
    out = dict(
        lin_vel_z_l2=T.lin_vel_z_l2(s["root_lin_vel_b"]),
        ang_vel_xy_l2=T.ang_vel_xy_l2(s["root_ang_vel_b"]),
        flat_orientation_l2=T.flat_orientation_l2(s["projected_gravity_b"]),
        base_height_l2=T.base_height_l2(rs[:, :3], 0.55),
        body_lin_acc_l2=T.body_lin_acc_l2(s["body_lin_acc_w"], feet),
        joint_torques_l2=T.joint_torques_l2(s["applied_torque"]),
        joint_torques_l2_ids=T.joint_torques_l2(s["applied_torque"], some_j),
        joint_vel_l1=T.joint_vel_l1(s["joint_vel"]),
        joint_vel_l2=T.joint_vel_l2(s["joint_vel"]),
        joint_acc_l2=T.joint_acc_l2(s["joint_acc"]),
        joint_deviation_l1=T.joint_deviation_l1(s["joint_pos"], s["default_joint_pos"], some_j),
        joint_pos_limits=T.joint_pos_limits(s["joint_pos"], s["soft_joint_pos_limits"]),
        joint_vel_limits=T.joint_vel_limits(s["joint_vel"], s["soft_joint_vel_limits"], 0.9),
        applied_torque_limits=T.applied_torque_limits(s["applied_torque"], s["computed_torque"]),
        action_rate_l2=T.action_rate_l2(s["action"], s["prev_action"]),
        action_l2=T.action_l2(s["action"]),
        undesired_contacts=T.undesired_contacts(hist, 1.0, thigh),
        contact_forces=T.contact_forces(hist, 5.0, feet),
        track_lin_vel_xy_exp=T.track_lin_vel_xy_exp(cmd, s["root_lin_vel_b"], SQ),
        track_ang_vel_z_exp=T.track_ang_vel_z_exp(cmd, s["root_ang_vel_b"], SQ),
        feet_air_time=T.feet_air_time(s["current_contact_time"], s["last_air_time"], cmd, 0.02, 0.5, feet),
        feet_air_time_positive_biped=T.feet_air_time_positive_biped(s["current_air_time"], s["current_contact_time"], cmd, 0.4, feet),
        feet_slide=T.feet_slide(hist, bs[..., 7:10], feet, feet),
        track_lin_vel_xy_yaw_frame_exp=T.track_lin_vel_xy_yaw_frame_exp(cmd, rs[:, 3:7], rs[:, 7:10], 0.5),
        track_ang_vel_z_world_exp=T.track_ang_vel_z_world_exp(cmd, rs[:, 10:13], 0.5),
        time_out=T.time_out(s["episode_length_buf"], 1000),
        illegal_contact=T.illegal_contact(hist, 1.0, base),
        bad_orientation=T.bad_orientation(s["projected_gravity_b"], 0.3),
        root_height_below_minimum=T.root_height_below_minimum(rs[:, :3], 0.5),
        joint_pos_out_of_manual_limit=T.joint_pos_out_of_manual_limit(s["joint_pos"], (-0.9, 0.9), some_j),
        joint_vel_out_of_limit=T.joint_vel_out_of_limit(s["joint_vel"], s["soft_joint_vel_limits"]),
        joint_vel_out_of_manual_limit=T.joint_vel_out_of_manual_limit(s["joint_vel"], 4.0),
        joint_effort_out_of_limit=T.joint_effort_out_of_limit(s["computed_torque"], s["applied_torque"]),
        obs_joint_pos_rel=T.joint_pos_rel(s["joint_pos"], s["default_joint_pos"]),
        obs_joint_vel_rel=T.joint_vel_rel(s["joint_vel"], s["default_joint_vel"]),
        obs_joint_pos_limit_normalized=T.joint_pos_limit_normalized(s["joint_pos"], s["soft_joint_pos_limits"]),
        obs_last_action=s["action"],
    )
    return out


DECISIONS = {"time_out", "illegal_contact", "root_height_below_minimum", "joint_pos_out_of_manual_limit",
             "joint_vel_out_of_limit", "joint_vel_out_of_manual_limit", "joint_effort_out_of_limit", "undesired_contacts"}


def test_locomotion_terms():
    g = load_golden("terms_locomotion")
    assert int(g["joint_pos_out_of_limit_raises"]) == 1  # terminations.py:85-87 cannot run in the reference
    for tag in ("anymal", "h1"):
        s = state(g, tag)
        for name, got in locomotion_terms(s).items():
            ref = s[name]
            if name in DECISIONS or ref.dtype == torch.bool:
                assert_exact(got, ref, f"{tag}.{name}")
            elif name == "bad_orientation":
                assert (got != ref).sum() <= 1, f"{tag}.bad_orientation (goes through acos)"
            else:
                assert_close(got, ref, what=f"{tag}.{name}")
        # both outcomes of every decision are present in the fixture
        for name in ("time_out", "illegal_contact", "joint_vel_out_of_manual_limit"):
            assert 0 < s[name].float().mean() < 1, name
        assert (s["feet_air_time"] != 0).any() and (s["undesired_contacts"] > 0).any()


def test_manipulation_terms():
    g = load_golden("terms_manipulation")
    s = state(g, "cartpole")
    assert_exact(T.is_alive(s["terminated"]), s["is_alive"], "is_alive")
    assert_exact(T.is_terminated(s["terminated"]), s["is_terminated"], "is_terminated")
    assert_exact(T.joint_pos_target_l2(s["joint_pos"], 0.0, [1]), s["pole_pos"], "cartpole pole_pos")
    assert_exact(T.joint_vel_l1(s["joint_vel"], [0]), s["cart_vel"], "cart_vel")
    assert_exact(T.joint_vel_l1(s["joint_vel"], [1]), s["pole_vel"], "pole_vel")
    assert_exact(T.joint_pos_out_of_manual_limit(s["joint_pos"], (-3.0, 3.0), [0]), s["out_of_bounds"], "out_of_bounds")
    assert 0 < s["out_of_bounds"].float().mean() < 1
    assert_exact(T.joint_pos_rel(s["joint_pos"], s["default_joint_pos"]), s["obs_joint_pos_rel"], "joint_pos_rel")
    s = state(g, "reach")
    bs = s["body_state_w"][:, 8]
    assert_close(T.position_command_error(s["command"], s["root_state_w"], bs[:, :3]), s["position_command_error"],
                 what="position_command_error")
    assert_close(T.position_command_error_tanh(s["command"], s["root_state_w"], bs[:, :3], 0.1),
                 s["position_command_error_tanh"], what="position_command_error_tanh")
    assert_close(T.orientation_command_error(s["command"], s["root_state_w"], bs[:, 3:7]), s["orientation_command_error"],
                 what="orientation_command_error")
    assert_close(T.action_rate_l2(s["action"], s["prev_action"]), s["action_rate_l2"], what="action_rate_l2")
    assert_close(T.joint_vel_l2(s["joint_vel"]), s["joint_vel_l2"], what="joint_vel_l2")


def _vel_cfg(g, tag):
    lo, hi, stand, head, stiff = g[f"{tag}_cfg"].tolist()
    return C.VelocityCommandCfg(resampling_time_range=(lo, hi), rel_standing_envs=stand, rel_heading_envs=head,
                                heading_control_stiffness=stiff)


def test_velocity_command():
    g = load_golden("velocity_command")
    for tag in ("rough", "mixed"):
        cfg = _vel_cfg(g, tag)
        p = lambda k: g[f"{tag}_pre_{k}"]  # noqa: E731
        s = C.VelocityCommandState(p("vel_command_b").clone(), p("heading_target").clone(), p("is_heading_env").clone(),
                                   p("is_standing_env").clone(), p("time_left").clone(), p("command_counter").clone(),
                                   p("metric_xy").clone(), p("metric_yaw").clone())
        for k in range(g[f"{tag}_samples"].shape[0]):
            mask = C.velocity_compute(cfg, s, 0.02, 0.02, p("root_lin_vel_b"), p("root_ang_vel_b"), p("heading_w"),
                                      g[f"{tag}_samples"][k])
            assert_exact(mask, g[f"{tag}_resampled"][k], f"{tag} resample mask step {k}")
            assert_exact(s.time_left, g[f"{tag}_time_left"][k], f"{tag} time_left step {k}")
            assert_exact(s.command_counter, g[f"{tag}_command_counter"][k], f"{tag} counter step {k}")
            assert_exact(s.is_heading_env, g[f"{tag}_is_heading_env"][k], f"{tag} is_heading_env step {k}")
            assert_exact(s.is_standing_env, g[f"{tag}_is_standing_env"][k], f"{tag} is_standing_env step {k}")
            assert_exact(s.heading_target, g[f"{tag}_heading_target"][k], f"{tag} heading_target step {k}")
            assert_close(s.vel_command_b, g[f"{tag}_vel_command_b"][k], what=f"{tag} command step {k}")
            assert_exact(s.metric_error_vel_xy, g[f"{tag}_metric_xy"][k], f"{tag} metric xy step {k}")
            assert_exact(s.metric_error_vel_yaw, g[f"{tag}_metric_yaw"][k], f"{tag} metric yaw step {k}")
        assert g[f"{tag}_resampled"][0].any()
        ids = g[f"{tag}_reset_ids"]
        mask = torch.zeros(s.time_left.shape[0], dtype=torch.bool).index_fill_(0, ids, True)
        extras = C.velocity_reset(cfg, s, mask, g[f"{tag}_reset_samples"])
        assert_close(torch.tensor([extras["error_vel_xy"], extras["error_vel_yaw"]], dtype=torch.float64),
                     g[f"{tag}_reset_extras"], what=f"{tag} reset extras")
        for a in ("vel_command_b", "heading_target", "is_heading_env", "is_standing_env", "time_left", "command_counter"):
            assert_exact(getattr(s, a), g[f"{tag}_reset_{a}"], f"{tag} reset {a}")
        assert_exact(s.metric_error_vel_xy, g[f"{tag}_reset_metric_xy"], f"{tag} reset metric")


def test_sample_draw_order_matches_reference():
    """draw_velocity_samples uses the same uniform_ sequence as CommandTerm._resample + _resample_command."""
    g = load_golden("velocity_command")
    cfg = _vel_cfg(g, "mixed")
    ids = g["mixed_resampled"][0].nonzero().flatten()
    torch.manual_seed(700)
    got = C.draw_velocity_samples(cfg, g["mixed_samples"].shape[1], ids)
    assert_exact(got, g["mixed_samples"][0], "drawn samples")


ANYMAL_WEIGHTS = [1.0, 0.5, -2.0, -0.05, -1.0e-5, -2.5e-7, -0.01, 0.125, -1.0, 0.0, 0.0]
FEET, THIGH, BASE = [13, 14, 15, 16], [5, 6, 7, 8], [0]


def anymal_reward_terms(s):
    """velocity_env_cfg.py:226-254 in declaration order (zero-weight terms evaluate to None: skipped)."""
    cmd, hist = s["command"], s["net_forces_w_history"]
    return [
        T.track_lin_vel_xy_exp(cmd, s["root_lin_vel_b"], SQ), T.track_ang_vel_z_exp(cmd, s["root_ang_vel_b"], SQ),
        T.lin_vel_z_l2(s["root_lin_vel_b"]), T.ang_vel_xy_l2(s["root_ang_vel_b"]), T.joint_torques_l2(s["applied_torque"]),
        T.joint_acc_l2(s["joint_acc"]), T.action_rate_l2(s["action"], s["prev_action"]),
        T.feet_air_time(s["current_contact_time"], s["last_air_time"], cmd, 0.02, 0.5, FEET),
        T.undesired_contacts(hist, 1.0, THIGH), None, None,
    ]


def anymal_obs(s, u):
    """velocity_env_cfg.py:119-143."""
    parts = [
        (s["root_lin_vel_b"], (-0.1, 0.1), None), (s["root_ang_vel_b"], (-0.2, 0.2), None),
        (s["projected_gravity_b"], (-0.05, 0.05), None), (s["command"], None, None),
        (T.joint_pos_rel(s["joint_pos"], s["default_joint_pos"]), (-0.01, 0.01), None),
        (T.joint_vel_rel(s["joint_vel"], s["default_joint_vel"]), (-1.5, 1.5), None), (s["action"], None, None),
        (T.height_scan(s["sensor_pos_w"], s["ray_hits_w"], 0.5), (-0.1, 0.1), (-1.0, 1.0)),
    ]
    out, col = [], 0
    for val, noise, clip in parts:
        w = val.shape[1]
        out.append(T.apply_obs_pipeline(val, u[:, col:col + w], noise, clip))
        col += w
    return MG.observation_concat(out)


def test_managers_anymal():
    g = load_golden("managers_anymal")
    steps, n = g["reward"].shape
    assert g["reward_weights"].tolist() == ANYMAL_WEIGHTS
    episode_sums = torch.zeros(len(ANYMAL_WEIGHTS), n)
    step_reward = torch.zeros(n, len(ANYMAL_WEIGHTS))
    for k in range(steps):
        s = {key: v[k] for key, v in g.items() if v.shape[:1] == (steps,)}
        dones, terminated, truncated, term_dones = MG.termination_compute(
            [T.time_out(s["episode_length_buf"], 1000), T.illegal_contact(s["net_forces_w_history"], 1.0, BASE)], [True, False])
        assert_exact(dones, s["dones"], f"dones step {k}")
        assert_exact(terminated, s["terminated"], f"terminated step {k}")
        assert_exact(truncated, s["time_outs"], f"time_outs step {k}")
        assert_exact(term_dones, s["term_dones"], f"term_dones step {k}")
        reward = MG.reward_compute(anymal_reward_terms(s), ANYMAL_WEIGHTS, 0.02, episode_sums, step_reward)
        assert_close(reward, s["reward"], what=f"reward step {k}")
        assert_close(step_reward, s["step_reward"], what=f"step_reward step {k}")
        assert_close(episode_sums, s["episode_sums"], what=f"episode_sums step {k}")
        assert_close(anymal_obs(s, s["noise_u"]), s["obs"], what=f"obs step {k}")
        assert s["obs"].shape == (n, 235)
        log = MG.reward_reset(episode_sums, dones, 20.0)
        assert_close(log, s["reset_reward_log"], what=f"reset reward log step {k}")
        assert_exact(MG.termination_reset_counts(term_dones, dones), s["reset_term_log"], f"reset termination log step {k}")
        assert_close(episode_sums, s["episode_sums_after_reset"], what=f"episode sums after reset step {k}")
        assert dones.any()


def _pose_cfg(g, tag):
    v = g[f"{tag}_cfg"].tolist()
    return C.PoseCommandCfg(resampling_time_range=(v[0], v[1]), pos_x=(v[2], v[3]), pos_y=(v[4], v[5]), pos_z=(v[6], v[7]),
                            roll=(v[8], v[9]), pitch=(v[10], v[11]), yaw=(v[12], v[13]), make_quat_unique=bool(v[14]))


def test_pose_command():
    """UniformPoseCommand (pose_command.py:92-127 under command_manager.py:151-187) against the reference class."""
    g = load_golden("pose_command")
    for tag in ("reach", "unique"):
        cfg = _pose_cfg(g, tag)
        cmd, tl, cnt = g[f"{tag}_pre_pose_command_b"], g[f"{tag}_pre_time_left"], g[f"{tag}_pre_command_counter"]
        root, body = g[f"{tag}_root_state_w"], g[f"{tag}_body_state_w"][:, 8]
        for k in range(3):
            pe, oe, pw = C.pose_metrics(cmd, root, body[:, :3], body[:, 3:7])   # metrics come before the resample
            assert_close(pe, g[f"{tag}_position_error"][k], atol=1e-6, what=f"{tag} position error step {k}")
            assert_close(oe, g[f"{tag}_orientation_error"][k], atol=2e-6, what=f"{tag} orientation error step {k}")
            cmd, tl, cnt, mask = C.pose_compute(cfg, cmd, tl, cnt, 1 / 30, g[f"{tag}_samples"][k])
            assert_exact(mask, g[f"{tag}_resampled"][k], f"{tag} resample mask step {k}")
            assert_exact(tl, g[f"{tag}_time_left"][k], f"{tag} time_left step {k}")
            assert_exact(cnt, g[f"{tag}_command_counter"][k], f"{tag} counter step {k}")
            assert_close(cmd, g[f"{tag}_pose_command_b"][k], atol=1e-6, what=f"{tag} command step {k}")
            assert mask.sum() > 10
