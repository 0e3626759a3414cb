"""GPU: the fused MDP kernel (ilx_mdp_step through the C ABI) against the reference-generated manager fixtures and
the oracle."""

import math

import pytest
import torch

from conftest import assert_close, assert_exact, load_golden

pytestmark = pytest.mark.gpu


def make_anymal_mdp(dev, n, command=None):
    from isaaclab_amd.envs.fused_mdp import FusedMdp
    from isaaclab_amd.tasks import locomotion_velocity as LV

    terms = LV.anymal_c_rough_terms()
    return FusedMdp(n, dev, num_joints=12, num_bodies=17, num_sensor_bodies=17, history_len=3, num_rays=187, action_dim=12,
                    command_dim=3, step_dt=LV.STEP_DT, max_episode_length=LV.MAX_EPISODE_LENGTH, command=command, **terms)


STATE_FIELDS = ["root_state_w", "joint_pos", "joint_vel", "joint_acc", "applied_torque", "computed_torque",
                "default_joint_pos", "default_joint_vel", "soft_joint_pos_limits", "soft_joint_vel_limits", "body_state_w",
                "net_forces_w_history", "current_air_time", "last_air_time", "current_contact_time", "last_contact_time",
                "ray_hits_w", "sensor_pos_w", "action", "prev_action"]


def bind_state(mdp, s, dev):
    mdp.bind(**{k: s[k].to(dev).contiguous() for k in STATE_FIELDS if k in s})


def test_anymal_managers_match_reference(dev):
    """Three steps of the real Reward/Termination/Observation managers (fixture) vs one fused launch per step."""
    from isaaclab_amd.envs import fused_mdp as FM

    g = load_golden("managers_anymal")
    steps, n = g["reward"].shape
    mdp = make_anymal_mdp(dev, n)
    assert mdp.obs_dim == 235
    for k in range(steps):
        s = {key: v[k] for key, v in g.items() if v.shape[:1] == (steps,)}
        bind_state(mdp, s, dev)
        mdp.bind(command=s["command"].to(dev).contiguous(), noise_u=s["noise_u"].to(dev).contiguous())
        # the fixture's episode_length_buf is the value terminations see, i.e. after the += 1 of the env step
        mdp.episode_length_buf.copy_((s["episode_length_buf"] - 1).to(dev))
        mdp.log_sums.zero_()
        mdp.step(FM.PHASE_TERMINATION | FM.PHASE_REWARD | FM.PHASE_OBSERVATION)
        assert_exact(mdp.terminated, s["terminated"], f"terminated step {k}")
        assert_exact(mdp.time_outs, s["time_outs"], f"time_outs step {k}")
        assert_exact(mdp.reset_buf, s["dones"], f"dones step {k}")
        assert_exact(mdp.term_dones, s["term_dones"], f"term_dones step {k}")
        assert_exact(mdp.episode_length_buf, s["episode_length_buf"], f"episode_length_buf step {k}")
        assert_close(mdp.reward_buf, s["reward"], atol=1e-6, what=f"reward step {k}")
        assert_close(mdp.step_reward, s["step_reward"], atol=2e-5, what=f"step_reward step {k}")
        assert_close(mdp.episode_sums, s["episode_sums"], atol=1e-6, what=f"episode_sums step {k}")
        assert_close(mdp.obs, s["obs"], atol=1e-6, what=f"obs step {k}")
        # reset bookkeeping: per-term sums over the done envs, termination counts, zeroed sums
        mdp.step(FM.PHASE_RESET)
        nd = int(s["dones"].sum())
        log = mdp.log_sums.cpu()
        assert log[-1].item() == nd
        assert_close(log[:11] / nd / 20.0, s["reset_reward_log"], rtol=1e-4, atol=1e-6, what=f"reward log step {k}")
        assert_exact(log[11:13].long(), s["reset_term_log"], f"termination log step {k}")
        assert_close(mdp.episode_sums, s["episode_sums_after_reset"], atol=1e-6, what=f"sums after reset step {k}")
        assert (mdp.episode_length_buf.cpu()[s["dones"]] == 0).all()


@pytest.mark.parametrize("n", [1, 15, 17, 150])
def test_ragged_env_counts_match_the_full_fixture_rows(dev, n):
    """Env counts that are not a multiple of the 16 envs a block owns (a last block with fewer envs, a launch of one
    partial block): every per-env output equals, bit for bit, the same rows of the launch over the whole fixture, and
    the logging sums are those of the first n envs."""
    from isaaclab_amd.envs import fused_mdp as FM

    g = load_golden("managers_anymal")
    steps, full = g["reward"].shape
    s = {key: v[0] for key, v in g.items() if v.shape[:1] == (steps,)}
    outs = []
    for count in (full, n):
        mdp = make_anymal_mdp(dev, count, command=None)
        rows = {k: (v[:count] if v.shape[:1] == (full,) else v) for k, v in s.items()}
        bind_state(mdp, rows, dev)
        mdp.bind(command=rows["command"].to(dev).contiguous(), noise_u=rows["noise_u"].to(dev).contiguous())
        mdp.episode_length_buf.copy_((rows["episode_length_buf"] - 1).to(dev))
        # let some envs of every block time out, so that the reset path of the partial block runs too
        mdp.episode_length_buf[::3] = 10 ** 6
        mdp.log_sums.zero_()
        mdp.step(FM.PHASE_ALL)
        outs.append({k: getattr(mdp, k).cpu().clone() for k in ("obs", "reward_buf", "terminated", "time_outs", "reset_buf",
                                                                 "episode_length_buf", "step_reward")}
                    | {"term_dones": mdp.term_dones.cpu().clone(), "episode_sums": mdp.episode_sums.cpu().clone(),
                       "log": mdp.log_sums.cpu().clone(), "action": mdp.action.cpu().clone()})
    whole, part = outs
    for k in ("obs", "reward_buf", "terminated", "time_outs", "reset_buf", "episode_length_buf", "step_reward", "action"):
        assert torch.equal(part[k], whole[k][:n]), k
    assert torch.equal(part["term_dones"], whole["term_dones"][:, :n]) and torch.equal(part["episode_sums"], whole["episode_sums"][:, :n])
    assert part["log"][-1].item() == int(whole["reset_buf"][:n].sum()) > 0


def test_velocity_command_matches_reference(dev):
    from isaaclab_amd.envs import fused_mdp as FM
    from isaaclab_amd.tasks import locomotion_velocity as LV

    g = load_golden("velocity_command")
    for tag in ("rough", "mixed"):
        lo, hi, stand, head, stiff = g[f"{tag}_cfg"].tolist()
        cmd = dict(LV.VELOCITY_COMMAND, resampling_time_range=(lo, hi), rel_standing_envs=stand, rel_heading_envs=head,
                   heading_control_stiffness=stiff)
        p = lambda key: g[f"{tag}_pre_{key}"]  # noqa: E731
        n = p("time_left").shape[0]
        mdp = make_anymal_mdp(dev, n, command=cmd)
        # root state whose derived body-frame velocities / heading are the fixture's: identity-free route is to hand
        # the kernel a state with unit quaternion and the body-frame velocities as world velocities, plus the yaw
        yaw = p("heading_w")
        quat = torch.stack([torch.cos(yaw / 2), torch.zeros(n), torch.zeros(n), torch.sin(yaw / 2)], -1)
        from oracle import mathx as M
        lin_w = M.quat_rotate(quat, p("root_lin_vel_b"))
        ang_w = M.quat_rotate(quat, p("root_ang_vel_b"))
        rs = torch.cat([torch.zeros(n, 3), quat, lin_w, ang_w], -1)
        mdp.bind(root_state_w=rs.to(dev).contiguous())
        mdp.command.copy_(p("vel_command_b"))
        mdp.heading_target.copy_(p("heading_target"))
        mdp.is_heading_env.copy_(p("is_heading_env"))
        mdp.is_standing_env.copy_(p("is_standing_env"))
        mdp.time_left.copy_(p("time_left"))
        mdp.command_counter.copy_(p("command_counter"))
        mdp.command_metrics[0].copy_(p("metric_xy"))
        mdp.command_metrics[1].copy_(p("metric_yaw"))
        for k in range(g[f"{tag}_samples"].shape[0]):
            mdp.bind(command_samples=g[f"{tag}_samples"][k].to(dev).contiguous())
            mdp.step(FM.PHASE_COMMAND)
            assert_exact(mdp.time_left, g[f"{tag}_time_left"][k], f"{tag} time_left step {k}")
            assert_exact(mdp.command_counter, g[f"{tag}_command_counter"][k], f"{tag} counter step {k}")
            assert_exact(mdp.is_heading_env, g[f"{tag}_is_heading_env"][k], f"{tag} is_heading_env step {k}")
            assert_exact(mdp.is_standing_env, g[f"{tag}_is_standing_env"][k], f"{tag} is_standing_env step {k}")
            assert_exact(mdp.heading_target, g[f"{tag}_heading_target"][k], f"{tag} heading_target step {k}")
            # the yaw/velocities went through a rotate/inverse-rotate round trip: 1e-5 absolute on O(1) values
            assert_close(mdp.command, g[f"{tag}_vel_command_b"][k], atol=2e-5, what=f"{tag} command step {k}")
            assert_close(mdp.command_metrics[0], g[f"{tag}_metric_xy"][k], atol=1e-6, what=f"{tag} metric xy step {k}")
            assert_close(mdp.command_metrics[1], g[f"{tag}_metric_yaw"][k], atol=1e-6, what=f"{tag} metric yaw step {k}")
        # reset of a subset
        ids = g[f"{tag}_reset_ids"]
        mdp.terminated.zero_()
        mdp.time_outs.zero_()
        mdp.time_outs[ids.to(dev)] = True
        mdp.term_dones.zero_()
        before = mdp.command_metrics.cpu().clone()
        mdp.bind(reset_command_samples=g[f"{tag}_reset_samples"].to(dev).contiguous())
        mdp.log_sums.zero_()
        mdp.step(FM.PHASE_RESET)
        for a, name in (("heading_target",) * 2, ("is_heading_env",) * 2, ("is_standing_env",) * 2, ("time_left",) * 2,
                        ("command_counter",) * 2):
            assert_exact(getattr(mdp, a), g[f"{tag}_reset_{name}"], f"{tag} reset {name}")
        # reset envs carry the injected samples exactly; the others still hold the (tolerance-level) values from above
        assert_exact(mdp.command.cpu()[ids], g[f"{tag}_reset_vel_command_b"][ids], f"{tag} reset command (reset envs)")
        assert_close(mdp.command, g[f"{tag}_reset_vel_command_b"], atol=2e-5, what=f"{tag} reset command")
        assert_close(mdp.command_metrics[0], g[f"{tag}_reset_metric_xy"], atol=1e-6, what=f"{tag} reset metric")
        assert (mdp.command_metrics.cpu()[:, ids] == 0).all()
        log = mdp.log_sums.cpu()
        assert_close(log[13:15].double() / len(ids), g[f"{tag}_reset_extras"], rtol=1e-5, atol=1e-6, what=f"{tag} reset extras")
        assert torch.equal(before[:, ~mdp.time_outs.cpu()], mdp.command_metrics.cpu()[:, ~mdp.time_outs.cpu()])


def test_generated_noise_and_samples_are_in_range_and_reproducible(dev):
    """Throughput mode: in-kernel counter-based generator (no injected tensors)."""
    from isaaclab_amd.envs import fused_mdp as FM
    from isaaclab_amd.tasks import locomotion_velocity as LV

    g = load_golden("managers_anymal")
    s = {key: v[0] for key, v in g.items() if v.shape[:1] == (g["reward"].shape[0],)}
    n = s["reward"].shape[0]
    outs = []
    for _ in range(2):
        mdp = make_anymal_mdp(dev, n, command=dict(LV.VELOCITY_COMMAND, resampling_time_range=(0.5, 1.5)))
        bind_state(mdp, s, dev)
        mdp.time_left.fill_(0.01)
        mdp.step(FM.PHASE_COMMAND | FM.PHASE_OBSERVATION, step_index=5)
        outs.append((mdp.obs.cpu(), mdp.command.cpu(), mdp.time_left.cpu(), mdp.heading_target.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    obs, cmd, tl, ht = outs[0]
    clean = make_anymal_mdp(dev, n, command=dict(LV.VELOCITY_COMMAND, resampling_time_range=(0.5, 1.5)))
    bind_state(clean, s, dev)
    clean.bind(noise_u=torch.full((n, 235), 0.5, device=dev))  # zero-mean noise at u = 0.5
    clean.time_left.fill_(0.01)
    clean.step(FM.PHASE_COMMAND | FM.PHASE_OBSERVATION, step_index=5)
    diff = (obs - clean.obs.cpu())
    assert diff[:, 0:3].abs().max() <= 0.1 + 1e-6 and diff[:, 3:6].abs().max() <= 0.2 + 1e-6
    assert diff[:, 9:12].abs().max() == 0, "commands carry no noise"
    assert 0.03 < diff[:, 0:3].std() < 0.08, "uniform(-0.1, 0.1) has std 0.0577"
    assert (tl >= 0.5).all() and (tl <= 1.5).all() and (ht.abs() <= math.pi).all()
    assert (cmd.abs() <= 1.0).all() and cmd.std() > 0.3


@pytest.mark.parametrize("on_device", [False, True])
def test_step_issued_as_two_launches_draws_like_one(dev, on_device):
    """Decisions in one launch, observations in the next, the first holding the generator's step count: bit for bit the
    outputs of the single launch, over several steps, with the count kept by the host or in device memory."""
    from isaaclab_amd.envs import fused_mdp as FM
    from isaaclab_amd.tasks import locomotion_velocity as LV

    g = load_golden("managers_anymal")
    s = {key: v[0] for key, v in g.items() if v.shape[:1] == (g["reward"].shape[0],)}
    n = s["reward"].shape[0]
    outs = []
    for split in (False, True):
        mdp = make_anymal_mdp(dev, n, command=dict(LV.VELOCITY_COMMAND, resampling_time_range=(0.02, 0.06)))
        bind_state(mdp, s, dev)
        mdp.count_steps_on_device(on_device)
        trace = []
        for _ in range(4):
            if split:
                mdp.step(FM.PHASE_ALL & ~FM.PHASE_OBSERVATION, hold_step_count=True)
                mdp.step(FM.PHASE_OBSERVATION)
            else:
                mdp.step(FM.PHASE_ALL)
            trace.append([t.cpu().clone() for t in (mdp.obs, mdp.command, mdp.time_left, mdp.reward_buf, mdp.episode_length_buf)])
        outs.append(trace)
    for one, two in zip(*outs):
        for a, b in zip(one, two):
            assert torch.equal(a, b)
    assert not torch.equal(outs[0][0][0], outs[0][1][0]), "the noise changes from step to step"


def test_begin_step_clears_the_statistics_and_advances_the_count(dev):
    """ilx_mdp_begin_step: one launch for the two things a caller does between env steps."""
    from isaaclab_amd.envs import fused_mdp as FM

    g = load_golden("managers_anymal")
    n = g["reward"].shape[0]
    mdp = make_anymal_mdp(dev, n)
    mdp.log_sums.fill_(3.0)
    mdp.begin_step()                       # host-side count: only the vector is touched on the device
    assert mdp.log_sums.abs().max().item() == 0.0 and mdp.desc.step_index == 1
    mdp.count_steps_on_device(start=-1)
    mdp.log_sums.fill_(2.0)
    mdp.begin_step(clear_log_sums=False)
    mdp.begin_step(clear_log_sums=False)
    assert mdp._step_counter.item() == 1 and mdp.log_sums.min().item() == 2.0 and mdp.desc.step_index == 1
    mdp.begin_step()
    assert mdp._step_counter.item() == 2 and mdp.log_sums.abs().max().item() == 0.0


def test_joint_action_launch_opens_the_step(dev):
    """ilx_joint_action_process_begin = ilx_joint_action_process + ilx_mdp_begin_step in one launch: same action buffers,
    statistics vector cleared, device-side count advanced; ragged sizes (the opening belongs to the first workgroup)."""
    from isaaclab_amd import _abi

    for n, a in ((1, 1), (5, 3), (300, 12), (4096, 7)):
        gen = torch.Generator().manual_seed(n)
        raw, act, default = (torch.randn(n, a, generator=gen).to(dev) for _ in range(3))
        outs = []
        for fused in (False, True):
            action, prev, target = act.clone(), torch.zeros(n, a, device=dev), torch.zeros(n, a, device=dev)
            log = torch.full((17,), 5.0, device=dev)
            count = torch.full((1,), 41, dtype=torch.int64, device=dev)
            args = (_abi.ptr(raw), _abi.ptr(action), _abi.ptr(prev), None, 0.5, _abi.ptr(default), 0.0, None, _abi.ptr(target), n, a)
            if fused:
                _abi.check(_abi.joint_action_process_begin(*args, _abi.ptr(log), 17, _abi.ptr(count), _abi.stream()), "begin")
            else:
                _abi.check(_abi.joint_action_process(*args, _abi.stream()), "ilx_joint_action_process")
                _abi.check(_abi.lib.ilx_mdp_begin_step(_abi.ptr(log), 17, _abi.ptr(count), _abi.stream()), "ilx_mdp_begin_step")
            outs.append([t.cpu() for t in (action, prev, target, log, count)])
        for x, y in zip(*outs):
            assert torch.equal(x, y)
        assert outs[1][3].abs().max().item() == 0.0 and outs[1][4].item() == 42
        assert torch.equal(outs[1][1], act.cpu()) and torch.equal(outs[1][0], raw.cpu())
    # the opening alone can be left out: no vector, no count
    action, prev, target = act.clone(), torch.zeros_like(act), torch.zeros_like(act)
    _abi.check(_abi.joint_action_process_begin(_abi.ptr(raw), _abi.ptr(action), _abi.ptr(prev), None, 0.5, _abi.ptr(default), 0.0, None,
                                               _abi.ptr(target), n, a, None, 0, None, _abi.stream()), "begin")
    assert torch.equal(target.cpu(), outs[1][2])


def test_pose_command_matches_reference(dev):
    """UniformPoseCommand through the fused kernel's COMMAND / RESET phases against the reference class."""
    from isaaclab_amd.envs import fused_mdp as FM

    g = load_golden("pose_command")
    for tag in ("reach", "unique"):
        v = g[f"{tag}_cfg"].tolist()
        cmd_cfg = dict(kind="pose", resampling_time_range=(v[0], v[1]), ranges=[(v[2], v[3]), (v[4], v[5]), (v[6], v[7]), (v[8], v[9]),
                                                                              (v[10], v[11]), (v[12], v[13])],
                       body_index=8, make_quat_unique=bool(v[14]))
        n = g[f"{tag}_pre_time_left"].shape[0]
        mdp = FM.FusedMdp(n, dev, num_joints=7, num_bodies=11, command_dim=7, step_dt=1 / 30, command=cmd_cfg)
        mdp.bind(root_state_w=g[f"{tag}_root_state_w"].to(dev).contiguous(), body_state_w=g[f"{tag}_body_state_w"].to(dev).contiguous())
        mdp.command.copy_(g[f"{tag}_pre_pose_command_b"])
        mdp.time_left.copy_(g[f"{tag}_pre_time_left"])
        mdp.command_counter.copy_(g[f"{tag}_pre_command_counter"])
        for k in range(3):
            mdp.bind(command_samples=g[f"{tag}_samples"][k].to(dev).contiguous())
            mdp.step(FM.PHASE_COMMAND)
            assert_exact(mdp.time_left, g[f"{tag}_time_left"][k], f"{tag} time_left step {k}")
            assert_exact(mdp.command_counter, g[f"{tag}_command_counter"][k], f"{tag} counter step {k}")
            assert_close(mdp.command, g[f"{tag}_pose_command_b"][k], atol=1e-6, what=f"{tag} command step {k}")
            assert_close(mdp.command_metrics[0], g[f"{tag}_position_error"][k], atol=1e-6, what=f"{tag} position error step {k}")
            assert_close(mdp.command_metrics[1], g[f"{tag}_orientation_error"][k], atol=3e-6, what=f"{tag} orientation error step {k}")
        ids = g[f"{tag}_reset_ids"]
        mdp.time_outs.zero_()
        mdp.terminated.zero_()
        mdp.time_outs[ids.to(dev)] = True
        mdp.bind(reset_command_samples=g[f"{tag}_reset_samples"].to(dev).contiguous())
        mdp.log_sums.zero_()
        mdp.step(FM.PHASE_RESET)
        assert_exact(mdp.time_left, g[f"{tag}_reset_time_left"], f"{tag} reset time_left")
        assert_exact(mdp.command_counter, g[f"{tag}_reset_command_counter"], f"{tag} reset counter")
        assert_close(mdp.command, g[f"{tag}_reset_pose_command_b"], atol=1e-6, what=f"{tag} reset command")
        assert_close(mdp.command_metrics[0], g[f"{tag}_reset_position_error"], atol=1e-6, what=f"{tag} reset metric")
        log = mdp.log_sums.cpu()
        assert_close(log[0:2].double() / len(ids), g[f"{tag}_reset_extras"], rtol=1e-5, atol=1e-6, what=f"{tag} reset extras")


def _managers_case(dev, tag, mdp, n_rewards, n_terms, max_len_s, obs_atol=1e-6):
    """Three steps of the reference's real managers (fixture) vs one fused launch per step, plus the reset bookkeeping."""
    from isaaclab_amd.envs import fused_mdp as FM

    g = load_golden(f"managers_{tag}")
    steps = g["reward"].shape[0]
    for k in range(steps):
        s = {key: v[k] for key, v in g.items() if v.shape[:1] == (steps,)}
        bind_state(mdp, s, dev)
        mdp.bind(command=s["command"].to(dev).contiguous(), noise_u=s["noise_u"].to(dev).contiguous())
        mdp.episode_length_buf.copy_((s["episode_length_buf"] - 1).to(dev))
        mdp.log_sums.zero_()
        mdp.step(FM.PHASE_TERMINATION | FM.PHASE_REWARD | FM.PHASE_OBSERVATION)
        assert_exact(mdp.terminated, s["terminated"], f"{tag} terminated step {k}")
        assert_exact(mdp.time_outs, s["time_outs"], f"{tag} time_outs step {k}")
        assert_exact(mdp.term_dones, s["term_dones"], f"{tag} term_dones step {k}")
        assert_close(mdp.reward_buf, s["reward"], atol=2e-6, what=f"{tag} reward step {k}")
        assert_close(mdp.step_reward, s["step_reward"], atol=5e-5, what=f"{tag} step_reward step {k}")
        assert_close(mdp.episode_sums, s["episode_sums"], atol=2e-6, what=f"{tag} episode_sums step {k}")
        assert_close(mdp.obs, s["obs"], atol=obs_atol, what=f"{tag} obs step {k}")
        mdp.step(FM.PHASE_RESET)
        nd = int(s["dones"].sum())
        log = mdp.log_sums.cpu()
        assert log[-1].item() == nd and nd > 0
        assert_close(log[:n_rewards] / nd / max_len_s, s["reset_reward_log"], rtol=1e-4, atol=1e-6, what=f"{tag} reward log step {k}")
        assert_exact(log[n_rewards:n_rewards + n_terms].long(), s["reset_term_log"], f"{tag} termination log step {k}")
        assert_close(mdp.episode_sums, s["episode_sums_after_reset"], atol=2e-6, what=f"{tag} sums after reset step {k}")
    return g


def test_h1_managers_match_reference(dev):
    """Unitree H1 rough (h1/rough_env_cfg.py): biped air time, feet slide, yaw-frame tracking, termination penalty,
    joint deviation groups, torso contact termination."""
    from isaaclab_amd.envs.fused_mdp import FusedMdp
    from isaaclab_amd.tasks import locomotion_velocity as LV

    mdp = FusedMdp(128, dev, num_joints=19, num_bodies=20, num_sensor_bodies=20, history_len=3, num_rays=187, action_dim=19,
                   command_dim=3, step_dt=LV.STEP_DT, max_episode_length=LV.MAX_EPISODE_LENGTH, **LV.h1_rough_terms())
    assert mdp.obs_dim == 256
    g = _managers_case(dev, "h1", mdp, 14, 2, 20.0)
    assert 0.1 < g["terminated"].float().mean() < 0.95


def test_cartpole_managers_match_reference(dev):
    from isaaclab_amd.envs.fused_mdp import FusedMdp
    from isaaclab_amd.tasks import cartpole as CP

    assert CP.MAX_EPISODE_LENGTH == 300
    mdp = FusedMdp(256, dev, num_joints=2, num_bodies=3, action_dim=1, command_dim=3, step_dt=CP.STEP_DT,
                   max_episode_length=CP.MAX_EPISODE_LENGTH, **CP.cartpole_terms())
    assert mdp.obs_dim == 4
    g = _managers_case(dev, "cartpole", mdp, 5, 2, CP.MAX_EPISODE_LENGTH * CP.STEP_DT)
    assert g["terminated"].any() and g["time_outs"].any()


def test_reach_managers_match_reference(dev):
    from isaaclab_amd.envs.fused_mdp import FusedMdp
    from isaaclab_amd.tasks import reach as RE

    assert RE.MAX_EPISODE_LENGTH == 360
    mdp = FusedMdp(256, dev, num_joints=9, num_bodies=11, action_dim=7, command_dim=7, step_dt=RE.STEP_DT,
                   max_episode_length=RE.MAX_EPISODE_LENGTH, **RE.reach_terms())
    assert mdp.obs_dim == 32
    _managers_case(dev, "reach", mdp, 5, 1, RE.MAX_EPISODE_LENGTH * RE.STEP_DT, obs_atol=2e-6)
