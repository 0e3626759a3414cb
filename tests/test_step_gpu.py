"""GPU: the complete non-physics env step of the rough-terrain velocity tasks (HIP, through the C ABI) against the
oracle step: Isaac-Velocity-Rough-Anymal-C (ActuatorNetMLP) and Isaac-Velocity-Rough-H1 (DC-motor actuators,
scanner on torso_link), two consecutive steps, all buffers, every env; and the ANYmal step at the benchmark's env
count against the oracle on a sample of its envs."""

import numpy as np
import pytest
import torch

from conftest import assert_close, assert_exact

pytestmark = pytest.mark.gpu


def build(dev, n, seed, rows=2, cols=3, robot="anymal_c"):
    from isaaclab_amd import synth
    from isaaclab_amd.envs.locomotion_step import LocomotionVelocityStep
    from isaaclab_amd.tasks import locomotion_velocity as LV
    from isaaclab_amd.utils.mesh import convert_to_mesh
    from isaaclab_amd.utils.string import resolve_matching_names_values
    from oracle import raycast as RC
    from oracle import step_locomotion as SL

    verts, tris, height_at = synth.rough_terrain_mesh(seed=seed, rows=rows, cols=cols)
    mesh = convert_to_mesh(verts, tris, str(dev))
    if robot == "anymal_c":
        ws, bs = synth.actuator_mlp_weights()
        idx, _, vals = resolve_matching_names_values(LV.ANYMAL_DEFAULT_JOINT_POS, LV.ANYMAL_JOINTS)
        default = torch.zeros(12)
        default[idx] = torch.tensor(vals)
        state = synth.locomotion_state(n, seed, num_joints=12, num_bodies=17, default_joint_pos=default, feet_ids=[13, 14, 15, 16],
                                       height_at=height_at, extent=(rows * 8.0, cols * 8.0))
        env = LocomotionVelocityStep(n, dev, mesh, "anymal_c", actuator_net=dict(weights=ws, biases=bs, activation="softsign"))
        make_oracle = lambda st: SL.AnymalRoughOracle(st["joint_pos"].shape[1], st, ws, bs, default, RC.CMesh(verts, tris))  # noqa: E731
    else:
        assert LV.H1_JOINTS == SL.H1_JOINTS and LV.H1_BODIES == SL.H1_BODIES
        idx, _, vals = resolve_matching_names_values(LV.H1_DEFAULT_JOINT_POS, LV.H1_JOINTS)
        default = torch.zeros(19)
        default[idx] = torch.tensor(vals)
        feet = [i for i, b in enumerate(LV.H1_BODIES) if b.endswith("ankle_link")]
        # joint velocities large enough for the DC-motor envelope to bite on some joints (velocity limit 100 rad/s)
        state = synth.locomotion_state(n, seed, num_joints=19, num_bodies=20, default_joint_pos=default, feet_ids=feet,
                                       height_at=height_at, extent=(rows * 8.0, cols * 8.0), joint_vel_std=40.0, with_bodies=True)
        env = LocomotionVelocityStep(n, dev, mesh, "h1")
        make_oracle = lambda st: SL.H1RoughOracle(st["joint_pos"].shape[1], st, RC.CMesh(verts, tris))  # noqa: E731
    # force a few time-outs so that the reset path runs (base contacts come with the synthetic forces)
    state["episode_length_buf"][:3] = 999
    env.load_state(state)
    return env, make_oracle, state


def kernel_ray_xy(hits):
    """x / y of the world ray starts the kernel used: exact in its hits (vertical rays); NaN where the ray missed."""
    xy = hits[..., :2].clone()
    xy[torch.isinf(hits[..., 2])] = float("nan")
    return xy


def check_step(env, oracle, ref, obs, reward, terminated, time_outs, step, rows=slice(None), robot="anymal_c"):
    """Everything one step produced, for the envs `rows` of the kernel's buffers against the oracle's."""
    cut = lambda t: t.cpu()[rows]  # noqa: E731
    cut1 = lambda t: t.cpu()[:, rows]  # noqa: E731  ([term][env] buffers)
    # decisions: bit exact
    assert_exact(cut(terminated), ref["terminated"], f"terminated step {step}")
    assert_exact(cut(time_outs), ref["time_outs"], f"time_outs step {step}")
    assert_exact(cut1(env.mdp.term_dones), ref["term_dones"], f"term_dones step {step}")
    assert_exact(cut(env.mdp.episode_length_buf), oracle.s["episode_length_buf"], f"episode_length_buf step {step}")
    assert_exact(cut(env.mdp.command_counter), oracle.cmd.command_counter, f"command_counter step {step}")
    assert_exact(cut(env.mdp.is_standing_env), oracle.cmd.is_standing_env, f"is_standing_env step {step}")
    assert_exact(cut(env.mdp.time_left), oracle.cmd.time_left, f"time_left step {step}")
    c = env.contact._data
    assert_exact(cut(env.contact.ordered_history()), oracle.s["net_forces_w_history"], f"contact history step {step}")
    for k in ("current_air_time", "last_air_time", "current_contact_time", "last_contact_time"):
        assert_exact(cut(getattr(c, k)), oracle.s[k], f"{k} step {step}")
    assert_exact(cut(env.data.joint_acc), ref["joint_acc"], f"joint_acc step {step}")
    # ray hits: the oracle cast the kernel's own ray x / y, so every ray compares bit for bit
    hits = cut(env.scanner._data.ray_hits_w)
    assert_exact(hits, ref["ray_hits_w"], f"ray hits step {step}")
    # float terms: 1e-5 relative (atol for values near zero)
    if robot == "anymal_c":
        assert_exact(cut(env.actuator._joint_pos_error_history), oracle.s["pos_error_history"], f"pos error history step {step}")
        assert_close(cut(env.data.computed_torque), ref["computed_torque"], atol=2e-5, what=f"computed torque step {step}")
        assert_close(cut(env.data.applied_torque), ref["applied_torque"], atol=2e-5, what=f"applied torque step {step}")
    else:
        assert_exact(cut(env.data.computed_torque), ref["computed_torque"], f"computed torque step {step}")
        assert_exact(cut(env.data.applied_torque), ref["applied_torque"], f"applied torque step {step}")
        assert (ref["computed_torque"] != ref["applied_torque"]).any(), "the DC-motor envelope clipped something"
    # observations: 1e-5 relative; the height-scan columns to 1e-5 m absolute (rays start 20 m up: one ulp of t is 2e-6 m)
    assert_close(cut(obs)[:, :48], ref["obs"][:, :48], atol=2e-6, what=f"proprioceptive obs step {step}")
    assert_close(cut(obs), ref["obs"], atol=1.2e-5, what=f"obs step {step}")
    assert_close(cut(reward), ref["reward"], atol=1e-6 if robot == "anymal_c" else 2e-5, what=f"reward step {step}")
    assert_close(cut(env.mdp.step_reward), ref["step_reward"], atol=5e-5, rtol=2e-5, what=f"step_reward step {step}")
    assert_close(cut1(env.mdp.episode_sums), ref["episode_sums"], atol=1e-6 if robot == "anymal_c" else 2e-5, what=f"episode_sums step {step}")
    assert_close(cut(env.mdp.command), oracle.cmd.vel_command_b, atol=1e-6, what=f"command step {step}")


def draws(gen, n, obs_dim):
    noise_u = torch.rand(n, obs_dim, generator=gen)
    samples = torch.rand(n, 7, generator=gen) * 2 - 1
    samples[:, 0] = 10.0
    samples[:, 5:] = torch.rand(n, 2, generator=gen)
    return noise_u, samples, samples.flip(0).contiguous()


@pytest.mark.parametrize("robot", ["anymal_c", "h1"])
def test_full_step_matches_oracle(dev, robot):
    n = 512
    env, make_oracle, state = build(dev, n, seed=21 if robot == "anymal_c" else 22, robot=robot)
    oracle = make_oracle(state)
    j, obs_dim = state["raw_action"].shape[1], 48 + 187 if robot == "anymal_c" else 3 * 4 + 3 * 19 + 187
    nr = oracle.num_rewards
    gen = torch.Generator().manual_seed(99)
    for step in range(2):
        raw = state["raw_action"] if step == 0 else torch.rand(n, j, generator=gen) * 2 - 1
        noise_u, samples, reset_samples = draws(gen, n, obs_dim)
        obs, reward, terminated, time_outs = env.step(raw.to(dev), noise_u.to(dev), samples.to(dev), reset_samples.to(dev))
        assert obs.shape == (n, obs_dim)
        xy = kernel_ray_xy(env.scanner._data.ray_hits_w.cpu())
        own_xy = oracle.world_rays()[0][..., :2]
        hit = torch.isfinite(xy)
        # the only difference between the two ray sets: yaw -> (cos, sin) of the half angle in two math libraries
        assert (xy - own_xy)[hit].abs().max() <= 2e-6, "world ray starts (x, y)"
        ref = oracle.step(raw, noise_u, samples, reset_samples, ray_xy=xy)
        check_step(env, oracle, ref, obs, reward, terminated, time_outs, step, robot=robot)
        nd = int(ref["dones"].sum())
        assert nd > 0 and ref["terminated"].any() and ref["time_outs"].any()
        log = env.mdp.log_sums.cpu()
        assert log[-1].item() == nd
        assert_close(log[:nr] / nd / 20.0, ref["reward_log"], rtol=1e-4, atol=1e-6, what="episode reward log")
        assert_exact(log[nr:nr + 2].long(), ref["term_log"], "termination counts")
        env.mdp.log_sums.zero_()


def subset_state(state, ids):
    """The synthetic state of the envs `ids`: per-substep arrays carry the env axis second."""
    out = {}
    for k, v in state.items():
        out[k] = v[:, ids].clone() if k in ("joint_pos", "joint_vel", "net_forces") else v[ids].clone()
    return out


@pytest.mark.parametrize("n", [16384, 131072, 1048576])
def test_anymal_step_at_benchmark_size(dev, n):
    """16,384 envs (the benchmark's shard) and eight times that (what one GPU holds when it takes a whole node's envs):
    catches index-width and grid-size errors that 512 envs cannot.  The oracle runs on 512 of the envs -- the first and
    last 64 and random ones; every per-env buffer of those rows is compared."""
    env, make_oracle, state = build(dev, n, seed=5, rows=3, cols=3)
    gen = torch.Generator().manual_seed(17)
    ids = torch.cat([torch.arange(64), torch.arange(n - 64, n), torch.randperm(n - 128, generator=gen)[:384] + 64]).sort().values
    oracle = make_oracle(subset_state(state, ids))
    for step in range(2):
        raw = state["raw_action"] if step == 0 else torch.rand(n, 12, generator=gen) * 2 - 1
        noise_u, samples, reset_samples = draws(gen, n, 235)
        obs, reward, terminated, time_outs = env.step(raw.to(dev), noise_u.to(dev), samples.to(dev), reset_samples.to(dev))
        xy = kernel_ray_xy(env.scanner._data.ray_hits_w.cpu()[ids])
        ref = oracle.step(raw[ids], noise_u[ids], samples[ids], reset_samples[ids], ray_xy=xy)
        check_step(env, oracle, ref, obs, reward, terminated, time_outs, step, rows=ids)
        # the whole shard: finite outputs, plausible reset counts, the logging vector counts every done env
        assert torch.isfinite(obs).all() and torch.isfinite(reward).all()
        done = (terminated | time_outs).cpu()
        assert env.mdp.log_sums.cpu()[-1].item() == int(done.sum()) > 0
        env.mdp.log_sums.zero_()


def test_step_replayed_from_a_graph_equals_eager(dev):
    """The step recorded in a HIP graph -- the sensor chain as a branch beside the actuator chain, the generator's step
    count kept on the device -- and replayed three times gives, bit for bit, what three eager steps of an identical env
    give: same launches, same arguments, fresh random numbers at every replay."""
    n = 1024
    outs = []
    for graph in (False, True):
        env, _, state = build(dev, n, seed=31)
        raw = state["raw_action"].to(dev)
        if graph:
            env.count_steps_on_device()      # the same (seed, step) keys as the host-side count of the eager env
        env.concurrent_chains = graph
        env.step(raw)                        # first step eagerly in both: per-stream workspaces come into being here
        steps = []
        if graph:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                env.step(raw)
            for _ in range(3):
                g.replay()
                torch.cuda.synchronize()
                steps.append([t.clone() for t in (env.mdp.obs, env.mdp.reward_buf, env.mdp.terminated, env.mdp.command,
                                                  env.contact.ordered_history(), env.actuator._joint_vel_history)])
        else:
            for _ in range(3):
                env.step(raw)
                torch.cuda.synchronize()
                steps.append([t.clone() for t in (env.mdp.obs, env.mdp.reward_buf, env.mdp.terminated, env.mdp.command,
                                                  env.contact.ordered_history(), env.actuator._joint_vel_history)])
        outs.append(steps)
    names = ("obs", "reward", "terminated", "command", "contact history", "velocity history")
    for k in range(3):
        for name, a, b in zip(names, outs[0][k], outs[1][k]):
            assert_exact(b, a, f"{name}, step {k}: graph replay vs eager")
    assert not torch.equal(outs[1][0][0], outs[1][1][0]), "a replay draws new observation noise"


@pytest.mark.parametrize("robot", ["anymal_c", "h1"])
def test_paired_substeps_equal_separate_launches(dev, robot):
    """The step as one chain with {scene.update(k), actuator model of substep k + 1} as one launch
    (ilx_actuator_net_mlp_scene_update, ilx_actuator_pd_scene_update) against the step of separate launches: every buffer
    bit for bit over several steps, at an env count that leaves ragged last workgroups in both halves of the grid."""
    n = 1237
    outs = []
    for paired in (False, True):
        env, _, state = build(dev, n, seed=33, robot=robot)
        env.paired_substeps = paired
        gen = torch.Generator().manual_seed(9)
        trace = []
        for step in range(4):
            raw = torch.rand(n, state["raw_action"].shape[1], generator=gen) * 2 - 1
            noise_u, samples, reset_samples = draws(gen, n, env.mdp.obs_dim)
            out = env.step(raw.to(dev), noise_u.to(dev), samples.to(dev), reset_samples.to(dev))
            sc, c, a = env.scanner, env.contact, env.actuator
            trace.append([t.cpu().clone() for t in (*out, sc._data.ray_hits_w, env.height_scan, sc._timestamp, sc._timestamp_last_update,
                                                    c._timestamp, c._timestamp_last_update, c.ordered_history(), c._data.current_air_time,
                                                    c._data.last_air_time, c._data.current_contact_time, c._data.last_contact_time,
                                                    env.data.joint_acc, env.data._previous_joint_vel, env.data.computed_torque,
                                                    env.data.applied_torque, env.mdp.episode_sums, env.mdp.command,
                                                    *((a._joint_pos_error_history, a._joint_vel_history) if robot == "anymal_c" else ()))])
        outs.append(trace)
    for step, (one, two) in enumerate(zip(*outs)):
        for k, (x, y) in enumerate(zip(one, two)):
            assert_exact(y, x, f"buffer {k}, step {step}")


@pytest.mark.parametrize("robot", ["anymal_c", "h1"])
def test_scene_update_that_prepares_the_scanner_equals_separate_launches(dev, robot):
    """ilx_scene_update_prepare_raycaster + ilx_raycaster_refresh(ILX_RAYS_PREPARED) against ilx_scene_update +
    ilx_raycaster_refresh: every buffer of the step bit for bit over several steps -- including the first step (the pattern
    is not packed yet: separate launches inside), a scanner that is due only every other step (envs that are not due keep
    their hits and clocks), clocks restarted for some envs in between, and the H1's scanner on a body (strided pose)."""
    n = 333
    outs = []
    for fused in (False, True):
        env, _, state = build(dev, n, seed=31, robot=robot)
        env.fuse_scanner_preparation = fused
        env.scanner.cfg.update_period = 0.04            # due every other env step ...
        env.scanner._timestamp[: n // 2] = 0.02         # ... half of the envs on odd steps, half on even ones
        gen = torch.Generator().manual_seed(7)
        j = state["raw_action"].shape[1]
        trace = []
        for step in range(5):
            if step == 3:   # what a reset of some envs does to the scanner's clocks (sensor_base.py:182-195)
                env.scanner.reset(torch.arange(0, n, 7, device=dev))
            raw = torch.rand(n, j, generator=gen) * 2 - 1
            noise_u, samples, reset_samples = draws(gen, n, env.mdp.obs_dim)
            out = env.step(raw.to(dev), noise_u.to(dev), samples.to(dev), reset_samples.to(dev))
            sc, c = env.scanner, env.contact
            trace.append([t.cpu().clone() for t in (*out, sc._data.ray_hits_w, sc._data.pos_w, sc._data.quat_w, env.height_scan,
                                                    sc._timestamp, sc._timestamp_last_update, sc._is_outdated,
                                                    c._timestamp, c._timestamp_last_update, c.ordered_history(),
                                                    c._data.current_air_time, c._data.last_contact_time, env.data.joint_acc)])
        outs.append(trace)
    names = ("obs", "reward", "terminated", "time_outs", "ray hits", "pos_w", "quat_w", "height scan", "scanner clock",
             "scanner last update", "scanner flags", "contact clock", "contact last update", "contact history", "air time",
             "last contact time", "joint_acc")
    for step, (one, two) in enumerate(zip(*outs)):
        for name, a, b in zip(names, one, two):
            assert_exact(b, a, f"{name}, step {step}")
    last = outs[1][-1]
    assert 0 < int((last[8] == last[9]).sum()) < n, "some envs refreshed on the last step, some did not"


def test_joint_action_on_the_first_actuator_launch_equals_its_own_launch(dev):
    """ilx_actuator_net_mlp_joint_action (process_action done by the step's first network launch, which also opens the step)
    against ilx_joint_action_process_begin + ilx_actuator_net_mlp: every buffer of the step bit for bit over several steps,
    with the statistics vector cleared and the device-side step count advanced by that launch."""
    n = 1237
    outs = []
    for rides in (False, True):
        env, _, state = build(dev, n, seed=35)
        env.joint_action_rides = rides
        env.clear_log_sums = True
        env.count_steps_on_device()
        gen = torch.Generator().manual_seed(11)
        trace = []
        for step in range(4):
            raw = torch.rand(n, 12, generator=gen) * 2 - 1
            noise_u, samples, reset_samples = draws(gen, n, env.mdp.obs_dim)
            env.mdp.log_sums.fill_(7.0)     # the step must start it at zero itself
            out = env.step(raw.to(dev), noise_u.to(dev), samples.to(dev), reset_samples.to(dev))
            a = env.actuator
            trace.append([t.cpu().clone() for t in (*out, env.action, env.prev_action, env.data.joint_pos_target, env.data.computed_torque,
                                                    env.data.applied_torque, a._joint_pos_error_history, a._joint_vel_history,
                                                    env.mdp.log_sums, env.mdp._step_counter, env.mdp.episode_sums)])
        outs.append(trace)
    for step, (one, two) in enumerate(zip(*outs)):
        for k, (x, y) in enumerate(zip(one, two)):
            if k == 11:   # the statistics vector: sums of atomic float adds, whose order is not fixed
                assert_close(y, x, rtol=1e-5, atol=1e-6, what=f"log sums, step {step}")
            else:
                assert_exact(y, x, f"buffer {k}, step {step}")
    assert outs[1][-1][12].item() == 3 and outs[1][-1][11][-1].item() > 0, "count advanced; this step's resets were logged"
    prev1, act0 = outs[1][1][5], outs[1][0][4]
    kept = prev1.abs().sum(1) > 0   # envs that the second step did not reset (a reset zeroes the action history)
    assert kept.sum() > n // 2 and torch.equal(prev1[kept], act0[kept]), "prev_action of a step is the action of the step before"
