"""GPU: the in-hand manipulation (Shadow hand) non-physics step against the reference's outputs and the oracle."""

import pytest
import torch

from conftest import assert_close, assert_exact, load_golden
from oracle import inhand as IH
from test_oracle_inhand import VARIANTS, load_case

pytestmark = pytest.mark.gpu


def make_env(dev, s, ema, max_succ, obs_dim=157):
    from isaaclab_amd.envs.inhand_step import InHandManipulationCfg, InHandManipulationStep

    n = s["joint_pos"].shape[0]
    cfg = InHandManipulationCfg(act_moving_average=ema, max_consecutive_success=max_succ)
    env = InHandManipulationStep(cfg, n, dev, num_joints=24, num_bodies=26, actuated_dof_indices=[int(i) for i in s["actuated"]],
                                 finger_bodies=[int(i) for i in s["fingers"]], lower_limits=s["lower"], upper_limits=s["upper"],
                                 in_hand_pos=s["in_hand_pos"], env_origins=s["env_origins"])
    assert env.max_episode_length == 600 and env.obs_dim == obs_dim
    for k in ("cur_targets", "prev_targets", "goal_rot", "successes", "consecutive_successes", "episode_length_buf"):
        getattr(env, k).copy_(s[k])
    env.set_sim_state(s["joint_pos"].to(dev), s["joint_vel"].to(dev), s["body_state"].to(dev), s["object_state"].to(dev))
    return env


def test_inhand_step_matches_reference(dev):
    g = load_golden("inhand")
    for tag, (ema, max_succ) in VARIANTS.items():
        s = load_case(g, tag)
        env = make_env(dev, s, ema, max_succ)
        obs, reward, terminated, time_outs = env.step(s["actions"].to(dev), g[f"{tag}_goal_samples"].to(dev),
                                                      g[f"{tag}_reset_goal_samples"].to(dev))
        assert_exact(env.cur_targets, g[f"{tag}_cur_targets"], f"{tag} cur_targets")
        assert_exact(env.prev_targets, g[f"{tag}_prev_targets"], f"{tag} prev_targets")
        assert_exact(terminated, g[f"{tag}_terminated"], f"{tag} terminated")
        assert_exact(time_outs, g[f"{tag}_time_outs"], f"{tag} time_outs")
        assert_exact(env.successes, g[f"{tag}_successes"], f"{tag} successes")
        assert_exact(env.episode_length_buf, g[f"{tag}_episode_length_buf"], f"{tag} episode_length_buf")
        assert not env.reset_goal_buf.any()
        assert_close(reward, g[f"{tag}_reward"], rtol=1e-5, atol=1e-5, what=f"{tag} reward")
        assert_close(env.consecutive_successes, g[f"{tag}_consecutive_successes"], rtol=1e-5, what=f"{tag} consecutive successes")
        assert_close(env.goal_rot, g[f"{tag}_goal_rot"], rtol=1e-5, atol=1e-6, what=f"{tag} goal_rot")
        assert_close(obs, g[f"{tag}_obs"], rtol=1e-5, atol=2e-6, what=f"{tag} observations")
        assert (env.stats == 0).all(), "the partial sums are cleared for the next step"


def test_inhand_second_step_and_generator(dev):
    """Two consecutive steps against the oracle (state carried over), then the in-kernel generator: new goals are unit
    quaternions and reproducible."""
    g = load_golden("inhand")
    s = load_case(g, "shadow_ema")
    env = make_env(dev, s, 0.8, 50)
    cfg = IH.InHandCfg(act_moving_average=0.8, max_consecutive_success=50)
    o = IH.InHandOracle(cfg, s, s["actuated"].tolist(), s["fingers"].tolist())
    gen = torch.Generator().manual_seed(3)
    for step in range(2):
        actions = s["actions"] if step == 0 else torch.rand_like(s["actions"]) * 2 - 1
        gs, rs = torch.rand(512, 2, generator=gen) * 2 - 1, torch.rand(512, 2, generator=gen) * 2 - 1
        r = o.step(actions, gs, rs)
        obs, reward, terminated, time_outs = env.step(actions.to(dev), gs.to(dev), rs.to(dev))
        assert_exact(terminated, r["terminated"], f"terminated step {step}")
        assert_exact(time_outs, r["time_outs"], f"time_outs step {step}")
        assert_exact(env.successes, o.s["successes"], f"successes step {step}")
        assert_exact(env.cur_targets, o.s["cur_targets"], f"targets step {step}")
        assert_close(reward, r["reward"], rtol=1e-5, atol=1e-5, what=f"reward step {step}")
        assert_close(obs, r["obs"], rtol=1e-5, atol=2e-6, what=f"obs step {step}")
        assert_close(env.consecutive_successes, o.s["consecutive_successes"], rtol=1e-5, what=f"consecutive successes step {step}")
    outs = []
    for _ in range(2):
        e2 = make_env(dev, s, 1.0, 0)
        e2.step(s["actions"].to(dev))
        outs.append(e2.goal_rot.cpu())
    assert torch.equal(outs[0], outs[1])
    changed = (outs[0] != s["goal_rot"]).any(dim=1)
    assert changed.sum() > 10
    assert_close(outs[0].norm(dim=1), torch.ones(512), rtol=1e-5, what="generated goals are unit quaternions")


def test_inhand_step_at_benchmark_size(dev):
    """8,192 envs (one GPU's shard of the configuration): every env against the oracle, two steps; the global
    success statistic is a sum over all of them."""
    from isaaclab_amd import synth

    n = 8192
    s = synth.inhand_state(n, seed=77)
    s["actuated"], s["fingers"] = torch.tensor(s["actuated"]), torch.tensor(s["fingers"])
    env = make_env(dev, s, 0.8, 50)
    cfg = IH.InHandCfg(act_moving_average=0.8, max_consecutive_success=50)
    o = IH.InHandOracle(cfg, s, s["actuated"].tolist(), s["fingers"].tolist())
    gen = torch.Generator().manual_seed(4)
    for step in range(2):
        actions = s["actions"] if step == 0 else torch.rand(n, s["actions"].shape[1], generator=gen) * 2 - 1
        gs, rs = torch.rand(n, 2, generator=gen) * 2 - 1, torch.rand(n, 2, generator=gen) * 2 - 1
        r = o.step(actions, gs, rs)
        obs, reward, terminated, time_outs = env.step(actions.to(dev), gs.to(dev), rs.to(dev))
        assert_exact(terminated, r["terminated"], f"terminated step {step}")
        assert_exact(time_outs, r["time_outs"], f"time_outs step {step}")
        assert_exact(env.successes, o.s["successes"], f"successes step {step}")
        assert_exact(env.cur_targets, o.s["cur_targets"], f"targets step {step}")
        assert_exact(env.episode_length_buf, o.s["episode_length_buf"], f"episode_length_buf step {step}")
        assert_close(reward, r["reward"], rtol=1e-5, atol=1e-5, what=f"reward step {step}")
        assert_close(obs, r["obs"], rtol=1e-5, atol=2e-6, what=f"obs step {step}")
        assert_close(env.consecutive_successes, o.s["consecutive_successes"], rtol=1e-5, what=f"consecutive successes step {step}")
        assert r["terminated"].sum() > 50 and (step > 0 or r["goal_resets"].sum() > 20)


@pytest.mark.parametrize("n, fingers", [(1000, None), (16 * 7 + 3, None), (1000, [3, 7, 11, 15, 19, 22, 25])],
                         ids=["ragged-last-block", "small", "seven-fingertips"])
def test_inhand_step_other_shapes(dev, n, fingers):
    """Env counts that leave the last 16-env workgroup partly empty (its spare threads read a clamped env and store
    nothing), and a hand with more fingertips than the unrolled kernel takes (the general kernel runs): two steps against
    the oracle."""
    from isaaclab_amd import synth

    s = synth.inhand_state(n, seed=12)
    if fingers is not None:
        s["fingers"] = fingers
    s["actuated"], s["fingers"] = torch.tensor(s["actuated"]), torch.tensor(s["fingers"])
    obs_dim = 2 * 24 + 24 + 13 * len(s["fingers"]) + len(s["actuated"])
    env = make_env(dev, s, 0.8, 50, obs_dim)
    o = IH.InHandOracle(IH.InHandCfg(act_moving_average=0.8, max_consecutive_success=50), s, s["actuated"].tolist(), s["fingers"].tolist())
    gen = torch.Generator().manual_seed(5)
    for step in range(2):
        actions = s["actions"] if step == 0 else torch.rand(n, s["actions"].shape[1], generator=gen) * 2 - 1
        gs, rs = torch.rand(n, 2, generator=gen) * 2 - 1, torch.rand(n, 2, generator=gen) * 2 - 1
        r = o.step(actions, gs, rs)
        obs, reward, terminated, time_outs = env.step(actions.to(dev), gs.to(dev), rs.to(dev))
        assert obs.shape == (n, obs_dim)
        assert_exact(terminated, r["terminated"], f"terminated step {step}")
        assert_exact(time_outs, r["time_outs"], f"time_outs step {step}")
        assert_exact(env.successes, o.s["successes"], f"successes step {step}")
        assert_exact(env.episode_length_buf, o.s["episode_length_buf"], f"episode_length_buf step {step}")
        assert_close(reward, r["reward"], rtol=1e-5, atol=1e-5, what=f"reward step {step}")
        assert_close(obs, r["obs"], rtol=1e-5, atol=2e-6, what=f"obs step {step}")
        assert_close(env.goal_rot, o.s["goal_rot"], rtol=1e-5, atol=1e-6, what=f"goal_rot step {step}")
        assert_close(env.consecutive_successes, o.s["consecutive_successes"], rtol=1e-5, what=f"consecutive successes step {step}")


def test_inhand_step_replayed_from_a_graph_equals_eager(dev):
    """The step recorded in a HIP graph with the generator's step count in device memory (ilx_inhand_finish advances it)
    against the same steps issued launch by launch with the host-side count: every replay draws the numbers of its own
    step -- goals, successes, rewards and observations bit for bit over several steps."""
    g = load_golden("inhand")
    s = load_case(g, "shadow_ema")
    actions = s["actions"].to(dev)
    outs = []
    for graph in (False, True):
        env = make_env(dev, s, 0.8, 50)
        env.episode_length_buf[::3] = 598      # time-outs on the first steps: the reset draws run
        trace = []
        if graph:
            env.count_steps_on_device()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            gr = torch.cuda.CUDAGraph()
            state = [t.clone() for t in (env.cur_targets, env.prev_targets, env.goal_rot, env.successes, env.consecutive_successes,
                                         env.episode_length_buf, env.reset_goal_buf, env._step_counter)]
            with torch.cuda.stream(side):
                env.step(actions)               # creates whatever is created lazily
            side.synchronize()
            for t, v in zip((env.cur_targets, env.prev_targets, env.goal_rot, env.successes, env.consecutive_successes,
                             env.episode_length_buf, env.reset_goal_buf, env._step_counter), state):
                t.copy_(v)                      # ... and is undone: the replays start from the same state as the eager run
            with torch.cuda.graph(gr, stream=side):
                env.step(actions)
        for step in range(5):
            if graph:
                gr.replay()
            else:
                env.step(actions)
            torch.cuda.synchronize()
            trace.append([t.cpu().clone() for t in (env.obs, env.reward, env.goal_rot, env.successes, env.episode_length_buf,
                                                    env.reset_terminated, env.reset_time_outs, env.consecutive_successes)])
        outs.append(trace)
    for step, (one, two) in enumerate(zip(*outs)):
        for k, (x, y) in enumerate(zip(one, two)):
            assert_exact(y, x, f"buffer {k}, step {step}")
    assert not torch.equal(outs[1][0][2], outs[1][1][2]), "new goals are drawn from step to step"


def test_inhand_sums_per_workgroup_equal_the_adds_to_one_address(dev):
    """The step's two global sums left as one pair per workgroup and added up by the finish launch (ilx_inhand::block_sums)
    against every workgroup adding its pair to `stats` itself: all buffers bit for bit over three steps (the sums are sums of
    small whole numbers: exact in any order)."""
    from isaaclab_amd import synth

    outs = []
    for per_workgroup in (True, False):
        s = synth.inhand_state(3000, seed=21)
        s["actuated"], s["fingers"] = torch.tensor(s["actuated"]), torch.tensor(s["fingers"])
        env = make_env(dev, s, 0.8, 50)
        env.sums_per_workgroup = per_workgroup
        env.episode_length_buf[::3] = 598
        trace = []
        for step in range(3):
            env.step(s["actions"].to(dev))
            torch.cuda.synchronize()
            assert (env.desc.block_sums is not None) == per_workgroup
            trace.append([t.cpu().clone() for t in (env.obs, env.reward, env.goal_rot, env.successes, env.consecutive_successes, env.stats)])
        outs.append(trace)
    assert outs[0][2][4].item() != 0.0, "resets happened: the statistic moved"
    for step, (a, b) in enumerate(zip(*outs)):
        for k, (x, y) in enumerate(zip(a, b)):
            assert_exact(x, y, f"buffer {k}, step {step}")
