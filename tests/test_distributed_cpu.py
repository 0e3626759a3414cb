"""World-size-2 (gloo, CPU) check of the multi-GPU layout: env shards + ONE all-reduce of the episode statistics
equal the single-process result on the concatenated env set (counts exactly, sums to 1e-5).  The per-rank partial
vectors come from the CPU oracle step here; on GPUs the fused MDP kernel writes the same vector (``log_sums``)."""

import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

N_TOTAL = 96
REWARDS = ["track_lin_vel_xy_exp", "track_ang_vel_z_exp", "lin_vel_z_l2", "ang_vel_xy_l2", "dof_torques_l2", "dof_acc_l2",
           "action_rate_l2", "feet_air_time", "undesired_contacts", "flat_orientation_l2", "dof_pos_limits"]
TERMS = ["time_out", "base_contact"]


def _state_and_mesh():
    from isaaclab_amd import synth
    from isaaclab_amd.tasks import locomotion_velocity as LV
    from isaaclab_amd.utils.string import resolve_matching_names_values

    verts, tris, height_at = synth.rough_terrain_mesh(seed=5, rows=1, cols=1)
    ws, bs = synth.actuator_mlp_weights()
    idx, _, vals = resolve_matching_names_values(LV.ANYMAL_DEFAULT_JOINT_POS, LV.ANYMAL_JOINTS)
    default = torch.zeros(12)
    default[idx] = torch.tensor(vals)
    state = synth.locomotion_state(N_TOTAL, 77, num_joints=12, num_bodies=17, default_joint_pos=default,
                                   feet_ids=[13, 14, 15, 16], height_at=height_at, extent=(8.0, 8.0))
    state["episode_length_buf"][::7] = 999  # time-outs on both shards
    gen = torch.Generator().manual_seed(5)
    noise = torch.rand(N_TOTAL, 235, generator=gen)
    samples = torch.rand(N_TOTAL, 7, generator=gen)
    return verts, tris, ws, bs, default, state, noise, samples


def _slice_state(state, ids):
    out = {}
    for k, v in state.items():
        if not torch.is_tensor(v):
            out[k] = v
        elif k in ("joint_pos", "joint_vel", "net_forces"):   # (substeps, N, ...)
            out[k] = v[:, ids].clone()
        else:
            out[k] = v[ids].clone()
    return out


def _partial(verts, tris, ws, bs, default, state, noise, samples, ids):
    """Packed statistics vector of one shard, in the layout of the fused kernel's ``log_sums``."""
    from oracle import raycast as RC
    from oracle.step_locomotion import AnymalRoughOracle, MAX_LEN, STEP_DT

    ids = torch.as_tensor(list(ids))
    o = AnymalRoughOracle(len(ids), _slice_state(state, ids), ws, bs, default, RC.CMesh(verts, tris))
    ref = o.step(state["raw_action"][ids], noise[ids], samples[ids], samples[ids])
    nd = int(ref["dones"].sum())
    rew = ref["reward_log"].double() * nd * (MAX_LEN * STEP_DT) if nd else torch.zeros(len(REWARDS), dtype=torch.double)
    cl = ref["cmd_log"]
    cmd = torch.tensor([cl.get("error_vel_xy", 0.0), cl.get("error_vel_yaw", 0.0)], dtype=torch.double) * nd
    return torch.cat([rew.float(), ref["term_log"].float(), cmd.float(), torch.tensor([float(nd)])])


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from isaaclab_amd.distributed import all_reduce_log_sums, env_shard

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        args = _state_and_mesh()
        vec = _partial(*args, env_shard(N_TOTAL, rank, world))
        all_reduce_log_sums(vec)
        torch.save(vec, os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_env_shard_ranges():
    from isaaclab_amd.distributed import env_shard

    assert list(env_shard(8, 0, 2)) == [0, 1, 2, 3] and list(env_shard(8, 1, 2)) == [4, 5, 6, 7]
    assert sum(len(env_shard(131072, r, 8)) for r in range(8)) == 131072
    with pytest.raises(ValueError):
        env_shard(10, 0, 4)
    with pytest.raises(ValueError):
        env_shard(8, 2, 2)


def test_two_rank_statistics_equal_single_process(tmp_path):
    from isaaclab_amd.distributed import episode_log
    from oracle.step_locomotion import MAX_LEN, STEP_DT

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f"rank{r}.pt") for r in range(2))
    assert torch.equal(r0, r1), "every rank holds the same reduced vector"
    single = _partial(*_state_and_mesh(), range(N_TOTAL))
    nr = len(REWARDS)
    assert torch.equal(r0[nr:nr + len(TERMS)], single[nr:nr + len(TERMS)]), "termination counts are exact"
    assert r0[-1] == single[-1] and single[-1] > 2, "number of reset envs is exact"
    torch.testing.assert_close(r0, single, rtol=1e-5, atol=1e-6)
    log = episode_log(r0, REWARDS, TERMS, MAX_LEN * STEP_DT)
    assert log["num_resets"] == int(single[-1]) and log["Episode_Termination/time_out"] >= 2
    assert abs(log["Episode_Reward/track_lin_vel_xy_exp"] - float(single[0]) / float(single[-1]) / (MAX_LEN * STEP_DT)) < 1e-6


# ---- Isaac-Repose-Cube-Shadow-Direct: the two success sums of compute_rewards (inhand_manipulation_env.py:424-431)
N_INHAND = 192


def _inhand_step(ids, reduce_stats=None):
    """Oracle step of the envs `ids` of one seeded in-hand state; returns (success sums used, consecutive_successes)."""
    from isaaclab_amd import synth
    from oracle import inhand as IH

    s = synth.inhand_state(N_INHAND, seed=31)
    ids = torch.as_tensor(list(ids))
    shard = {k: (v[ids].clone() if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == N_INHAND else v) for k, v in s.items()}
    o = IH.InHandOracle(IH.InHandCfg(act_moving_average=0.8, max_consecutive_success=50), shard, s["actuated"], s["fingers"])
    gen = torch.Generator().manual_seed(9)
    gs, rs = torch.rand(N_INHAND, 2, generator=gen) * 2 - 1, torch.rand(N_INHAND, 2, generator=gen) * 2 - 1
    o.step(shard["actions"], gs[ids], rs[ids], reduce_stats=reduce_stats)
    return o.last_stats.clone(), o.s["consecutive_successes"].clone()


def _inhand_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from isaaclab_amd.distributed import all_reduce_log_sums, env_shard

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        stats, cons = _inhand_step(env_shard(N_INHAND, rank, world), reduce_stats=all_reduce_log_sums)
        torch.save((stats, cons), os.path.join(out_dir, f"inhand{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_two_rank_success_statistic_equals_single_process(tmp_path):
    """The all-reduce InHandManipulationStep.step issues between ilx_inhand_step and ilx_inhand_finish: with the two sums
    reduced over the shards, every rank's consecutive_successes is the single-process value on all envs."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_inhand_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    (s0, c0), (s1, c1) = (torch.load(tmp_path / f"inhand{r}.pt") for r in range(2))
    single_stats, single_cons = _inhand_step(range(N_INHAND))
    assert torch.equal(s0, s1) and torch.equal(c0, c1), "every rank holds the same sums and the same statistic"
    assert s0[0] == single_stats[0] and single_stats[0] > 4, "the number of resets is exact"
    torch.testing.assert_close(s0, single_stats, rtol=1e-6, atol=0.0)
    torch.testing.assert_close(c0, single_cons, rtol=1e-6, atol=0.0)
    # and it differs from what a rank would log from its own shard alone
    from isaaclab_amd.distributed import env_shard
    _, alone = _inhand_step(env_shard(N_INHAND, 0, 2))
    assert not torch.equal(alone, single_cons)


def _overlap_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from isaaclab_amd.distributed import OverlappedLogReducer

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        red = OverlappedLogReducer(5, "cpu")
        got = []
        for k in range(5):
            buf, totals = red.begin_step(want_totals=True)
            if totals is not None:
                got.append(totals)
            buf += torch.arange(5.0) * (k + 1) + rank      # what the fused kernel would accumulate in step k
            red.end_step()
        got += red.finish()
        torch.save(torch.stack(got), os.path.join(out_dir, f"overlap{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_overlapped_reducer_matches_blocking_all_reduce(tmp_path):
    """Two ranks, five steps: the overlapped reducer hands out every step's global sums, in order, two steps late."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_overlap_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f"overlap{r}.pt") for r in range(2))
    assert torch.equal(r0, r1)
    want = torch.stack([2 * torch.arange(5.0) * (k + 1) + 1.0 for k in range(5)])   # rank 0 adds 0, rank 1 adds 1
    assert torch.equal(r0, want)


def test_overlapped_reducer_single_process():
    from isaaclab_amd.distributed import OverlappedLogReducer

    red = OverlappedLogReducer(3, "cpu")
    seen = []
    for k in range(4):
        buf, totals = red.begin_step(want_totals=True)
        if totals is not None:
            seen.append(totals)
        buf += float(k + 1)
        red.end_step()
    seen += red.finish()
    assert [float(t[0]) for t in seen] == [1.0, 2.0, 3.0, 4.0]
