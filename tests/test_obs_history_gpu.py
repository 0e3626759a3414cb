"""GPU: observation histories in the ObservationManager (observation_manager.py:319-327, 403-420) against the ring-buffer
restatement fed with the group's own un-histored output."""

import pytest
import torch

from conftest import assert_exact, load_golden

pytestmark = pytest.mark.gpu


def test_observation_manager_history():
    from isaaclab_amd.envs import mdp
    from isaaclab_amd.managers import ObservationManager, ObservationTermCfg as ObsTerm
    from oracle import buffers as O
    from test_terms_gpu import make_env

    dev = torch.device("cuda:0")
    g = load_golden("managers_anymal")
    steps = g["reward"].shape[0]
    s0 = {k: v[0] for k, v in g.items() if v.shape[:1] == (steps,)}
    n = s0["joint_pos"].shape[0]
    env = make_env(s0, dev)
    plain = dict(base_lin_vel=ObsTerm(mdp.base_lin_vel, {}), joint_pos=ObsTerm(mdp.joint_pos_rel, {}, scale=2.0),
                 joint_vel=ObsTerm(mdp.joint_vel_rel, {}, clip=(-1.0, 1.0)))
    per_term = dict(base_lin_vel=ObsTerm(mdp.base_lin_vel, {}, history_length=3), joint_pos=ObsTerm(mdp.joint_pos_rel, {}, scale=2.0),
                    joint_vel=ObsTerm(mdp.joint_vel_rel, {}, clip=(-1.0, 1.0), history_length=2))
    whole = dict(base_lin_vel=ObsTerm(mdp.base_lin_vel, {}), joint_pos=ObsTerm(mdp.joint_pos_rel, {}, scale=2.0),
                 joint_vel=ObsTerm(mdp.joint_vel_rel, {}, clip=(-1.0, 1.0)), history_length=2, flatten_history_dim=True)
    om = ObservationManager(dict(plain=plain, per_term=per_term, whole=whole), env)
    assert om.group_obs_dim == {"plain": (27,), "per_term": (9 + 12 + 24,), "whole": (54,)}
    r_lin, r_vel = O.Ring(3, n, 3), O.Ring(2, n, 12)
    w = [O.Ring(2, n, 3), O.Ring(2, n, 12), O.Ring(2, n, 12)]
    for step in range(steps):
        s = {k: v[step] for k, v in g.items() if v.shape[:1] == (steps,)}
        env.scene["robot"].data = make_env(s, dev).scene["robot"].data
        if step == 2:
            ids = [1, 4, n - 1]
            om.reset(ids)
            for ring in (r_lin, r_vel, *w):
                ring.reset(ids)
        raw = om.compute_group("plain").cpu()
        lin, pos, vel = raw[:, :3], raw[:, 3:15], raw[:, 15:]
        r_lin.append(lin), r_vel.append(vel)
        want = torch.cat([r_lin.hist.reshape(n, -1), pos, r_vel.hist.reshape(n, -1)], dim=1)
        assert_exact(om.compute_group("per_term"), want, f"per-term history, step {step}")
        for ring, x in zip(w, (lin, pos, vel)):
            ring.append(x)
        want = torch.cat([ring.hist.reshape(n, -1) for ring in w], dim=1)
        assert_exact(om.compute_group("whole"), want, f"group history, step {step}")


def test_groups_without_concatenation():
    """observation_manager.py:332-335: ``concatenate_terms=False`` hands out the terms one by one; a history that is not
    flattened keeps its (num_envs, history_length, width) shape."""
    from isaaclab_amd.envs import mdp
    from isaaclab_amd.managers import ObservationManager, ObservationTermCfg as ObsTerm
    from oracle import buffers as O
    from test_terms_gpu import make_env

    dev = torch.device("cuda:0")
    g = load_golden("managers_anymal")
    steps = g["reward"].shape[0]
    s0 = {k: v[0] for k, v in g.items() if v.shape[:1] == (steps,)}
    n = s0["joint_pos"].shape[0]
    env = make_env(s0, dev)
    plain = dict(base_lin_vel=ObsTerm(mdp.base_lin_vel, {}), joint_vel=ObsTerm(mdp.joint_vel_rel, {}, scale=0.5))
    split = dict(base_lin_vel=ObsTerm(mdp.base_lin_vel, {}, history_length=3, flatten_history_dim=False),
                 joint_vel=ObsTerm(mdp.joint_vel_rel, {}, scale=0.5), concatenate_terms=False)
    om = ObservationManager(dict(plain=plain, split=split, split_plain=dict(concatenate_terms=False, **plain)), env)
    assert om.group_obs_dim["split"] == [(3, 3), (12,)] and om.group_obs_dim["split_plain"] == [(3,), (12,)]
    ring = O.Ring(3, n, 3)
    for step in range(3):
        s = {k: v[step] for k, v in g.items() if v.shape[:1] == (steps,)}
        env.scene["robot"].data = make_env(s, dev).scene["robot"].data
        raw = om.compute_group("plain").cpu()
        ring.append(raw[:, :3])
        got = om.compute_group("split")
        assert list(got) == ["base_lin_vel", "joint_vel"]
        assert_exact(got["base_lin_vel"], ring.hist, f"un-flattened history, step {step}")
        assert_exact(got["joint_vel"], raw[:, 3:], f"plain term beside it, step {step}")
        got = om.compute_group("split_plain")
        assert_exact(got["base_lin_vel"], raw[:, :3], "term of a split group")
    with pytest.raises(NotImplementedError, match="concatenate_terms=False"):
        ObservationManager(dict(p=dict(t=ObsTerm(mdp.base_lin_vel, {}, history_length=2, flatten_history_dim=False))), env)
