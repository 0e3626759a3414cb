"""GPU: reset / push event terms through the C ABI against the reference functions' fixture and the oracle."""

import types

import pytest
import torch

from conftest import assert_close, assert_exact, load_golden
from oracle import events as OE

pytestmark = pytest.mark.gpu

POSE = {"x": (-0.5, 0.5), "y": (-0.5, 0.5), "yaw": (-3.14, 3.14), "roll": (-0.2, 0.1)}
VEL = {"x": (-0.5, 0.5), "y": (-0.5, 0.5), "z": (-0.1, 0.2), "roll": (-0.5, 0.5), "pitch": (-0.5, 0.5), "yaw": (-0.5, 0.5)}
PUSH = {"x": (-0.5, 0.5), "y": (-0.5, 0.5), "yaw": (-1.0, 0.25)}


class Scene(dict):
    env_origins = None


def _env(g, dev):
    from isaaclab_amd.assets.articulation_data import ArticulationData

    n, j = g["default_joint_pos"].shape
    d = ArticulationData(n, j, 1, dev)
    for k in ("default_root_state", "default_joint_pos", "default_joint_vel", "soft_joint_pos_limits", "soft_joint_vel_limits"):
        getattr(d, k).copy_(g[k])
    d.root_state_w[:, 7:] = g["root_vel_w"].to(dev)
    scene = Scene(robot=types.SimpleNamespace(data=d))
    scene.env_origins = g["env_origins"].to(dev)
    return types.SimpleNamespace(scene=scene), d


def test_events_match_reference(dev):
    from isaaclab_amd.envs import mdp

    g = load_golden("events")
    ids = g["env_ids"]
    others = torch.tensor([i for i in range(g["default_joint_pos"].shape[0]) if i not in ids.tolist()])
    env, d = _env(g, dev)
    before = d.root_state_w.clone()
    # push first (it adds to the current velocity), then the resets
    mdp.push_by_setting_velocity(env, ids.to(dev), PUSH, _u=g["push_u"].to(dev))
    assert_exact(d.root_state_w[ids.to(dev), 7:], g["push_vel"], "pushed root velocity")
    assert_exact(d.root_state_w[others.to(dev)], before[others.to(dev)], "other envs untouched by the push")
    mdp.reset_root_state_uniform(env, ids.to(dev), POSE, VEL, _u=torch.cat([g["root_u_pose"], g["root_u_vel"]], -1).to(dev))
    rows = d.root_state_w[ids.to(dev)]
    assert_exact(rows[:, :3], g["root_pose"][:, :3], "reset root position")
    assert_close(rows[:, 3:7], g["root_pose"][:, 3:], atol=1e-6, what="reset root orientation")
    assert_exact(rows[:, 7:], g["root_vel"], "reset root velocity")
    assert_exact(d.root_state_w[others.to(dev)], before[others.to(dev)], "other envs untouched by the reset")
    for tag, fn, pr, vr in (("scale", mdp.reset_joints_by_scale, (0.5, 1.5), (-1.0, 2.0)),
                            ("offset", mdp.reset_joints_by_offset, (-0.6, 0.6), (-0.5, 0.5))):
        d.joint_pos.zero_(), d.joint_vel.zero_()
        fn(env, ids.to(dev), pr, vr, _u=torch.cat([g[f"{tag}_u_pos"], g[f"{tag}_u_vel"]], -1).to(dev))
        assert_exact(d.joint_pos[ids.to(dev)], g[f"{tag}_joint_pos"], f"{tag} joint positions")
        assert_exact(d.joint_vel[ids.to(dev)], g[f"{tag}_joint_vel"], f"{tag} joint velocities")
        assert not bool(d.joint_pos[others.to(dev)].any()), "other envs untouched"


def test_events_device_rng(dev):
    """Without injected draws: reproducible per (seed, draw), inside the ranges, and different between envs."""
    from isaaclab_amd.envs import mdp
    from isaaclab_amd.utils import rng

    g = load_golden("events")
    env, d = _env(g, dev)
    n = d.root_state_w.shape[0]
    rng.seed_all(77)
    mdp.reset_root_state_uniform(env, None, POSE, VEL)
    first = d.root_state_w.clone()
    rng.seed_all(77)
    mdp.reset_root_state_uniform(env, slice(None), POSE, VEL)
    assert_exact(d.root_state_w, first, "same seed, draw counter restarted -> same reset")
    rng.seed_all(78)
    mdp.reset_root_state_uniform(env, None, POSE, VEL)
    assert not torch.equal(d.root_state_w, first), "another seed draws other values"
    rng.seed_all(77)
    mdp.reset_root_state_uniform(env, None, POSE, VEL)
    mdp.reset_root_state_uniform(env, None, POSE, VEL)
    assert not torch.equal(d.root_state_w, first)
    off = (first[:, :3].cpu() - g["default_root_state"][:, :3] - g["env_origins"])
    assert (off[:, :2].abs() <= 0.5 + 1e-4).all() and (off[:, 2].abs() < 1e-4).all()
    assert off[:, 0].std() > 0.15 and len(torch.unique(off[:, 0])) > n // 2
    # the oracle on the same draws is what the kernel computes: recover u from the x offset and compare one column
    mdp.reset_joints_by_offset(env, None, (-0.1, 0.1), (0.0, 0.0))
    lim = g["soft_joint_pos_limits"]
    jp = d.joint_pos.cpu()
    assert ((jp >= lim[..., 0]) & (jp <= lim[..., 1])).all()
    assert (d.joint_vel.cpu() == torch.clamp(g["default_joint_vel"], -g["soft_joint_vel_limits"], g["soft_joint_vel_limits"])).all()


def test_random_orientation_wrench_and_default_events(dev):
    from isaaclab_amd.envs import mdp
    from isaaclab_amd.managers import SceneEntityCfg

    g = load_golden("events")
    ids = g["env_ids"]
    idd = ids.to(dev)
    n = g["default_joint_pos"].shape[0]
    others = torch.tensor([i for i in range(n) if i not in ids.tolist()]).to(dev)
    env, d = _env(g, dev)
    before = d.root_state_w.clone()
    mdp.reset_root_state_with_random_orientation(env, idd, {"x": (-1.0, 1.0), "z": (0.0, 0.3)}, VEL,
                                                 _u=torch.cat([g["rand_orient_u_pos"], g["rand_orient_u_vel"]], -1).to(dev),
                                                 _z=g["rand_orient_z"].to(dev))
    rows = d.root_state_w[idd]
    assert_exact(rows[:, :3], g["rand_orient_pose"][:, :3], "position")
    assert_close(rows[:, 3:7], g["rand_orient_pose"][:, 3:], atol=1e-6, what="random orientation")
    assert_exact(rows[:, 7:], g["rand_orient_vel"], "velocity")
    assert_exact(d.root_state_w[others], before[others], "other envs untouched")
    # generated draws: unit quaternions, different per env, reproducible for the same (seed, call)
    mdp.reset_root_state_with_random_orientation(env, None, {}, {})
    q = d.root_state_w[:, 3:7].cpu()
    assert_close(q.norm(dim=-1), torch.ones(n), atol=1e-6, what="unit quaternions")
    assert len(torch.unique(q[:, 0])) == n and float(q.mean(0).abs().max()) < 0.35
    # external wrench on three of five bodies
    asset = env.scene["robot"]
    asset.num_bodies = 5
    d.body_state_w = torch.zeros(n, 5, 13, device=dev)
    cfg = SceneEntityCfg("robot")
    cfg.body_ids = [0, 3, 4]
    mdp.apply_external_force_torque(env, idd, (-8.0, 12.0), (-1.5, 0.5), cfg, _u=g["wrench_u"].to(dev))
    assert asset.has_external_wrench
    assert_exact(asset._external_force_b[idd][:, [0, 3, 4]], g["wrench_force"], "forces")
    assert_exact(asset._external_torque_b[idd][:, [0, 3, 4]], g["wrench_torque"], "torques")
    assert float(asset._external_force_b[:, [1, 2]].abs().max()) == 0 and float(asset._external_force_b[others].abs().max()) == 0
    mdp.apply_external_force_torque(env, None, (0.0, 2.0), (0.0, 0.0), SceneEntityCfg("robot"), _u=g["wrench_all_u"].to(dev))
    assert_exact(asset._external_force_b, g["wrench_all_force"], "forces on every body")
    assert_exact(asset._external_torque_b, g["wrench_all_torque"], "zero torques")
    mdp.apply_external_force_torque(env, None, (0.0, 0.0), (0.0, 0.0), SceneEntityCfg("robot"))
    assert not asset.has_external_wrench and float(asset._external_force_b.abs().max()) == 0
    # scene back to its defaults for the listed envs
    env.scene.articulations = {"robot": asset}
    d.joint_pos.fill_(9.0)
    d.joint_vel.fill_(9.0)
    mdp.reset_scene_to_default(env, idd)
    assert_exact(d.root_state_w[idd][:, :7], g["default_pose"], "default pose")
    assert_exact(d.root_state_w[idd][:, 7:], g["default_vel"], "default velocity")
    assert_exact(d.joint_pos[idd], g["default_joint_pos_rows"], "default joint positions")
    assert_exact(d.joint_vel[idd], g["default_joint_vel_rows"], "default joint velocities")
    assert float((d.joint_pos[others] - 9.0).abs().max()) == 0


def test_reset_root_state_from_terrain(dev):
    from isaaclab_amd.envs import mdp

    g = load_golden("events")
    env, d = _env(g, dev)
    n = d.root_state_w.shape[0]
    env.scene.terrain = types.SimpleNamespace(flat_patches={"init_pos": g["terrain_valid_positions"].to(dev)},
                                              terrain_levels=g["terrain_levels"].to(dev), terrain_types=g["terrain_types"].to(dev))
    pose_range = {"roll": (-0.2, 0.1), "yaw": (-3.14, 3.14)}
    mdp.reset_root_state_from_terrain(env, torch.arange(n, device=dev), pose_range, VEL, _u=g["terrain_u"].to(dev),
                                      _patch_ids=g["terrain_picks"].to(dev))
    assert_exact(d.root_state_w[:, :3], g["terrain_pose"][:, :3], "root position on a flat patch")
    assert_close(d.root_state_w[:, 3:7], g["terrain_pose"][:, 3:], atol=1e-6, what="root orientation")
    assert_exact(d.root_state_w[:, 7:], g["terrain_vel"], "root velocity")
    # generated picks and draws: every env lands on one of the patches of its own sub-terrain
    mdp.reset_root_state_from_terrain(env, None, pose_range, VEL)
    own = g["terrain_valid_positions"][g["terrain_levels"], g["terrain_types"]] + g["default_root_state"][:, None, :3]
    gap = (own - d.root_state_w[:, None, :3].cpu()).abs().amax(dim=-1).amin(dim=-1)
    assert float(gap.max()) < 1e-5
    # the reference adds the samples to the default velocity of ALL envs: a subset fails there, and here
    with pytest.raises(RuntimeError, match="must match the size of tensor b"):
        mdp.reset_root_state_from_terrain(env, g["env_ids"].to(dev), pose_range, VEL)
    env.scene.terrain.flat_patches = {"target": None}
    with pytest.raises(ValueError, match="requires valid flat patches under 'init_pos'"):
        mdp.reset_root_state_from_terrain(env, None, pose_range, VEL)


def test_derived_root_quantities_follow_event_writes(dev):
    """root_lin_vel_b / projected_gravity_b read before and after a push or a reset within one sim step: the second read
    sees the new root state (the reference recomputes these at every read, articulation_data.py:502-518)."""
    from isaaclab_amd.assets.articulation import Articulation
    from isaaclab_amd.envs import mdp
    from oracle import mathx as MX

    g = load_golden("events")
    for bare in (True, False):
        env, d = _env(g, dev)
        if bare:                                   # the scene holds the data container itself
            env.scene["robot"] = d
        d.root_state_w[:, 3] = 1.0
        d.root_state_w[:, 4:7] = 0.0
        first = d.root_lin_vel_b.clone()
        mdp.push_by_setting_velocity(env, None, PUSH)
        assert not torch.equal(d.root_lin_vel_b, first), "the pushed velocity is visible at the next read"
        assert_close(d.root_lin_vel_b, MX.quat_rotate_inverse(d.root_state_w[:, 3:7].cpu(), d.root_state_w[:, 7:10].cpu()), atol=1e-6,
                     what="root_lin_vel_b after the push")
        grav = d.projected_gravity_b.clone()
        mdp.reset_root_state_uniform(env, None, POSE, VEL)
        assert not torch.equal(d.projected_gravity_b, grav), "the reset orientation is visible at the next read"
    # the asset's own writers (what reset events call on an Articulation)
    art = Articulation.__new__(Articulation)
    art.data = d
    before = d.root_ang_vel_b.clone()
    art.write_root_velocity_to_sim(torch.ones(d.root_state_w.shape[0], 6, device=dev))
    assert not torch.equal(d.root_ang_vel_b, before)
    heading = d.heading_w.clone()
    pose = d.root_state_w[:, :7].clone()
    pose[:, 3:7] = torch.tensor([0.0, 0.0, 0.0, 1.0], device=dev)   # yaw of pi
    art.write_root_pose_to_sim(pose)
    assert not torch.equal(d.heading_w, heading)
