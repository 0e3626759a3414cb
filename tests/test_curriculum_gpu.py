"""GPU: terrain curriculum (ilx_curriculum_terrain_levels through TerrainImporter / terrain_levels_vel /
CurriculumManager) against the fixture recorded from the reference."""

from types import SimpleNamespace as NS

import pytest
import torch

from conftest import assert_close, assert_exact, load_golden

pytestmark = pytest.mark.gpu


class Scene(dict):
    pass


def _importer(g, dev):
    from isaaclab_amd.terrains import TerrainImporter, TerrainImporterCfg
    from isaaclab_amd.terrains.terrain_importer import generator_cfg

    n = g["init_levels"].shape[0]
    t = TerrainImporter(TerrainImporterCfg(num_envs=n, max_init_terrain_level=3, terrain_generator=generator_cfg((8.0, 8.0))), dev)
    t.configure_env_origins(g["terrain_origins"].numpy())
    assert t.max_terrain_level == 5 and int(t.terrain_levels.max()) <= 3
    assert_exact(t.terrain_types, g["terrain_types"], "terrain types")
    assert_exact(t.env_origins, g["terrain_origins"][t.terrain_levels.cpu(), g["terrain_types"]], "initial placement")
    # continue from the reference's initial draw
    t.terrain_levels.copy_(g["init_levels"])
    t.env_origins.copy_(g["init_env_origins"])
    return t, n


def test_terrain_levels_vel_sequence(dev):
    from isaaclab_amd.envs.mdp.curriculums import CurriculumManager, CurriculumTermCfg, terrain_levels_vel
    from isaaclab_amd.managers import SceneEntityCfg

    g = load_golden("curriculum")
    t, n = _importer(g, dev)
    ids_all = torch.split(g["ids"], g["id_counts"].tolist())
    rnd_all = torch.split(g["random_levels"], g["id_counts"].tolist())
    state = {}
    scene = Scene(robot=NS(data=NS()))
    scene.terrain, scene.env_origins = t, t.env_origins
    env = NS(scene=scene, max_episode_length_s=20.0, command_manager=NS(get_command=lambda name: state["command"]))
    mgr = CurriculumManager(dict(terrain_levels=CurriculumTermCfg(func=terrain_levels_vel, params={"asset_cfg": SceneEntityCfg("robot")}),
                                 off=None), env)
    assert mgr.active_terms == ["terrain_levels"] and mgr.reset() == {}
    for k, (ids, rnd) in enumerate(zip(ids_all, rnd_all)):
        scene["robot"].data.root_state_w = g["root"][k].to(dev).contiguous()
        state["command"] = g["command"][k].to(dev).contiguous()
        mgr._term_cfgs[0].params["_random_levels"] = rnd
        mgr.compute(None if k == 5 else ids.to(dev))
        assert_exact(t.terrain_levels, g["levels"][k], f"levels step {k}")
        assert_exact(t.env_origins, g["env_origins"][k], f"env origins step {k}")
        assert mgr.reset()["Curriculum/terrain_levels"] == pytest.approx(float(g["mean_level"][k]), abs=1e-6)


def test_update_env_origins_with_given_flags(dev):
    """TerrainImporter.update_env_origins alone (flags from the oracle), and its generated replacement levels."""
    from oracle import curriculum as C

    g = load_golden("curriculum")
    t, n = _importer(g, dev)
    ids = torch.split(g["ids"], g["id_counts"].tolist())[0]
    rnd = torch.split(g["random_levels"], g["id_counts"].tolist())[0]
    up, down = C.move_flags(g["root"][0][:, :3], g["init_env_origins"], g["command"][0], ids, 4.0, 20.0)
    t.update_env_origins(ids.to(dev), up, down, _random_levels=rnd)
    assert_exact(t.terrain_levels, g["levels"][0], "levels")
    assert_exact(t.env_origins, g["env_origins"][0], "env origins")
    # every env at the top level moving up: replacement levels come from the generator, spread over all rows
    t.terrain_levels.fill_(4)
    all_up = torch.ones(n, dtype=torch.bool)
    t.update_env_origins(None, all_up, ~all_up)
    lv = t.terrain_levels.cpu()
    assert int(lv.min()) >= 0 and int(lv.max()) <= 4 and len(lv.unique()) == 5
    assert_exact(t.env_origins, g["terrain_origins"][lv, g["terrain_types"]], "origins follow the levels")
    # bottom level moving down stays at zero
    t.terrain_levels.fill_(0)
    t.update_env_origins(None, ~all_up, all_up)
    assert int(t.terrain_levels.abs().max()) == 0


def test_grid_origins(dev):
    from isaaclab_amd.terrains import TerrainImporter, TerrainImporterCfg

    g = load_golden("curriculum")
    for num in (7, 16, 50):
        t = TerrainImporter(TerrainImporterCfg(num_envs=num, env_spacing=2.5), dev)
        t.configure_env_origins(None)
        assert_exact(t.env_origins, g[f"grid_{num}"], f"grid origins for {num} envs")
        t.update_env_origins(None, torch.ones(num, dtype=torch.bool), torch.zeros(num, dtype=torch.bool))  # no-op without sub-terrains
    with pytest.raises(ValueError, match="Environment spacing must be specified"):
        TerrainImporter(TerrainImporterCfg(num_envs=4), dev).configure_env_origins(None)


def test_modify_reward_weight(dev):
    """curriculums.py:20-36 on the fused reward manager: the new weight applies from the step after ``num_steps``."""
    import math

    from isaaclab_amd.envs import mdp
    from isaaclab_amd.managers import RewardManager, RewardTermCfg as RewTerm
    from test_terms_gpu import make_env

    g = load_golden("managers_anymal")
    steps = g["reward"].shape[0]
    s0 = {k: v[0] for k, v in g.items() if v.shape[:1] == (steps,)}
    env = make_env(s0, dev)
    rm = RewardManager(dict(lin_vel_z_l2=RewTerm(mdp.lin_vel_z_l2, {}, -2.0), ang_vel_xy_l2=RewTerm(mdp.ang_vel_xy_l2, {}, -0.05)), env)
    env.reward_manager = rm
    before = rm.compute(dt=0.02).clone()
    parts = rm._step_reward.clone()
    env.common_step_counter = 10
    mdp.modify_reward_weight(env, None, "lin_vel_z_l2", -4.0, num_steps=10)       # not yet: the counter must exceed num_steps
    assert rm.get_term_cfg("lin_vel_z_l2").weight == -2.0
    env.common_step_counter = 11
    plans = rm._plans
    mdp.modify_reward_weight(env, None, "lin_vel_z_l2", -4.0, num_steps=10)
    assert rm.get_term_cfg("lin_vel_z_l2").weight == -4.0
    after = rm.compute(dt=0.02)
    assert_close(after, before + parts[:, 0] * 0.02, rtol=1e-5, atol=1e-6, what="reward with the doubled weight")
    assert rm._plans is not plans                                                  # rebuilt once for the new weight
    kept = rm._plans                                                               # same weight again: the plan is kept
    mdp.modify_reward_weight(env, None, "lin_vel_z_l2", -4.0, num_steps=10)
    assert rm._plans is kept
    with pytest.raises(ValueError, match="not found"):
        mdp.modify_reward_weight(env, None, "nothing", 1.0, num_steps=0)
    assert math.isfinite(float(after.sum()))
