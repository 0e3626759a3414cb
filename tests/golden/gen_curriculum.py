"""Golden vectors for the terrain curriculum, produced by the REFERENCE ``terrain_levels_vel``
(``isaaclab_tasks/.../velocity/mdp/curriculums.py``) on the REFERENCE ``TerrainImporter``'s origin methods
(``isaaclab/terrains/terrain_importer.py``; the object is created without its constructor, which needs the simulator).
The ``randint_like`` draws are re-drawn after re-seeding and stored.  Writes tests/golden/curriculum.npz."""

import os
import sys
from types import SimpleNamespace as NS

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refload as R  # noqa: E402

m = R.bootstrap()
R.placeholder("isaaclab.terrains.terrain_generator")
R.placeholder("isaaclab.terrains.terrain_importer_cfg")
R.placeholder("isaaclab.terrains.utils")
R.placeholder("isaaclab.utils.warp")
R.placeholder("isaaclab.utils.timer")
R.placeholder("isaaclab.sim")
ti = R.load("isaaclab.terrains.terrain_importer", f"{R.L}/terrains/terrain_importer.py")
sys.modules["isaaclab.terrains"].TerrainImporter = ti.TerrainImporter
cur = R.load("isaaclab_tasks.manager_based.locomotion.velocity.mdp.curriculums",
             f"{R.T}/manager_based/locomotion/velocity/mdp/curriculums.py")

out = {}
g = torch.Generator().manual_seed(9107)
n, rows, cols, tile = 240, 5, 6, 8.0
terrain = object.__new__(ti.TerrainImporter)
terrain.cfg = NS(num_envs=n, max_init_terrain_level=3, env_spacing=2.5, terrain_generator=NS(size=(tile, tile)))
terrain.device = "cpu"
origins = torch.zeros(rows, cols, 3)
origins[..., 0] = (torch.arange(rows)[:, None] + 0.5) * tile
origins[..., 1] = (torch.arange(cols)[None, :] + 0.5) * tile
origins[..., 2] = torch.rand(rows, cols, generator=g) * 0.3
torch.manual_seed(31)
terrain.configure_env_origins(origins.numpy())
out["terrain_origins"], out["terrain_types"] = origins, terrain.terrain_types.clone()
out["init_levels"], out["init_env_origins"] = terrain.terrain_levels.clone(), terrain.env_origins.clone()
assert int(terrain.terrain_levels.max()) <= 3

mels = 20.0
steps = []
for k in range(6):
    ids = torch.randperm(n, generator=g)[: 60 + 10 * k].sort().values if k < 5 else slice(None)
    count = n if isinstance(ids, slice) else len(ids)
    # robots somewhere between "did not move" and "left the sub-terrain"; a few exactly on the thresholds
    walk = torch.rand(n, generator=g) * 6.0
    ang = torch.rand(n, generator=g) * 6.2832
    root = torch.zeros(n, 13)
    root[:, 0] = terrain.env_origins[:, 0] + walk * torch.cos(ang)
    root[:, 1] = terrain.env_origins[:, 1] + walk * torch.sin(ang)
    root[:5, 0], root[:5, 1] = terrain.env_origins[:5, 0] + 4.0, terrain.env_origins[:5, 1]   # distance == size / 2 -> not up
    command = torch.rand(n, 3, generator=g) * 2 - 1
    command[5:10, :2] = 0.0   # zero command: never "less than required"
    env = NS(scene=NS(terrain=terrain, env_origins=terrain.env_origins, __getitem__=None), max_episode_length_s=mels,
             command_manager=NS(get_command=lambda name: command))

    class Scene(dict):
        pass

    scene = Scene(robot=NS(data=NS(root_pos_w=root[:, :3])))
    scene.terrain, scene.env_origins = terrain, terrain.env_origins
    env.scene = scene
    levels_before = terrain.terrain_levels.clone()
    seed = 500 + k
    torch.manual_seed(seed)
    rnd = torch.randint_like(terrain.terrain_levels[ids], terrain.max_terrain_level)
    torch.manual_seed(seed)
    mean_level = cur.terrain_levels_vel(env, ids, R.SceneEntity("robot"))
    steps.append(dict(ids=torch.arange(n) if isinstance(ids, slice) else ids, root=root, command=command, random_levels=rnd,
                      levels=terrain.terrain_levels.clone(), env_origins=terrain.env_origins.clone(), mean_level=mean_level.double(),
                      levels_before=levels_before))
    print(k, "count", count, "mean level", float(mean_level), "wrapped", int(((levels_before[ids] + 1) >= rows).sum()))
for key in ("root", "command", "levels", "env_origins", "mean_level", "levels_before"):
    out[key] = torch.stack([s[key] for s in steps])
out["ids"] = torch.cat([s["ids"] for s in steps])
out["id_counts"] = torch.tensor([len(s["ids"]) for s in steps])
out["random_levels"] = torch.cat([s["random_levels"] for s in steps])
# grid origins (no sub-terrain origins given)
for num in (7, 16, 50):
    terrain.cfg.num_envs = num
    terrain.configure_env_origins(None)
    out[f"grid_{num}"] = terrain.env_origins.clone()
np.savez_compressed(os.path.join(HERE, "curriculum.npz"), **{k: v.numpy() for k, v in out.items()})
print("wrote curriculum.npz with", len(out), "arrays")
