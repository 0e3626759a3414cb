"""CPU: the terrain-curriculum oracle against the fixture recorded from the reference's term and TerrainImporter."""

import torch

from conftest import assert_exact, load_golden
from oracle import curriculum as C


def test_terrain_levels_sequence():
    g = load_golden("curriculum")
    levels, origins = g["init_levels"].clone(), g["init_env_origins"].clone()
    ids_all = torch.split(g["ids"], g["id_counts"].tolist())
    rnd_all = torch.split(g["random_levels"], g["id_counts"].tolist())
    for k, (ids, rnd) in enumerate(zip(ids_all, rnd_all)):
        assert_exact(levels, g["levels_before"][k], f"levels before step {k}")
        up, down = C.move_flags(g["root"][k][:, :3], origins, g["command"][k], ids, 4.0, 20.0)
        assert not (up & down).any()
        C.update_env_origins(levels, g["terrain_types"], g["terrain_origins"], origins, ids, up, down, rnd)
        assert_exact(levels, g["levels"][k], f"levels step {k}")
        assert_exact(origins, g["env_origins"][k], f"env origins step {k}")
        assert abs(float(levels.float().mean()) - float(g["mean_level"][k])) < 1e-6
    assert int(levels.min()) >= 0 and int(levels.max()) < g["terrain_origins"].shape[0]
