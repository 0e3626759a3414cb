"""CPU: the 2-D pose command oracle against the fixture recorded from the reference classes."""

import pytest
import torch

from conftest import assert_close, assert_exact, load_golden
from oracle import command as C


@pytest.mark.parametrize("tag", ["uniform", "uniform_simple", "terrain", "terrain_simple"])
def test_pose2d_command_sequence(tag):
    g = load_golden("pose2d_command")
    st = {k: g[f"{tag}_pre_{k}"].clone() for k in ("pos_command_w", "heading_command_w", "time_left", "command_counter")}
    kw = dict(simple_heading="simple" in tag, env_origins=g[f"{tag}_env_origins"], default_root_z=g[f"{tag}_default_root_state"][:, 2])
    if tag.startswith("terrain"):
        kw.update(valid_targets=g[f"{tag}_valid_targets"], terrain_levels=g[f"{tag}_terrain_levels"], terrain_types=g[f"{tag}_terrain_types"])
    for k in range(g[f"{tag}_samples"].shape[0]):
        cmd, e_pos, e_head = C.pose2d_compute(st, g[f"{tag}_root_state_w"][k], 0.02, g[f"{tag}_samples"][k], g[f"{tag}_picks"][k], **kw)
        assert_exact(st["time_left"], g[f"{tag}_time_left"][k], f"{tag} time_left {k}")
        assert_exact(st["command_counter"], g[f"{tag}_command_counter"][k], f"{tag} counter {k}")
        assert_exact(st["pos_command_w"], g[f"{tag}_pos_command_w"][k], f"{tag} pos_command_w {k}")
        assert_close(st["heading_command_w"], g[f"{tag}_heading_command_w"][k], atol=1e-6, what=f"{tag} heading_command_w {k}")
        assert_close(cmd, g[f"{tag}_command"][k], atol=2e-6, what=f"{tag} command {k}")
        assert_close(e_pos, g[f"{tag}_error_pos_2d"][k], atol=1e-6, what=f"{tag} error_pos_2d {k}")
        assert_close(e_head, g[f"{tag}_error_heading"][k], atol=1e-6, what=f"{tag} error_heading {k}")
