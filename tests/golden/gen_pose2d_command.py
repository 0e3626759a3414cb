"""Golden vectors for UniformPose2dCommand / TerrainBasedPose2dCommand, produced by the REFERENCE classes
(``isaaclab/envs/mdp/commands/pose_2d_command.py`` on ``managers/command_manager.py``) with the RNG stream replayed so that
the drawn values are stored as the samples the kernels consume.  Writes tests/golden/pose2d_command.npz."""

import os
import sys
from types import SimpleNamespace as NS

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refload as R  # noqa: E402

m = R.bootstrap()
R.placeholder("isaaclab.managers.action_manager")
cm = R.load("isaaclab.managers.command_manager", f"{R.L}/managers/command_manager.py")
sys.modules["isaaclab.managers"].CommandTerm = cm.CommandTerm
R.placeholder("isaaclab.envs.mdp.commands")
p2 = R.load("isaaclab.envs.mdp.commands.pose_2d_command", f"{R.L}/envs/mdp/commands/pose_2d_command.py")
M = m["math"]

out = {}
g = torch.Generator().manual_seed(6402)
n, rows, cols, patches = 120, 4, 5, 6
dt = 0.02


def heading_of(quat):
    fwd = M.quat_apply(quat, torch.tensor([[1.0, 0.0, 0.0]]).repeat(len(quat), 1))
    return torch.atan2(fwd[:, 1], fwd[:, 0])


def make(cls, tag, simple):
    cfg = NS(resampling_time_range=(0.03, 0.09), simple_heading=simple, debug_vis=False, asset_name="robot",
             ranges=NS(pos_x=(-3.0, 3.0), pos_y=(-2.0, 2.5), heading=(-3.0, 3.1)))
    term = object.__new__(cls)
    term._debug_vis_handle = None
    term.cfg = cfg
    origins = torch.randn(n, 3, generator=g) * 10.0
    term._env = NS(step_dt=dt, num_envs=n, device="cpu", scene=NS(env_origins=origins))
    default = torch.zeros(n, 13)
    default[:, 2] = 0.4 + torch.rand(n, generator=g) * 0.3
    term.robot = NS(data=NS(default_root_state=default))
    term.pos_command_w = origins + torch.randn(n, 3, generator=g)
    term.heading_command_w = torch.rand(n, generator=g) * 6.0 - 3.0
    term.pos_command_b, term.heading_command_b = torch.zeros(n, 3), torch.zeros(n)
    term.metrics = dict(error_pos=torch.zeros(n), error_heading=torch.zeros(n))
    term.time_left = torch.rand(n, generator=g) * 0.06 - 0.01
    term.command_counter = torch.randint(0, 3, (n,), generator=g)
    out[f"{tag}_env_origins"], out[f"{tag}_default_root_state"] = origins, default
    if cls is p2.TerrainBasedPose2dCommand:
        term.valid_targets = torch.randn(rows, cols, patches, 3, generator=g) * 8.0
        term.terrain = NS(terrain_levels=torch.randint(0, rows, (n,), generator=g), terrain_types=torch.randint(0, cols, (n,), generator=g))
        out[f"{tag}_valid_targets"] = term.valid_targets
        out[f"{tag}_terrain_levels"], out[f"{tag}_terrain_types"] = term.terrain.terrain_levels, term.terrain.terrain_types
    for k in ("pos_command_w", "heading_command_w", "time_left", "command_counter"):
        out[f"{tag}_pre_{k}"] = getattr(term, k).clone()
    return term, cfg


def replay_draws(term, cfg, ids, seed, terrain):
    """The values the reference draws in CommandTerm._resample + _resample_command for ``ids`` under ``seed``."""
    samples, picks = torch.zeros(n, 4), torch.zeros(n, dtype=torch.long)
    torch.manual_seed(seed)
    if len(ids) > 0:
        samples[ids, 0] = torch.zeros(len(ids)).uniform_(*cfg.resampling_time_range)
        if terrain:
            picks[ids] = torch.randint(0, patches, size=(len(ids),))
            if not cfg.simple_heading:
                samples[ids, 3] = torch.empty(len(ids)).uniform_(*cfg.ranges.heading)
        else:
            r = torch.empty(len(ids))
            samples[ids, 1] = r.uniform_(*cfg.ranges.pos_x).clone()
            samples[ids, 2] = r.uniform_(*cfg.ranges.pos_y).clone()
            if not cfg.simple_heading:
                samples[ids, 3] = r.uniform_(*cfg.ranges.heading).clone()
    return samples, picks


for tag, cls, simple in (("uniform", p2.UniformPose2dCommand, False), ("uniform_simple", p2.UniformPose2dCommand, True),
                         ("terrain", p2.TerrainBasedPose2dCommand, False), ("terrain_simple", p2.TerrainBasedPose2dCommand, True)):
    term, cfg = make(cls, tag, simple)
    terrain = cls is p2.TerrainBasedPose2dCommand
    steps = []
    for k in range(4):
        root = torch.zeros(n, 13)
        root[:, :3] = out[f"{tag}_env_origins"] + torch.randn(n, 3, generator=g)
        root[:, 3:7] = M.normalize(torch.randn(n, 4, generator=g) * torch.tensor([1.0, 0.15, 0.15, 1.0]))
        term.robot.data.root_pos_w, term.robot.data.root_quat_w = root[:, :3], root[:, 3:7]
        term.robot.data.heading_w = heading_of(root[:, 3:7])
        ids = ((term.time_left - dt) <= 0.0).nonzero().flatten()
        seed = 700 + k
        samples, picks = replay_draws(term, cfg, ids, seed, terrain)
        torch.manual_seed(seed)
        term.compute(dt)
        steps.append(dict(root_state_w=root, samples=samples, picks=picks, pos_command_w=term.pos_command_w.clone(),
                          heading_command_w=term.heading_command_w.clone(), command=term.command.clone(),
                          time_left=term.time_left.clone(), command_counter=term.command_counter.clone(),
                          error_pos_2d=term.metrics["error_pos_2d"].clone(), error_heading=term.metrics["error_heading"].clone()))
    for key in steps[0]:
        out[f"{tag}_{key}"] = torch.stack([s[key] for s in steps])
    ids = torch.tensor([2, 11, 50, 119])
    samples, picks = replay_draws(term, cfg, ids, 790, terrain)
    torch.manual_seed(790)
    extras = term.reset(ids)
    out[f"{tag}_reset_ids"], out[f"{tag}_reset_samples"], out[f"{tag}_reset_picks"] = ids, samples, picks
    out[f"{tag}_reset_extras"] = torch.tensor([extras["error_pos"], extras["error_heading"], extras["error_pos_2d"]], dtype=torch.float64)
    for a in ("pos_command_w", "heading_command_w", "time_left", "command_counter"):
        out[f"{tag}_reset_{a}"] = getattr(term, a).clone()
    out[f"{tag}_reset_error_pos_2d"] = term.metrics["error_pos_2d"].clone()
    print(tag, "resampled per step", [int((s["samples"][:, 0] != 0).sum()) for s in steps])
np.savez_compressed(os.path.join(HERE, "pose2d_command.npz"), **{k: v.numpy() for k, v in out.items()})
print("wrote pose2d_command.npz with", len(out), "arrays")
