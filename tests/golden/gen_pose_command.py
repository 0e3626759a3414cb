"""Golden vectors for UniformPoseCommand (Reach task), produced by the REFERENCE class
(``isaaclab/envs/mdp/commands/pose_command.py`` on top of ``managers/command_manager.py``) with the RNG stream replayed so
that the drawn values are stored as the ``samples`` the kernels consume.  Writes tests/golden/pose_command.npz."""

import math
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
pc = R.load("isaaclab.envs.mdp.commands.pose_command", f"{R.L}/envs/mdp/commands/pose_command.py")
M = m["math"]

out = {}
g = torch.Generator().manual_seed(6301)
n, nb, body = 160, 11, 8
# reach_env_cfg.py:104-117 ranges; short resampling times so that commands expire within the recorded steps
ranges = NS(pos_x=(0.35, 0.65), pos_y=(-0.2, 0.2), pos_z=(0.15, 0.5), roll=(0.0, 0.0), pitch=(math.pi, math.pi), yaw=(-3.14, 3.14))
for tag, unique, rng in (("reach", False, ranges),
                         ("unique", True, NS(pos_x=(0.35, 0.65), pos_y=(-0.2, 0.2), pos_z=(0.15, 0.5), roll=(-1.0, 1.0),
                                             pitch=(2.0, 3.5), yaw=(-3.14, 3.14)))):
    cfg = NS(resampling_time_range=(0.04, 0.2), make_quat_unique=unique, ranges=rng, debug_vis=False, body_name="ee", asset_name="robot")
    root = torch.zeros(n, 13)
    root[:, :3] = torch.randn(n, 3, generator=g)
    root[:, 3:7] = M.normalize(torch.randn(n, 4, generator=g))
    bs = torch.zeros(n, nb, 13)
    bs[..., :3] = torch.randn(n, nb, 3, generator=g)
    bs[..., 3:7] = M.normalize(torch.randn(n, nb, 4, generator=g))
    data = NS(root_pos_w=root[:, :3], root_quat_w=root[:, 3:7], body_state_w=bs)
    term = object.__new__(pc.UniformPoseCommand)
    term._debug_vis_handle = None
    term.cfg = cfg
    term._env = NS(step_dt=1 / 30, num_envs=n, device="cpu")
    term.robot = NS(data=data)
    term.body_idx = body
    term.pose_command_b = torch.cat([torch.rand(n, 3, generator=g), M.normalize(torch.randn(n, 4, generator=g))], -1)
    term.pose_command_w = torch.zeros(n, 7)
    term.metrics = dict(position_error=torch.zeros(n), orientation_error=torch.zeros(n))
    term.time_left = torch.rand(n, generator=g) * 0.1 - 0.02
    term.command_counter = torch.randint(0, 3, (n,), generator=g)
    out[f"{tag}_root_state_w"], out[f"{tag}_body_state_w"] = root, bs
    for k in ("pose_command_b", "time_left", "command_counter"):
        out[f"{tag}_pre_{k}"] = getattr(term, k).clone()
    dt = 1 / 30
    steps = []
    for k in range(3):
        seed = 900 + k
        ids = ((term.time_left - dt) <= 0.0).nonzero().flatten()
        samples = torch.zeros(n, 7)
        torch.manual_seed(seed)
        if len(ids) > 0:  # same stream, same call order as CommandTerm._resample + _resample_command
            tl = torch.zeros(len(ids)).uniform_(*cfg.resampling_time_range)
            r = torch.empty(len(ids))
            px, py, pz = (r.uniform_(*rg).clone() for rg in (rng.pos_x, rng.pos_y, rng.pos_z))
            e = torch.zeros(len(ids), 3)
            e[:, 0].uniform_(*rng.roll), e[:, 1].uniform_(*rng.pitch), e[:, 2].uniform_(*rng.yaw)
            samples[ids] = torch.stack([tl, px, py, pz, e[:, 0], e[:, 1], e[:, 2]], dim=-1)
        torch.manual_seed(seed)
        term.compute(dt)
        steps.append(dict(samples=samples, pose_command_b=term.pose_command_b.clone(), time_left=term.time_left.clone(),
                          command_counter=term.command_counter.clone(), position_error=term.metrics["position_error"].clone(),
                          orientation_error=term.metrics["orientation_error"].clone(), pose_command_w=term.pose_command_w.clone(),
                          resampled=torch.zeros(n, dtype=torch.bool).index_fill_(0, ids, True)))
    for k in steps[0]:
        out[f"{tag}_{k}"] = torch.stack([st[k] for st in steps])
    ids = torch.tensor([1, 7, 33, 120])
    torch.manual_seed(950)
    tl = torch.zeros(len(ids)).uniform_(*cfg.resampling_time_range)
    r = torch.empty(len(ids))
    px, py, pz = (r.uniform_(*rg).clone() for rg in (rng.pos_x, rng.pos_y, rng.pos_z))
    e = torch.zeros(len(ids), 3)
    e[:, 0].uniform_(*rng.roll), e[:, 1].uniform_(*rng.pitch), e[:, 2].uniform_(*rng.yaw)
    samples = torch.zeros(n, 7)
    samples[ids] = torch.stack([tl, px, py, pz, e[:, 0], e[:, 1], e[:, 2]], dim=-1)
    torch.manual_seed(950)
    extras = term.reset(ids)
    out[f"{tag}_reset_ids"], out[f"{tag}_reset_samples"] = ids, samples
    out[f"{tag}_reset_extras"] = torch.tensor([extras["position_error"], extras["orientation_error"]], dtype=torch.float64)
    for a in ("pose_command_b", "time_left", "command_counter"):
        out[f"{tag}_reset_{a}"] = getattr(term, a).clone()
    out[f"{tag}_reset_position_error"] = term.metrics["position_error"].clone()
    out[f"{tag}_cfg"] = torch.tensor([*cfg.resampling_time_range, *rng.pos_x, *rng.pos_y, *rng.pos_z, *rng.roll, *rng.pitch, *rng.yaw,
                                      float(unique)], dtype=torch.float64)
    print(tag, "resampled per step", [int(s["resampled"].sum()) for s in steps])
np.savez_compressed(os.path.join(HERE, "pose_command.npz"), **{k: v.numpy() for k, v in out.items()})
print("wrote pose_command.npz with", len(out), "arrays")
