"""Golden vectors for NormalVelocityCommand, produced by the REFERENCE class
(``isaaclab/envs/mdp/commands/velocity_command.py`` on ``managers/command_manager.py``) with the RNG stream replayed so that
the drawn values are stored as the samples the kernel consumes.  Writes tests/golden/normal_velocity_command.npz."""

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
vc = R.load("isaaclab.envs.mdp.commands.velocity_command", f"{R.L}/envs/mdp/commands/velocity_command.py")
M = m["math"]

out = {}
g = torch.Generator().manual_seed(8117)
n, dt = 150, 0.02
cfg = NS(resampling_time_range=(0.03, 0.09), debug_vis=False, asset_name="robot", heading_command=False, rel_standing_envs=0.15,
         ranges=NS(mean_vel=(1.0, 0.4, 0.6), std_vel=(0.5, 0.2, 0.3), zero_prob=(0.1, 0.3, 0.2)))
term = object.__new__(vc.NormalVelocityCommand)
term._debug_vis_handle = None
term.cfg = cfg
term._env = NS(step_dt=dt, num_envs=n, device="cpu")
term.robot = NS(data=NS())
term.vel_command_b = torch.randn(n, 3, generator=g)
term.heading_target = torch.zeros(n)
term.is_heading_env = torch.zeros(n, dtype=torch.bool)
term.is_standing_env = torch.rand(n, generator=g) < 0.1
term.is_zero_vel_x_env = torch.rand(n, generator=g) < 0.1
term.is_zero_vel_y_env = torch.rand(n, generator=g) < 0.1
term.is_zero_vel_yaw_env = torch.rand(n, generator=g) < 0.1
term.metrics = dict(error_vel_xy=torch.rand(n, generator=g), error_vel_yaw=torch.rand(n, generator=g))
term.time_left = torch.rand(n, generator=g) * 0.06 - 0.01
term.command_counter = torch.randint(0, 3, (n,), generator=g)
for k in ("vel_command_b", "is_standing_env", "is_zero_vel_x_env", "is_zero_vel_y_env", "is_zero_vel_yaw_env", "time_left", "command_counter"):
    out[f"pre_{k}"] = getattr(term, k).clone()
out["pre_error_vel_xy"], out["pre_error_vel_yaw"] = term.metrics["error_vel_xy"].clone(), term.metrics["error_vel_yaw"].clone()


def replay_draws(ids, seed):
    """The values the reference draws in CommandTerm._resample + _resample_command for ``ids`` under ``seed``."""
    s = torch.zeros(n, 11)
    torch.manual_seed(seed)
    if len(ids) > 0:
        s[ids, 0] = torch.zeros(len(ids)).uniform_(*cfg.resampling_time_range)
        r = torch.empty(len(ids))
        for k in range(3):
            s[ids, 1 + k] = r.normal_(mean=cfg.ranges.mean_vel[k], std=cfg.ranges.std_vel[k]).clone()
            s[ids, 4 + k] = r.uniform_(0.0, 1.0).clone()
        for k in range(4):
            s[ids, 7 + k] = r.uniform_(0.0, 1.0).clone()
    return s


def snapshot():
    return dict(vel_command_b=term.vel_command_b.clone(), is_standing_env=term.is_standing_env.clone(),
                is_zero=torch.stack([term.is_zero_vel_x_env, term.is_zero_vel_y_env, term.is_zero_vel_yaw_env], dim=1).clone(),
                time_left=term.time_left.clone(), command_counter=term.command_counter.clone(),
                error_vel_xy=term.metrics["error_vel_xy"].clone(), error_vel_yaw=term.metrics["error_vel_yaw"].clone())


steps = []
for k in range(5):
    root = torch.zeros(n, 13)
    root[:, :3] = torch.randn(n, 3, generator=g)
    root[:, 3:7] = M.normalize(torch.randn(n, 4, generator=g))
    root[:, 7:] = torch.randn(n, 6, generator=g)
    term.robot.data.root_lin_vel_b = M.quat_rotate_inverse(root[:, 3:7], root[:, 7:10])
    term.robot.data.root_ang_vel_b = M.quat_rotate_inverse(root[:, 3:7], root[:, 10:13])
    ids = ((term.time_left - dt) <= 0.0).nonzero().flatten()
    seed = 900 + k
    samples = replay_draws(ids, seed)
    torch.manual_seed(seed)
    term.compute(dt)
    steps.append(dict(root_state_w=root, samples=samples, **snapshot()))
for key in steps[0]:
    out[key] = torch.stack([s[key] for s in steps])
ids = torch.tensor([0, 7, 64, 149])
samples = replay_draws(ids, 990)
torch.manual_seed(990)
extras = term.reset(ids)
out["reset_ids"], out["reset_samples"] = ids, samples
out["reset_extras"] = torch.tensor([extras["error_vel_xy"], extras["error_vel_yaw"]], dtype=torch.float64)
for k, v in snapshot().items():
    out[f"reset_{k}"] = v
print("resampled per step", [int((s["samples"][:, 0] != 0).sum()) for s in steps])
np.savez_compressed(os.path.join(HERE, "normal_velocity_command.npz"), **{k: v.numpy() for k, v in out.items()})
print("wrote normal_velocity_command.npz with", len(out), "arrays")
