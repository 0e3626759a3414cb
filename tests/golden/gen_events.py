"""Golden fixtures for the reset / push event terms (envs/mdp/events.py:610-866), produced by the reference's own
functions on a recording asset.  The uniform draws come from ``torch.rand`` on the global generator; they are
re-drawn after re-seeding and stored, so that a kernel can be fed the same numbers."""

from __future__ import annotations

import types

import torch

import _refload as R
from gen_golden import gen, save

SECTIONS = {}


def section(fn):
    SECTIONS[fn.__name__] = fn
    return fn


class Asset:
    def __init__(self, data):
        self.data = data
        self.device = "cpu"
        self.calls = []

    def write_root_pose_to_sim(self, pose, env_ids=None):
        self.calls.append(("pose", pose.clone(), env_ids.clone()))

    def write_root_velocity_to_sim(self, vel, env_ids=None):
        self.calls.append(("vel", vel.clone(), env_ids.clone()))

    def write_joint_state_to_sim(self, pos, vel, env_ids=None):
        self.calls.append(("joint", pos.clone(), vel.clone(), env_ids.clone()))


def redraw(seed, shapes):
    torch.manual_seed(seed)
    return [torch.rand(*s) for s in shapes]


@section
def events():
    R.bootstrap()
    ev = R.load("isaaclab.envs.mdp.events", f"{R.L}/envs/mdp/events.py")
    g = gen(5101)
    n, j = 64, 12
    quat = torch.nn.functional.normalize(torch.randn(n, 4, generator=g), dim=-1)
    data = types.SimpleNamespace(
        default_root_state=torch.cat([torch.randn(n, 3, generator=g) * 0.1 + torch.tensor([0.0, 0.0, 0.6]), quat,
                                      torch.randn(n, 6, generator=g) * 0.05], dim=-1),
        default_joint_pos=torch.randn(n, j, generator=g) * 0.5, default_joint_vel=torch.randn(n, j, generator=g) * 0.3,
        root_vel_w=torch.randn(n, 6, generator=g))
    lo = data.default_joint_pos - 0.4 - torch.rand(n, j, generator=g) * 0.3
    data.soft_joint_pos_limits = torch.stack([lo, data.default_joint_pos + 0.35 + torch.rand(n, j, generator=g) * 0.3], dim=-1)
    data.soft_joint_vel_limits = torch.full((n, j), 0.35)
    origins = torch.randn(n, 3, generator=g) * 20.0
    asset = Asset(data)

    class Scene(dict):
        env_origins = origins

    env = types.SimpleNamespace(scene=Scene(robot=asset))
    env.scene.num_envs = n
    cfg = types.SimpleNamespace(name="robot", joint_ids=slice(None), body_ids=slice(None))
    ids = torch.tensor([3, 0, 17, 63, 40, 41, 9], dtype=torch.long)
    out = dict(default_root_state=data.default_root_state, default_joint_pos=data.default_joint_pos,
               default_joint_vel=data.default_joint_vel, soft_joint_pos_limits=data.soft_joint_pos_limits,
               soft_joint_vel_limits=data.soft_joint_vel_limits, env_origins=origins, root_vel_w=data.root_vel_w, env_ids=ids)
    # velocity_env_cfg.py:166-183 (reset_base) and :192-197 (push_robot) parameter shapes
    pose_range = {"x": (-0.5, 0.5), "y": (-0.5, 0.5), "yaw": (-3.14, 3.14), "roll": (-0.2, 0.1)}
    vel_range = {"x": (-0.5, 0.5), "y": (-0.5, 0.5), "z": (-0.1, 0.2), "roll": (-0.5, 0.5), "pitch": (-0.5, 0.5), "yaw": (-0.5, 0.5)}
    torch.manual_seed(11)
    ev.reset_root_state_uniform(env, ids, pose_range, vel_range, cfg)
    out["root_u_pose"], out["root_u_vel"] = redraw(11, [(len(ids), 6), (len(ids), 6)])
    out["root_pose"], out["root_vel"] = asset.calls[0][1], asset.calls[1][1]
    asset.calls.clear()
    for tag, fn, pr, vr in (("scale", ev.reset_joints_by_scale, (0.5, 1.5), (-1.0, 2.0)),
                            ("offset", ev.reset_joints_by_offset, (-0.6, 0.6), (-0.5, 0.5))):
        torch.manual_seed(12)
        fn(env, ids, pr, vr, cfg)
        out[f"{tag}_u_pos"], out[f"{tag}_u_vel"] = redraw(12, [(len(ids), j), (len(ids), j)])
        out[f"{tag}_joint_pos"], out[f"{tag}_joint_vel"] = asset.calls[0][1], asset.calls[0][2]
        asset.calls.clear()
    torch.manual_seed(13)
    ev.push_by_setting_velocity(env, ids, {"x": (-0.5, 0.5), "y": (-0.5, 0.5), "yaw": (-1.0, 0.25)}, cfg)
    out["push_u"] = redraw(13, [(len(ids), 6)])[0]
    out["push_vel"] = asset.calls[0][1]
    asset.calls.clear()
    # reset_root_state_with_random_orientation (:683-732): rand (k,3), randn (k,4), rand (k,6) in that order
    torch.manual_seed(14)
    ev.reset_root_state_with_random_orientation(env, ids, {"x": (-1.0, 1.0), "z": (0.0, 0.3)}, vel_range, cfg)
    torch.manual_seed(14)
    out["rand_orient_u_pos"] = torch.rand(len(ids), 3)
    out["rand_orient_z"] = torch.randn(len(ids), 4)
    out["rand_orient_u_vel"] = torch.rand(len(ids), 6)
    out["rand_orient_pose"], out["rand_orient_vel"] = asset.calls[0][1], asset.calls[1][1]
    asset.calls.clear()
    # apply_external_force_torque (:579-607) on a body subset, through a recording set_external_force_and_torque
    nb = 5
    asset.num_bodies = nb
    wrench = {}
    asset.set_external_force_and_torque = lambda f, t, env_ids=None, body_ids=None: wrench.update(f=f.clone(), t=t.clone(), env_ids=env_ids.clone(), body_ids=body_ids)
    bcfg = types.SimpleNamespace(name="robot", joint_ids=slice(None), body_ids=[0, 3, 4])
    torch.manual_seed(15)
    ev.apply_external_force_torque(env, ids, (-8.0, 12.0), (-1.5, 0.5), bcfg)
    torch.manual_seed(15)
    out["wrench_u"] = torch.stack([torch.rand(len(ids), 3, 3), torch.rand(len(ids), 3, 3)])
    out["wrench_force"], out["wrench_torque"] = wrench["f"], wrench["t"]
    torch.manual_seed(16)
    ev.apply_external_force_torque(env, None, (0.0, 2.0), (0.0, 0.0), cfg)   # env_ids None -> every env, all bodies
    torch.manual_seed(16)
    out["wrench_all_u"] = torch.stack([torch.rand(n, nb, 3), torch.rand(n, nb, 3)])
    out["wrench_all_force"], out["wrench_all_torque"] = wrench["f"], wrench["t"]
    # reset_scene_to_default (:911-933)
    env.scene.rigid_objects, env.scene.articulations, env.scene.deformable_objects = {}, {"robot": asset}, {}
    ev.reset_scene_to_default(env, ids)
    out["default_pose"], out["default_vel"] = asset.calls[0][1], asset.calls[1][1]
    out["default_joint_pos_rows"], out["default_joint_vel_rows"] = asset.calls[2][1], asset.calls[2][2]
    asset.calls.clear()
    # reset_root_state_from_terrain (:735-799): randint picks, rand (n,3), rand (n,6).  It adds the velocity samples to the
    # defaults of ALL envs, so only the full env set runs; a subset raises.
    rows, cols, patches = 3, 4, 5
    env.device = "cpu"
    env.scene.terrain = types.SimpleNamespace(flat_patches={"init_pos": torch.randn(rows, cols, patches, 3, generator=g) * 6.0},
                                              terrain_levels=torch.randint(0, rows, (n,), generator=g),
                                              terrain_types=torch.randint(0, cols, (n,), generator=g))
    all_ids = torch.arange(n)
    torch.manual_seed(17)
    ev.reset_root_state_from_terrain(env, all_ids, {"roll": (-0.2, 0.1), "yaw": (-3.14, 3.14)}, vel_range, cfg)
    torch.manual_seed(17)
    out["terrain_picks"] = torch.randint(0, patches, size=(n,))
    out["terrain_u"] = torch.cat([torch.rand(n, 3), torch.rand(n, 6)], dim=1)
    out["terrain_valid_positions"] = env.scene.terrain.flat_patches["init_pos"]
    out["terrain_levels"], out["terrain_types"] = env.scene.terrain.terrain_levels, env.scene.terrain.terrain_types
    out["terrain_pose"], out["terrain_vel"] = asset.calls[0][1], asset.calls[1][1]
    try:
        ev.reset_root_state_from_terrain(env, ids, {}, vel_range, cfg)
        raise SystemExit("the reference accepted a subset of envs")
    except RuntimeError as e:
        print("subset of envs:", str(e)[:90])
    save("events", **out)
