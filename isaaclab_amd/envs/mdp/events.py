"""Reset / push event terms on the HIP kernels ``ilx_event_*`` (interface of ``isaaclab/envs/mdp/events.py:610-866``).

Same signatures as the reference: ``func(env, env_ids, <ranges>, asset_cfg)``.  The reference hands the new state to the
simulator through ``asset.write_*_to_sim``; here it is written into the asset's state buffers
(``data.root_state_w`` / ``joint_pos`` / ``joint_vel``), and the asset's ``write_*_to_sim`` methods are called with the
new rows as well when it has them.  Random numbers come from Philox on the device, keyed by the seed the environment's
``seed`` / ``reset(seed=)`` installed (:mod:`isaaclab_amd.utils.rng`) and that module's draw counter; ``_u`` injects
unit-interval draws instead (parity tests).
"""

from __future__ import annotations

import ctypes

import torch

from ... import _abi
from ...utils import rng as _generators
from ...managers.scene_entity_cfg import SceneEntityCfg

_AXES = ("x", "y", "z", "roll", "pitch", "yaw")


def _range12(d: dict):
    lo = [float(d.get(k, (0.0, 0.0))[0]) for k in _AXES]
    hi = [float(d.get(k, (0.0, 0.0))[1]) for k in _AXES]
    return (ctypes.c_float * 12)(*lo, *hi)


def _ids(env_ids, n, device) -> torch.Tensor:
    if env_ids is None or (isinstance(env_ids, slice) and env_ids == slice(None)):
        return torch.arange(n, dtype=torch.long, device=device)
    return torch.as_tensor(env_ids, dtype=torch.long, device=device).contiguous()


def _data(asset):
    """The asset's data container; a scene may hold a bare ArticulationData in place of an Articulation."""
    return getattr(asset, "data", asset)


def _root_state_written(d):
    """A kernel wrote ``d.root_state_w`` in place: the quantities derived from it (root velocities in the base frame,
    projected gravity, heading) must be recomputed at their next read, as the reference does at every read."""
    if hasattr(d, "_derived_timestamp"):
        d._derived_timestamp = -1.0


def _rng(env, u):
    return (None if u is None else _abi.ptr(_abi.f32c(u)), _generators.seed(), _generators.next_draw("events"))


def reset_root_state_uniform(env, env_ids, pose_range: dict, velocity_range: dict, asset_cfg: SceneEntityCfg = SceneEntityCfg("robot"),
                             *, _u=None):
    """events.py:638-680."""
    asset = env.scene[asset_cfg.name]
    d = _data(asset)
    ids = _ids(env_ids, d.root_state_w.shape[0], d.root_state_w.device)
    origins = getattr(env.scene, "env_origins", None)
    up, seed, draw = _rng(env, _u)
    _abi.check(_abi.event_reset_root_state_uniform(ids.data_ptr(), ids.numel(), _abi.ptr(d.default_root_state),
                                                   None if origins is None else _abi.ptr(_abi.f32c(origins)), _range12(pose_range),
                                                   _range12(velocity_range), up, seed, draw, _abi.ptr(d.root_state_w), _abi.stream()),
               "ilx_event_reset_root_state_uniform")
    _root_state_written(d)
    if hasattr(asset, "write_root_pose_to_sim"):
        asset.write_root_pose_to_sim(d.root_state_w[ids, :7], env_ids=ids)
        asset.write_root_velocity_to_sim(d.root_state_w[ids, 7:], env_ids=ids)


def reset_root_state_from_terrain(env, env_ids, pose_range: dict, velocity_range: dict, asset_cfg: SceneEntityCfg = SceneEntityCfg("robot"),
                                  *, _u=None, _patch_ids=None):
    """events.py:735-799.  ``_u`` (count, 9) / ``_patch_ids`` (count) replay given draws.  Like the reference, which adds the
    velocity samples to the defaults of ALL envs, it only accepts the full env set."""
    asset = env.scene[asset_cfg.name]
    terrain = env.scene.terrain if hasattr(env.scene, "terrain") else env.scene["terrain"]
    valid_positions = terrain.flat_patches.get("init_pos")
    if valid_positions is None:
        raise ValueError("The event term 'reset_root_state_from_terrain' requires valid flat patches under 'init_pos'."
                         f" Found: {list(terrain.flat_patches.keys())}")
    d = _data(asset)
    n = d.root_state_w.shape[0]
    ids = _ids(env_ids, n, d.root_state_w.device)
    if ids.numel() != n:
        raise RuntimeError(f"The size of tensor a ({n}) must match the size of tensor b ({ids.numel()}) at non-singleton dimension 0")
    valid = _abi.f32c(valid_positions)
    picks = None if _patch_ids is None else _patch_ids.to(device=ids.device, dtype=torch.long).contiguous()
    up, seed, draw = _rng(env, _u)
    _abi.check(_abi.event_reset_root_state_from_terrain(
        ids.data_ptr(), ids.numel(), _abi.ptr(d.default_root_state), _abi.ptr(valid), terrain.terrain_levels.data_ptr(),
        terrain.terrain_types.data_ptr(), valid.shape[1], valid.shape[2], None if picks is None else picks.data_ptr(),
        _range12(pose_range), _range12(velocity_range), up, seed, draw, _abi.ptr(d.root_state_w), _abi.stream()),
        "ilx_event_reset_root_state_from_terrain")
    _root_state_written(d)
    if hasattr(asset, "write_root_pose_to_sim"):
        asset.write_root_pose_to_sim(d.root_state_w[ids, :7], env_ids=ids)
        asset.write_root_velocity_to_sim(d.root_state_w[ids, 7:], env_ids=ids)


def _reset_joints(by_scale, env, env_ids, position_range, velocity_range, asset_cfg, u):
    asset = env.scene[asset_cfg.name]
    d = _data(asset)
    n, j = d.joint_pos.shape
    ids = _ids(env_ids, n, d.joint_pos.device)
    up, seed, draw = _rng(env, u)
    _abi.check(_abi.event_reset_joints(int(by_scale), ids.data_ptr(), ids.numel(), _abi.ptr(d.default_joint_pos),
                                       _abi.ptr(d.default_joint_vel), _abi.ptr(d.soft_joint_pos_limits),
                                       _abi.ptr(d.soft_joint_vel_limits), float(position_range[0]), float(position_range[1]),
                                       float(velocity_range[0]), float(velocity_range[1]), up, seed, draw, _abi.ptr(d.joint_pos),
                                       _abi.ptr(d.joint_vel), j, _abi.stream()), "ilx_event_reset_joints")
    if hasattr(asset, "write_joint_state_to_sim"):
        asset.write_joint_state_to_sim(d.joint_pos[ids], d.joint_vel[ids], env_ids=ids)


def reset_joints_by_scale(env, env_ids, position_range, velocity_range, asset_cfg: SceneEntityCfg = SceneEntityCfg("robot"), *, _u=None):
    """events.py:802-832."""
    _reset_joints(True, env, env_ids, position_range, velocity_range, asset_cfg, _u)


def reset_joints_by_offset(env, env_ids, position_range, velocity_range, asset_cfg: SceneEntityCfg = SceneEntityCfg("robot"), *, _u=None):
    """events.py:835-866."""
    _reset_joints(False, env, env_ids, position_range, velocity_range, asset_cfg, _u)


def push_by_setting_velocity(env, env_ids, velocity_range: dict, asset_cfg: SceneEntityCfg = SceneEntityCfg("robot"), *, _u=None):
    """events.py:610-635."""
    asset = env.scene[asset_cfg.name]
    d = _data(asset)
    ids = _ids(env_ids, d.root_state_w.shape[0], d.root_state_w.device)
    up, seed, draw = _rng(env, _u)
    _abi.check(_abi.event_push_by_setting_velocity(ids.data_ptr(), ids.numel(), _range12(velocity_range), up, seed, draw,
                                                   _abi.ptr(d.root_state_w), _abi.stream()), "ilx_event_push_by_setting_velocity")
    _root_state_written(d)
    if hasattr(asset, "write_root_velocity_to_sim"):
        asset.write_root_velocity_to_sim(d.root_state_w[ids, 7:], env_ids=ids)


def reset_root_state_with_random_orientation(env, env_ids, pose_range: dict, velocity_range: dict,
                                             asset_cfg: SceneEntityCfg = SceneEntityCfg("robot"), *, _u=None, _z=None):
    """events.py:683-732.  ``_u`` (count, 9) unit-interval draws (3 position, 6 velocity) and ``_z`` (count, 4) standard
    normal draws replay the reference's; they are given together."""
    asset = env.scene[asset_cfg.name]
    d = _data(asset)
    ids = _ids(env_ids, d.root_state_w.shape[0], d.root_state_w.device)
    origins = getattr(env.scene, "env_origins", None)
    up, seed, draw = _rng(env, _u)
    _abi.check(_abi.event_reset_root_state_random_orientation(
        ids.data_ptr(), ids.numel(), _abi.ptr(d.default_root_state), None if origins is None else _abi.ptr(_abi.f32c(origins)),
        _range12(pose_range), _range12(velocity_range), up, None if _z is None else _abi.ptr(_abi.f32c(_z)), seed, draw,
        _abi.ptr(d.root_state_w), _abi.stream()), "ilx_event_reset_root_state_random_orientation")
    _root_state_written(d)
    if hasattr(asset, "write_root_pose_to_sim"):
        asset.write_root_pose_to_sim(d.root_state_w[ids, :7], env_ids=ids)
        asset.write_root_velocity_to_sim(d.root_state_w[ids, 7:], env_ids=ids)


def apply_external_force_torque(env, env_ids, force_range: tuple[float, float], torque_range: tuple[float, float],
                                asset_cfg: SceneEntityCfg = SceneEntityCfg("robot"), *, _u=None):
    """events.py:579-607: uniform forces / torques into the asset's external wrench buffers
    (``data``-side ``_external_force_b`` / ``_external_torque_b`` of shape (N, bodies, 3), created on first use);
    ``asset.has_external_wrench`` is raised as ``set_external_force_and_torque`` does.  ``_u`` (2, count, bodies, 3)."""
    asset = env.scene[asset_cfg.name]
    d = getattr(asset, "data", asset)
    num_bodies = asset.num_bodies if hasattr(asset, "num_bodies") else d.body_state_w.shape[1]
    n, dev = d.root_state_w.shape[0], d.root_state_w.device
    ids = _ids(env_ids, n, dev)
    for name in ("_external_force_b", "_external_torque_b"):
        if getattr(asset, name, None) is None:
            setattr(asset, name, torch.zeros(n, num_bodies, 3, device=dev))
    body_ids = asset_cfg.body_ids
    if isinstance(body_ids, list):
        bid = torch.tensor(body_ids, dtype=torch.int32, device=dev)
        nb, bptr = len(body_ids), bid.data_ptr()
    else:
        nb, bptr = num_bodies, None
    up, seed, draw = _rng(env, _u)
    _abi.check(_abi.event_apply_external_force_torque(ids.data_ptr(), ids.numel(), bptr, nb, num_bodies, float(force_range[0]),
                                                      float(force_range[1]), float(torque_range[0]), float(torque_range[1]), up, seed,
                                                      draw, _abi.ptr(asset._external_force_b), _abi.ptr(asset._external_torque_b),
                                                      _abi.stream()), "ilx_event_apply_external_force_torque")
    # articulation.py:778-781 tests the sampled values; a non-degenerate range always yields some non-zero entry
    asset.has_external_wrench = any(float(v) != 0.0 for v in (*force_range, *torque_range))


def reset_scene_to_default(env, env_ids):
    """events.py:911-933 for the scene's articulations (rigid and deformable objects are not modelled here)."""
    origins = getattr(env.scene, "env_origins", None)
    for asset in env.scene.articulations.values():
        d = getattr(asset, "data", asset)
        ids = _ids(env_ids, d.root_state_w.shape[0], d.root_state_w.device)
        _abi.check(_abi.event_reset_scene_to_default(ids.data_ptr(), ids.numel(), _abi.ptr(d.default_root_state),
                                                     None if origins is None else _abi.ptr(_abi.f32c(origins)), _abi.ptr(d.root_state_w),
                                                     _abi.ptr(d.default_joint_pos), _abi.ptr(d.default_joint_vel), _abi.ptr(d.joint_pos),
                                                     _abi.ptr(d.joint_vel), d.joint_pos.shape[1], _abi.stream()),
                   "ilx_event_reset_scene_to_default")
        _root_state_written(d)
        if hasattr(asset, "write_root_pose_to_sim"):
            asset.write_root_pose_to_sim(d.root_state_w[ids, :7], env_ids=ids)
            asset.write_root_velocity_to_sim(d.root_state_w[ids, 7:], env_ids=ids)
            asset.write_joint_state_to_sim(d.joint_pos[ids], d.joint_vel[ids], env_ids=ids)
