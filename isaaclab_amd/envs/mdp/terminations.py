"""Termination terms with the reference's names and call signatures (``isaaclab/envs/mdp/terminations.py``):
``func(env, **params) -> (N,) bool tensor``."""

from __future__ import annotations

from ...managers.scene_entity_cfg import SceneEntityCfg
from ._fused import Spec, termination_term

_ROBOT = SceneEntityCfg("robot")


@termination_term
def time_out():  # :30-32
    return Spec("time_out", asset=None)


def command_resample(env, command_name: str, num_resamples: int = 1):
    """:35-42 (reads the command term's ``time_left`` / ``command_counter``)."""
    return _command_resample(env, command_name=command_name, num_resamples=num_resamples, step_dt=env.step_dt)


@termination_term
def _command_resample(command_name: str, num_resamples: int, step_dt: float):
    return Spec("command_resample", p=(step_dt, num_resamples, 0, 0), asset=None, command=command_name)


command_resample._ilx_build = lambda command_name, num_resamples=1, step_dt=0.02: _command_resample._ilx_build(
    command_name=command_name, num_resamples=num_resamples, step_dt=step_dt)
command_resample._ilx_group = "termination"
command_resample._ilx_needs_step_dt = True


@termination_term
def bad_orientation(limit_angle: float, asset_cfg: SceneEntityCfg = _ROBOT):  # :50-59
    return Spec("bad_orientation", p=(limit_angle, 0, 0, 0), asset=asset_cfg.name)


@termination_term
def root_height_below_minimum(minimum_height: float, asset_cfg: SceneEntityCfg = _ROBOT):  # :62-72
    return Spec("root_height_below_minimum", p=(minimum_height, 0, 0, 0), asset=asset_cfg.name)


def joint_pos_out_of_limit(env, asset_cfg: SceneEntityCfg = _ROBOT):
    """:80-87.  In the reference snapshot this term cannot run: it indexes a 1-D ``(N,)`` result with
    ``[:, joint_ids]`` and raises ``IndexError`` on every call.  The same exception is raised here."""
    raise IndexError("too many indices for tensor of dimension 1")


@termination_term
def joint_pos_out_of_manual_limit(bounds: tuple[float, float], asset_cfg: SceneEntityCfg = _ROBOT):  # :90-105
    return Spec("joint_pos_out_of_manual_limit", p=(bounds[0], bounds[1], 0, 0), ids=asset_cfg.joint_ids, asset=asset_cfg.name)


@termination_term
def joint_vel_out_of_limit(asset_cfg: SceneEntityCfg = _ROBOT):  # :108-114
    return Spec("joint_vel_out_of_limit", ids=asset_cfg.joint_ids, asset=asset_cfg.name)


@termination_term
def joint_vel_out_of_manual_limit(max_velocity: float, asset_cfg: SceneEntityCfg = _ROBOT):  # :117-124
    return Spec("joint_vel_out_of_manual_limit", p=(max_velocity, 0, 0, 0), ids=asset_cfg.joint_ids, asset=asset_cfg.name)


@termination_term
def joint_effort_out_of_limit(asset_cfg: SceneEntityCfg = _ROBOT):
    """:127-142, as written in the reference: ``any(isclose(computed, applied))`` -- true for envs with at least one
    UNclipped joint (the docstring there says the opposite)."""
    return Spec("joint_effort_out_of_limit", ids=asset_cfg.joint_ids, asset=asset_cfg.name)


@termination_term
def illegal_contact(threshold: float, sensor_cfg: SceneEntityCfg):  # :150-158
    return Spec("illegal_contact", p=(threshold, 0, 0, 0), ids=sensor_cfg.body_ids, asset=None, contact=sensor_cfg.name)
