"""Reward terms with the reference's names and call signatures ``func(env, **params) -> (N,) tensor``.

Library terms: ``isaaclab/envs/mdp/rewards.py`` (R).  Task terms: ``isaaclab_tasks/manager_based/locomotion/velocity/
mdp/rewards.py`` (V), ``.../manipulation/reach/mdp/rewards.py`` (RE), ``.../classic/cartpole/mdp/rewards.py`` (CP).
Each definition below only maps the parameters to a kernel term; the formulas live in ``csrc/mdp.hip`` (one
``case`` per term, citing the same lines)."""

from __future__ import annotations

from ...managers.scene_entity_cfg import SceneEntityCfg
from ._fused import Spec, reward_term

_ROBOT = SceneEntityCfg("robot")


@reward_term
def is_alive():  # R:31-33
    return Spec("is_alive", asset=None)


@reward_term
def is_terminated():  # R:36-38
    return Spec("is_terminated", asset=None)


@reward_term
def is_terminated_term(term_keys: str | list[str] = ".*"):  # R:41-68 (a class term in the reference)
    return Spec("is_terminated_term", asset=None, term_names=None if term_keys == ".*" else
                ([term_keys] if isinstance(term_keys, str) else list(term_keys)))


@reward_term
def lin_vel_z_l2(asset_cfg: SceneEntityCfg = _ROBOT):  # R:76-80
    return Spec("lin_vel_z_l2", asset=asset_cfg.name)


@reward_term
def ang_vel_xy_l2(asset_cfg: SceneEntityCfg = _ROBOT):  # R:83-87
    return Spec("ang_vel_xy_l2", asset=asset_cfg.name)


@reward_term
def flat_orientation_l2(asset_cfg: SceneEntityCfg = _ROBOT):  # R:90-97
    return Spec("flat_orientation_l2", asset=asset_cfg.name)


@reward_term
def base_height_l2(target_height: float, asset_cfg: SceneEntityCfg = _ROBOT, sensor_cfg: SceneEntityCfg | None = None):  # R:100-122
    return Spec("base_height_l2", p=(target_height, 0, 0, 0), flags=1 if sensor_cfg is not None else 0, asset=asset_cfg.name,
                ray=None if sensor_cfg is None else sensor_cfg.name)


@reward_term
def body_lin_acc_l2(asset_cfg: SceneEntityCfg = _ROBOT):  # R:125-128
    return Spec("body_lin_acc_l2", ids=asset_cfg.body_ids, asset=asset_cfg.name)


@reward_term
def joint_torques_l2(asset_cfg: SceneEntityCfg = _ROBOT):  # R:136-143
    return Spec("joint_torques_l2", ids=asset_cfg.joint_ids, asset=asset_cfg.name)


@reward_term
def joint_vel_l1(asset_cfg: SceneEntityCfg):  # R:146-150
    return Spec("joint_vel_l1", ids=asset_cfg.joint_ids, asset=asset_cfg.name)


@reward_term
def joint_vel_l2(asset_cfg: SceneEntityCfg = _ROBOT):  # R:153-160
    return Spec("joint_vel_l2", ids=asset_cfg.joint_ids, asset=asset_cfg.name)


@reward_term
def joint_acc_l2(asset_cfg: SceneEntityCfg = _ROBOT):  # R:163-170
    return Spec("joint_acc_l2", ids=asset_cfg.joint_ids, asset=asset_cfg.name)


@reward_term
def joint_deviation_l1(asset_cfg: SceneEntityCfg = _ROBOT):  # R:173-179
    return Spec("joint_deviation_l1", ids=asset_cfg.joint_ids, asset=asset_cfg.name)


@reward_term
def joint_pos_limits(asset_cfg: SceneEntityCfg = _ROBOT):  # R:182-196
    return Spec("joint_pos_limits", ids=asset_cfg.joint_ids, asset=asset_cfg.name)


@reward_term
def joint_vel_limits(soft_ratio: float, asset_cfg: SceneEntityCfg = _ROBOT):  # R:199-218
    return Spec("joint_vel_limits", p=(soft_ratio, 0, 0, 0), ids=asset_cfg.joint_ids, asset=asset_cfg.name)


@reward_term
def applied_torque_limits(asset_cfg: SceneEntityCfg = _ROBOT):  # R:226-242
    return Spec("applied_torque_limits", ids=asset_cfg.joint_ids, asset=asset_cfg.name)


@reward_term
def action_rate_l2():  # R:245-247
    return Spec("action_rate_l2", asset=None)


@reward_term
def action_l2():  # R:250-252
    return Spec("action_l2", asset=None)


@reward_term
def undesired_contacts(threshold: float, sensor_cfg: SceneEntityCfg):  # R:260-268
    return Spec("undesired_contacts", p=(threshold, 0, 0, 0), ids=sensor_cfg.body_ids, asset=None, contact=sensor_cfg.name)


@reward_term
def contact_forces(threshold: float, sensor_cfg: SceneEntityCfg):  # R:271-279
    return Spec("contact_forces", p=(threshold, 0, 0, 0), ids=sensor_cfg.body_ids, asset=None, contact=sensor_cfg.name)


@reward_term
def track_lin_vel_xy_exp(std: float, command_name: str, asset_cfg: SceneEntityCfg = _ROBOT):  # R:287-298
    return Spec("track_lin_vel_xy_exp", p=(std**2, 0, 0, 0), asset=asset_cfg.name, command=command_name)


@reward_term
def track_ang_vel_z_exp(std: float, command_name: str, asset_cfg: SceneEntityCfg = _ROBOT):  # R:301-309
    return Spec("track_ang_vel_z_exp", p=(std**2, 0, 0, 0), asset=asset_cfg.name, command=command_name)


# ---- locomotion velocity task (V) -----------------------------------------------------------------------------------
def feet_air_time(env, command_name: str, sensor_cfg: SceneEntityCfg, threshold: float):
    """V:25-44.  ``compute_first_contact(env.step_dt)`` compares against ``step_dt + 1e-8`` (contact_sensor.py:200-203)."""
    return _feet_air_time(env, command_name=command_name, sensor_cfg=sensor_cfg, threshold=threshold, step_dt=env.step_dt)


@reward_term
def _feet_air_time(command_name: str, sensor_cfg: SceneEntityCfg, threshold: float, step_dt: float):
    return Spec("feet_air_time", p=(threshold, step_dt + 1.0e-8, 0, 0), ids=sensor_cfg.body_ids, contact=sensor_cfg.name,
                command=command_name)


feet_air_time._ilx_build = lambda command_name, sensor_cfg, threshold, step_dt=0.02: _feet_air_time._ilx_build(
    command_name=command_name, sensor_cfg=sensor_cfg, threshold=threshold, step_dt=step_dt)
feet_air_time._ilx_group = "reward"
feet_air_time._ilx_needs_step_dt = True


@reward_term
def feet_air_time_positive_biped(command_name: str, threshold: float, sensor_cfg: SceneEntityCfg):  # V:47-66
    return Spec("feet_air_time_positive_biped", p=(threshold, 0, 0, 0), ids=sensor_cfg.body_ids, contact=sensor_cfg.name,
                command=command_name)


@reward_term
def feet_slide(sensor_cfg: SceneEntityCfg, asset_cfg: SceneEntityCfg = _ROBOT):  # V:69-83
    return Spec("feet_slide", ids=sensor_cfg.body_ids, ids2=asset_cfg.body_ids, asset=asset_cfg.name, contact=sensor_cfg.name)


@reward_term
def track_lin_vel_xy_yaw_frame_exp(std: float, command_name: str, asset_cfg: SceneEntityCfg = _ROBOT):  # V:86-96
    return Spec("track_lin_vel_xy_yaw_frame_exp", p=(std**2, 0, 0, 0), asset=asset_cfg.name, command=command_name)


@reward_term
def track_ang_vel_z_world_exp(command_name: str, std: float, asset_cfg: SceneEntityCfg = _ROBOT):  # V:99-106
    return Spec("track_ang_vel_z_world_exp", p=(std**2, 0, 0, 0), asset=asset_cfg.name, command=command_name)


# ---- reach task (RE) ------------------------------------------------------------------------------------------------
def _one_body(asset_cfg: SceneEntityCfg):
    ids = asset_cfg.body_ids
    if isinstance(ids, slice) or len(ids) != 1:
        raise ValueError("the pose-command rewards track exactly one body (asset_cfg.body_ids[0])")
    return ids


@reward_term
def position_command_error(command_name: str, asset_cfg: SceneEntityCfg):  # RE:19-33
    return Spec("position_command_error", ids=_one_body(asset_cfg), asset=asset_cfg.name, command=command_name)


@reward_term
def position_command_error_tanh(std: float, command_name: str, asset_cfg: SceneEntityCfg):  # RE:36-52
    return Spec("position_command_error_tanh", p=(std, 0, 0, 0), ids=_one_body(asset_cfg), asset=asset_cfg.name,
                command=command_name)


@reward_term
def orientation_command_error(command_name: str, asset_cfg: SceneEntityCfg):  # RE:55-69
    return Spec("orientation_command_error", ids=_one_body(asset_cfg), asset=asset_cfg.name, command=command_name)


# ---- cartpole task (CP) ---------------------------------------------------------------------------------------------
@reward_term
def joint_pos_target_l2(target: float, asset_cfg: SceneEntityCfg):  # CP:19-26
    return Spec("joint_pos_target_l2", p=(target, 0, 0, 0), ids=asset_cfg.joint_ids, asset=asset_cfg.name)
