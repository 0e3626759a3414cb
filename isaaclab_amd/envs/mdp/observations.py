"""Observation terms with the reference's names and call signatures (``isaaclab/envs/mdp/observations.py``):
``func(env, **params) -> (N, d) tensor``."""

from __future__ import annotations

from .. import fused_mdp as FM
from ...managers.scene_entity_cfg import SceneEntityCfg
from ._fused import Spec, observation_term

_ROBOT = SceneEntityCfg("robot")


@observation_term
def base_pos_z(asset_cfg: SceneEntityCfg = _ROBOT):  # :33-37
    return Spec("base_pos_z", asset=asset_cfg.name)


@observation_term
def base_lin_vel(asset_cfg: SceneEntityCfg = _ROBOT):  # :40-44
    return Spec("base_lin_vel", asset=asset_cfg.name)


@observation_term
def base_ang_vel(asset_cfg: SceneEntityCfg = _ROBOT):  # :47-51
    return Spec("base_ang_vel", asset=asset_cfg.name)


@observation_term
def projected_gravity(asset_cfg: SceneEntityCfg = _ROBOT):  # :54-58
    return Spec("projected_gravity", asset=asset_cfg.name)


@observation_term
def root_pos_w(asset_cfg: SceneEntityCfg = _ROBOT):  # :61-65
    return Spec("root_pos_w", asset=asset_cfg.name)


@observation_term
def root_quat_w(make_quat_unique: bool = False, asset_cfg: SceneEntityCfg = _ROBOT):  # :68-82
    return Spec("root_quat_w", flags=FM.OBS_QUAT_UNIQUE if make_quat_unique else 0, asset=asset_cfg.name)


@observation_term
def root_lin_vel_w(asset_cfg: SceneEntityCfg = _ROBOT):  # :85-89
    return Spec("root_lin_vel_w", asset=asset_cfg.name)


@observation_term
def root_ang_vel_w(asset_cfg: SceneEntityCfg = _ROBOT):  # :92-96
    return Spec("root_ang_vel_w", asset=asset_cfg.name)


@observation_term
def joint_pos(asset_cfg: SceneEntityCfg = _ROBOT):  # :104-111
    return Spec("joint_pos", ids=asset_cfg.joint_ids, asset=asset_cfg.name)


@observation_term
def joint_pos_rel(asset_cfg: SceneEntityCfg = _ROBOT):  # :114-121
    return Spec("joint_pos_rel", ids=asset_cfg.joint_ids, asset=asset_cfg.name)


@observation_term
def joint_pos_limit_normalized(asset_cfg: SceneEntityCfg = _ROBOT):  # :124-137
    return Spec("joint_pos_limit_normalized", ids=asset_cfg.joint_ids, asset=asset_cfg.name)


@observation_term
def joint_vel(asset_cfg: SceneEntityCfg = _ROBOT):  # :140-147
    return Spec("joint_vel", ids=asset_cfg.joint_ids, asset=asset_cfg.name)


@observation_term
def joint_vel_rel(asset_cfg: SceneEntityCfg = _ROBOT):  # :150-157
    return Spec("joint_vel_rel", ids=asset_cfg.joint_ids, asset=asset_cfg.name)


@observation_term
def height_scan(sensor_cfg: SceneEntityCfg, offset: float = 0.5):  # :165-173
    return Spec("height_scan", aux=offset, asset=None, ray=sensor_cfg.name)


@observation_term
def last_action(action_name: str | None = None):  # :512-521
    if action_name is not None:
        raise NotImplementedError("last_action of a single action term: pass action_name=None (the whole action vector)")
    return Spec("last_action", asset=None)


@observation_term
def generated_commands(command_name: str):  # :529-531
    return Spec("generated_commands", asset=None, command=command_name)


# ---- terms that hand out a buffer another component keeps (observations.py:177-230): copied into the row by the fused launch
def _sensor(env, name):
    sensors = getattr(env.scene, "sensors", None)
    return sensors[name] if sensors is not None and name in sensors else env.scene[name]


@observation_term
def body_incoming_wrench(asset_cfg: SceneEntityCfg):  # :177-185
    def read(env):
        asset = env.scene[asset_cfg.name]
        forces = asset.root_physx_view.get_link_incoming_joint_force()[:, asset_cfg.body_ids]
        return forces.reshape(env.num_envs, -1)
    return Spec("external", asset=None, external=read)


@observation_term
def imu_orientation(asset_cfg: SceneEntityCfg = SceneEntityCfg("imu")):  # :188-201
    return Spec("external", asset=None, external=lambda env: _sensor(env, asset_cfg.name).data.quat_w)


@observation_term
def imu_ang_vel(asset_cfg: SceneEntityCfg = SceneEntityCfg("imu")):  # :204-217
    return Spec("external", asset=None, external=lambda env: _sensor(env, asset_cfg.name).data.ang_vel_b)


@observation_term
def imu_lin_acc(asset_cfg: SceneEntityCfg = SceneEntityCfg("imu")):  # :220-231
    return Spec("external", asset=None, external=lambda env: _sensor(env, asset_cfg.name).data.lin_acc_b)
