"""``InteractiveScene.update(dt)`` (``isaaclab/scene/interactive_scene.py:496-513``) as ONE launch.

The reference loops over the scene entities: ``articulation.update(dt)`` (joint accelerations), then every sensor's
``update(dt, force_recompute=not lazy_sensor_update)`` (clocks; eager refresh for sensors with a history).  For the
usual locomotion scene -- one articulation, one contact sensor, one ray caster -- ``ilx_scene_update`` does all of it
in a single kernel; any other composition falls back to the entities' own ``update`` methods (one launch each).
"""

from __future__ import annotations

import ctypes

from . import _abi
from .assets.articulation_data import ArticulationData  # noqa: F401  (the type of the entries of `articulations`, or of their `.data`)
from .sensors.contact_sensor import ContactSensor
from .sensors.ray_caster import RayCaster


class InteractiveScene:
    def __init__(self, num_envs: int, device, lazy_sensor_update: bool = True):
        self.num_envs, self.device = num_envs, device
        self.lazy_sensor_update = lazy_sensor_update
        self.articulations: dict[str, ArticulationData] = {}
        self.sensors: dict[str, object] = {}

    def __getitem__(self, key: str):
        """interactive_scene.py:547-570: assets and sensors share one namespace."""
        if key in self.articulations:
            return self.articulations[key]
        if key in self.sensors:
            return self.sensors[key]
        raise KeyError(f"Scene entity with key '{key}' not found. Available Entities: '{self.keys()}'")

    def keys(self) -> list[str]:
        return list(self.articulations) + list(self.sensors)

    def reset(self, env_ids=None) -> None:
        """interactive_scene.py:440-468: every asset and sensor resets the given envs."""
        for a in self.articulations.values():
            if hasattr(a, "reset"):
                a.reset(env_ids)
        for s in self.sensors.values():
            if hasattr(s, "reset"):
                s.reset(env_ids)

    def write_data_to_sim(self) -> None:
        """interactive_scene.py:470-482: the assets push their buffers (actuator models run here)."""
        for a in self.articulations.values():
            if hasattr(a, "write_data_to_sim"):
                a.write_data_to_sim()

    def _describe(self, dt: float):
        """The one-launch update's descriptor (``ilx_scene``) for a scene of up to one articulation, one contact sensor and
        one ray caster, with the host-side clocks advanced as ``update`` advances them; None for any other scene."""
        arts = list(self.articulations.values())
        contacts = [s for s in self.sensors.values() if isinstance(s, ContactSensor)]
        rays = [s for s in self.sensors.values() if isinstance(s, RayCaster)]
        fusable = len(arts) <= 1 and len(contacts) <= 1 and len(rays) <= 1 and len(contacts) + len(rays) == len(self.sensors)
        if not fusable:
            return None
        force = not self.lazy_sensor_update
        sc = _abi.Scene(num_envs=self.num_envs, dt=float(dt))
        if arts:
            d = getattr(arts[0], "data", arts[0])   # an Articulation or its ArticulationData
            d._sim_timestamp += dt                                   # articulation_data.py:77-82
            sc.joint_vel, sc.previous_joint_vel = _abi.ptr(d.joint_vel), _abi.ptr(d._previous_joint_vel)
            sc.joint_acc, sc.num_joints = _abi.ptr(d._joint_acc), d._j
            sc.joint_acc_elapsed = float(d._sim_timestamp - d._joint_acc_timestamp)
            d._joint_acc_timestamp = d._sim_timestamp
        if contacts:
            c = contacts[0]
            cd, h, t = c._data, c.cfg.history_length, c.cfg.track_air_time
            sc.net_forces_in, sc.net_forces_w = _abi.ptr(c._forces_in), (None if c._skip_net() else _abi.ptr(cd.net_forces_w))
            sc.net_forces_w_history = _abi.ptr(cd.net_forces_w_history) if h > 0 else None
            sc.contact_history_head = None if c._history_head is None else c._history_head.data_ptr()
            if t:
                sc.current_air_time, sc.last_air_time = _abi.ptr(cd.current_air_time), _abi.ptr(cd.last_air_time)
                sc.current_contact_time, sc.last_contact_time = _abi.ptr(cd.current_contact_time), _abi.ptr(cd.last_contact_time)
            sc.contact_timestamp, sc.contact_timestamp_last_update = _abi.ptr(c._timestamp), _abi.ptr(c._timestamp_last_update)
            sc.contact_is_outdated = c._is_outdated.data_ptr()
            sc.contact_update_period, sc.force_threshold = float(c.cfg.update_period), float(c.cfg.force_threshold)
            sc.num_bodies, sc.history_length = c.num_bodies, h
            sc.contact_lazy = int(not (force or h > 0))                # sensor_base.py:203-205
        if rays:
            r = rays[0]
            sc.ray_timestamp, sc.ray_timestamp_last_update = _abi.ptr(r._timestamp), _abi.ptr(r._timestamp_last_update)
            sc.ray_is_outdated, sc.ray_update_period = r._is_outdated.data_ptr(), float(r.cfg.update_period)
        return sc, rays, force

    def describe_update(self, dt: float):
        """``update(dt)`` for a caller that issues the launch itself, together with another one
        (``ActuatorNetMLP.compute(..., scene_update=...)``): the descriptor, or None when this scene's update is not one
        launch (several assets, sensors that refresh eagerly) -- then call ``update``."""
        desc = self._describe(dt)
        if desc is None:
            return None
        sc, rays, force = desc
        if rays and (force or rays[0].cfg.history_length > 0):
            raise _abi.IlxError("describe_update: the ray caster of this scene refreshes with every update; use update()")
        return sc

    def update(self, dt: float, prepare_scanner: bool = False) -> bool:
        """interactive_scene.py:496-513.  ``prepare_scanner`` (no counterpart in the reference): this is the last update
        before the height scanner is read, and the scanner's pose is final (``set_pose``): the launch also issues the
        preparation of the scanner's refresh.  Returns whether it did -- the sensor then knows (``RayCaster._prepared``) that
        its next refresh on this stream, from the step or from a reader of ``scanner.data``, is the cast alone."""
        desc = self._describe(dt)
        if desc is None:
            for a in self.articulations.values():
                a.update(dt)
            for s in self.sensors.values():
                s.update(dt, force_recompute=not self.lazy_sensor_update)
            return False
        sc, rays, force = desc
        eager_rays = bool(rays) and (force or rays[0].cfg.history_length > 0)
        if prepare_scanner and rays and not eager_rays and not rays[0].static_workspace:
            rc = rays[0]._descriptor()
            _abi.check(_abi.scene_update_prepare_raycaster(ctypes.byref(sc), ctypes.byref(rc), _abi.stream()),
                       "ilx_scene_update_prepare_raycaster")
            rays[0]._prepared = True   # consumed by the sensor's next refresh, whoever issues it
            return True
        _abi.check(_abi.scene_update(ctypes.byref(sc), _abi.stream()), "ilx_scene_update")
        if eager_rays:
            rays[0]._update_outdated_buffers()
        return False
