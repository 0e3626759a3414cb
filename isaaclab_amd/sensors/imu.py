"""IMU sensor on the HIP kernel ``ilx_imu_update`` (interface of ``isaaclab/sensors/imu/imu.py``, ``imu_cfg.py``,
``imu_data.py``).

The reference reads the link's pose, velocity and centre of mass from a PhysX ``RigidBodyView``; here the caller hands those
buffers in (``set_physics_state``) in the view's own layout, everything after the read is the same: sensor-frame pose,
body-frame velocities, finite-difference accelerations with the gravity bias, and the update clocks of ``SensorBase``.
"""

from __future__ import annotations

import ctypes
from collections.abc import Sequence
from ctypes import c_float, c_int32, c_int64, c_void_p
from dataclasses import dataclass, field

import torch

from .. import _abi


@dataclass
class ImuOffsetCfg:
    """imu_cfg.py:23-31: pose of the sensor frame in the link frame."""

    pos: tuple[float, float, float] = (0.0, 0.0, 0.0)
    rot: tuple[float, float, float, float] = (1.0, 0.0, 0.0, 0.0)


@dataclass
class ImuCfg:
    """imu_cfg.py:18-46."""

    prim_path: str = ""
    update_period: float = 0.0
    history_length: int = 0
    debug_vis: bool = False
    offset: ImuOffsetCfg = field(default_factory=ImuOffsetCfg)
    gravity_bias: tuple[float, float, float] = (0.0, 0.0, 9.81)


@dataclass
class ImuData:
    """imu_data.py:13-50."""

    pos_w: torch.Tensor | None = None
    quat_w: torch.Tensor | None = None
    lin_vel_b: torch.Tensor | None = None
    ang_vel_b: torch.Tensor | None = None
    lin_acc_b: torch.Tensor | None = None
    ang_acc_b: torch.Tensor | None = None


class ImuDesc(ctypes.Structure):
    """Mirror of ``ilx_imu`` (include/isaaclab_amd.h)."""

    _fields_ = [
        ("num_sensors", c_int64), ("transforms", c_void_p), ("velocities", c_void_p), ("coms", c_void_p),
        ("offset_pos", c_float * 3), ("offset_quat", c_float * 4), ("gravity_bias", c_float * 3), ("update_period", c_float),
        ("pos_w", c_void_p), ("quat_w", c_void_p), ("lin_vel_b", c_void_p), ("ang_vel_b", c_void_p), ("lin_acc_b", c_void_p),
        ("ang_acc_b", c_void_p), ("prev_lin_vel_w", c_void_p), ("prev_ang_vel_w", c_void_p), ("timestamp", c_void_p),
        ("timestamp_last_update", c_void_p), ("is_outdated", c_void_p),
    ]


_update = _abi._sig("ilx_imu_update", ctypes.POINTER(ImuDesc), c_int32, c_float, c_float, _abi.P)


class Imu:
    def __init__(self, cfg: ImuCfg, num_envs: int, device):
        self.cfg = cfg
        self._num_envs, self._device = num_envs, torch.device(device)
        n, dev = num_envs, self._device
        self._is_outdated = torch.ones(n, dtype=torch.bool, device=dev)
        self._timestamp = torch.zeros(n, device=dev)
        self._timestamp_last_update = torch.zeros(n, device=dev)
        # imu.py:184-203
        d = self._data = ImuData()
        d.pos_w = torch.zeros(n, 3, device=dev)
        d.quat_w = torch.zeros(n, 4, device=dev)
        d.quat_w[:, 0] = 1.0
        d.lin_vel_b, d.ang_vel_b = torch.zeros(n, 3, device=dev), torch.zeros(n, 3, device=dev)
        d.lin_acc_b, d.ang_acc_b = torch.zeros(n, 3, device=dev), torch.zeros(n, 3, device=dev)
        self._prev_lin_vel_w, self._prev_ang_vel_w = torch.zeros(n, 3, device=dev), torch.zeros(n, 3, device=dev)
        self._transforms = self._velocities = None
        self._coms = torch.zeros(n, 7, device=dev)

    def __str__(self) -> str:
        return (f"Imu sensor @ '{self.cfg.prim_path}': \n\tupdate period (s) : {self.cfg.update_period}\n"
                f"\tnumber of sensors : {self._num_envs}\n")

    @property
    def data(self) -> ImuData:
        self._update_outdated_buffers()
        return self._data

    @property
    def num_instances(self) -> int:
        return self._num_envs

    @property
    def device(self):
        return self._device

    # -- physics hand-over
    def set_physics_state(self, transforms: torch.Tensor, velocities: torch.Tensor, coms: torch.Tensor | None = None):
        """What ``RigidBodyView.get_transforms()`` (N, 7: position, quaternion x,y,z,w), ``get_velocities()`` (N, 6) and
        ``get_coms()`` (N, 7) would return now."""
        self._transforms, self._velocities = _abi.f32c(transforms), _abi.f32c(velocities)
        if coms is not None:
            self._coms = _abi.f32c(coms.to(self._device))

    # -- operations
    def reset(self, env_ids: Sequence[int] | None = None):
        """sensor_base.py:182-195 + imu.py:92-103 (the orientation is zeroed too, as in the reference)."""
        if env_ids is None:
            env_ids = slice(None)
        self._timestamp[env_ids] = 0.0
        self._timestamp_last_update[env_ids] = 0.0
        self._is_outdated[env_ids] = True
        d = self._data
        for buf in (d.quat_w, d.lin_vel_b, d.ang_vel_b, d.lin_acc_b, d.ang_acc_b):
            buf[env_ids] = 0.0

    def update(self, dt: float, force_recompute: bool = False):
        """imu.py:105-109 + sensor_base.py:197-205."""
        self._dt = dt
        if force_recompute or self.cfg.history_length > 0:
            self._launch(dt, True)
        else:
            self._timestamp += dt
            self._is_outdated |= self._timestamp - self._timestamp_last_update + 1e-6 >= self.cfg.update_period

    def _update_outdated_buffers(self):
        self._launch(0.0, False)

    def _launch(self, dt_clock: float, advance_clock: bool):
        if not hasattr(self, "_dt"):   # imu.py:144-148
            raise RuntimeError("The update function must be called before the data buffers are accessed the first time.")
        if self._transforms is None:
            raise RuntimeError("The Imu sensor has no physics state: call set_physics_state() before reading data.")
        d, c = self._data, self.cfg
        desc = ImuDesc(num_sensors=self._num_envs, transforms=_abi.ptr(self._transforms), velocities=_abi.ptr(self._velocities),
                       coms=_abi.ptr(self._coms), update_period=float(c.update_period), pos_w=_abi.ptr(d.pos_w), quat_w=_abi.ptr(d.quat_w),
                       lin_vel_b=_abi.ptr(d.lin_vel_b), ang_vel_b=_abi.ptr(d.ang_vel_b), lin_acc_b=_abi.ptr(d.lin_acc_b),
                       ang_acc_b=_abi.ptr(d.ang_acc_b), prev_lin_vel_w=_abi.ptr(self._prev_lin_vel_w),
                       prev_ang_vel_w=_abi.ptr(self._prev_ang_vel_w), timestamp=_abi.ptr(self._timestamp),
                       timestamp_last_update=_abi.ptr(self._timestamp_last_update), is_outdated=self._is_outdated.data_ptr())
        desc.offset_pos, desc.offset_quat = (c_float * 3)(*c.offset.pos), (c_float * 4)(*c.offset.rot)
        desc.gravity_bias = (c_float * 3)(*c.gravity_bias)
        _abi.check(_update(ctypes.byref(desc), int(advance_clock), float(dt_clock), float(self._dt), _abi.stream()), "ilx_imu_update")
