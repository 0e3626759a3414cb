"""Contact sensor bookkeeping on the HIP kernel ``ilx_contact_update``
(interface of ``isaaclab/sensors/contact_sensor/contact_sensor.py`` + ``contact_sensor_data.py``).

The reference reads net contact forces from a PhysX ``RigidContactView``; here the forces are handed in by the
caller (``set_net_contact_forces``), everything after that read is the same: history ring, air / contact
timers and the per-sensor update clocks of ``SensorBase``.
"""

from __future__ import annotations

import copy
from collections.abc import Sequence
from dataclasses import dataclass, field

import torch

from .. import _abi
from ..utils import string as string_utils


@dataclass
class ContactSensorCfg:
    """Subset of ``contact_sensor_cfg.py`` that affects the bookkeeping."""

    update_period: float = 0.0
    history_length: int = 0
    track_air_time: bool = False
    force_threshold: float = 1.0
    prim_path: str = ""
    debug_vis: bool = False
    filter_prim_paths_expr: list[str] = field(default_factory=list)


@dataclass
class ContactSensorData:
    """contact_sensor_data.py:14-95."""

    pos_w: torch.Tensor | None = None
    quat_w: torch.Tensor | None = None
    net_forces_w: torch.Tensor | None = None
    net_forces_w_history: torch.Tensor | None = None
    force_matrix_w: torch.Tensor | None = None
    last_air_time: torch.Tensor | None = None
    current_air_time: torch.Tensor | None = None
    last_contact_time: torch.Tensor | None = None
    current_contact_time: torch.Tensor | None = None


class ContactSensor:
    def __init__(self, cfg: ContactSensorCfg, num_envs: int, body_names: Sequence[str], device):
        self.cfg = cfg
        self._num_envs = num_envs
        self._body_names = list(body_names)
        self._num_bodies = len(self._body_names)
        self._device = torch.device(device)
        n, b, h = num_envs, self._num_bodies, cfg.history_length
        if len(cfg.filter_prim_paths_expr) != 0:
            raise NotImplementedError("filtered contact force matrices come from the physics engine and are not modelled")
        self._is_outdated = torch.ones(n, dtype=torch.bool, device=self._device)
        self._timestamp = torch.zeros(n, device=self._device)
        self._timestamp_last_update = torch.zeros(n, device=self._device)
        self._data = ContactSensorData()
        self._data.net_forces_w = torch.zeros(n, b, 3, device=self._device)
        # The history is stored as a ring per env (entry of age k at slot (head + k) % h): a refresh writes one row
        # instead of moving h.  `data.net_forces_w_history` hands out the reference's newest-first order.
        self._history_head = torch.zeros(n, dtype=torch.int32, device=self._device) if h > 1 else None
        if h > 0:
            self._data.net_forces_w_history = torch.zeros(n, h, b, 3, device=self._device)
        else:
            self._data.net_forces_w_history = self._data.net_forces_w.unsqueeze(1)
        if cfg.track_air_time:
            self._data.last_air_time = torch.zeros(n, b, device=self._device)
            self._data.current_air_time = torch.zeros(n, b, device=self._device)
            self._data.last_contact_time = torch.zeros(n, b, device=self._device)
            self._data.current_contact_time = torch.zeros(n, b, device=self._device)
        self._forces_in = torch.zeros(n, b, 3, device=self._device)
        # With a history, its newest entry and net_forces_w hold the same values at all times (contact_sensor.py:324-328:
        # one refresh writes both, one reset zeroes both).  True: a refresh leaves the copy into net_forces_w out and
        # `data.net_forces_w` hands out the newest entry -- for steps that read only the history (a seventh of the
        # bytes of a scene update).  `raw_data.net_forces_w` is then stale.
        self.net_forces_from_history = False

    # -- properties (contact_sensor.py:80-130)
    @property
    def num_instances(self) -> int:
        return self._num_envs

    @property
    def num_bodies(self) -> int:
        return self._num_bodies

    @property
    def body_names(self) -> list[str]:
        return self._body_names

    @property
    def device(self):
        return self._device

    @property
    def data(self) -> ContactSensorData:
        self._update_outdated_buffers()
        if self._history_head is None and not self._skip_net():
            return self._data
        d = copy.copy(self._data)
        d.net_forces_w_history = self.ordered_history()
        if self._skip_net():
            d.net_forces_w = d.net_forces_w_history[:, 0]
        return d

    def _skip_net(self) -> bool:
        return self.net_forces_from_history and self.cfg.history_length > 0

    @property
    def raw_data(self) -> ContactSensorData:
        """The buffers as the kernels keep them (history as a ring, see ``history_head``), refreshed like ``data``."""
        self._update_outdated_buffers()
        return self._data

    @property
    def history_head(self) -> torch.Tensor | None:
        """(n,) int32 slot of the newest history entry per env, or None when the history is stored newest-first."""
        return self._history_head

    def ordered_history(self) -> torch.Tensor:
        """``net_forces_w_history`` in the reference's order (index 0 = newest), without touching the update clocks."""
        if self._history_head is None:
            return self._data.net_forces_w_history
        h = self.cfg.history_length
        slots = (self._history_head.long().unsqueeze(1) + torch.arange(h, device=self._device)) % h   # (n, h): age -> slot
        return self._data.net_forces_w_history[torch.arange(self._num_envs, device=self._device).unsqueeze(1), slots]

    def find_bodies(self, name_keys, preserve_order: bool = False):
        """contact_sensor.py:157-167."""
        return string_utils.resolve_matching_names(name_keys, self.body_names, preserve_order)

    # -- physics hand-over
    def set_net_contact_forces(self, net_forces: torch.Tensor):
        """What ``RigidContactView.get_net_contact_forces`` would return this substep: (N*B, 3) or (N, B, 3)."""
        self._forces_in = _abi.f32c(net_forces).view(self._num_envs, self._num_bodies, 3)

    # -- operations
    def reset(self, env_ids: Sequence[int] | None = None):
        """sensor_base.py:182-195 + contact_sensor.py:136-155."""
        if env_ids is None:
            env_ids = slice(None)
        self._timestamp[env_ids] = 0.0
        self._timestamp_last_update[env_ids] = 0.0
        self._is_outdated[env_ids] = True
        self._data.net_forces_w[env_ids] = 0.0
        self._data.net_forces_w_history[env_ids] = 0.0
        if self.cfg.track_air_time:
            self._data.current_air_time[env_ids] = 0.0
            self._data.last_air_time[env_ids] = 0.0
            self._data.current_contact_time[env_ids] = 0.0
            self._data.last_contact_time[env_ids] = 0.0

    def update(self, dt: float, force_recompute: bool = False):
        """sensor_base.py:197-205.  With a history the buffers refresh eagerly, otherwise on the next ``data`` read."""
        if force_recompute or self.cfg.history_length > 0:
            self._launch(dt, True)
        else:
            self._timestamp += dt
            self._is_outdated |= self._timestamp - self._timestamp_last_update + 1e-6 >= self.cfg.update_period

    def _update_outdated_buffers(self):
        self._launch(0.0, False)

    def _launch(self, dt: float, advance_clock: bool):
        d, h = self._data, self.cfg.history_length
        t = self.cfg.track_air_time
        _abi.check(_abi.contact_update(
            _abi.ptr(self._forces_in), None if self._skip_net() else _abi.ptr(d.net_forces_w),
            _abi.ptr(d.net_forces_w_history) if h > 0 else None,
            None if self._history_head is None else self._history_head.data_ptr(), _abi.ptr(d.current_air_time) if t else None, _abi.ptr(d.last_air_time) if t else None,
            _abi.ptr(d.current_contact_time) if t else None, _abi.ptr(d.last_contact_time) if t else None,
            _abi.ptr(self._timestamp), _abi.ptr(self._timestamp_last_update), self._is_outdated.data_ptr(),
            int(advance_clock), float(dt), float(self.cfg.update_period), float(self.cfg.force_threshold), self._num_envs, self._num_bodies,
            h, _abi.stream()), "ilx_contact_update")

    def compute_first_contact(self, dt: float, abs_tol: float = 1.0e-8) -> torch.Tensor:
        """contact_sensor.py:169-203."""
        if not self.cfg.track_air_time:
            raise RuntimeError("The contact sensor is not configured to track contact time."
                               "Please enable the 'track_air_time' in the sensor configuration.")
        t = self.data.current_contact_time
        return (t > 0.0) * (t < (dt + abs_tol))

    def compute_first_air(self, dt: float, abs_tol: float = 1.0e-8) -> torch.Tensor:
        """contact_sensor.py:205-238."""
        if not self.cfg.track_air_time:
            raise RuntimeError("The contact sensor is not configured to track contact time."
                               "Please enable the 'track_air_time' in the sensor configuration.")
        t = self.data.current_air_time
        return (t > 0.0) * (t < (dt + abs_tol))
