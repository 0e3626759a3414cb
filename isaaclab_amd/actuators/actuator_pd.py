"""PD-family actuators on the HIP kernel ``ilx_actuator_pd`` (interface of ``isaaclab/actuators/actuator_pd.py``).

One launch per ``compute`` replaces the 6 (ideal PD) to 14 (DC motor) eager kernels of the reference.
"""

from __future__ import annotations

from collections.abc import Sequence

import ctypes

import torch

from .. import _abi
from ..utils.types import ArticulationActions
from .actuator_base import ActuatorBase

_IDEAL, _DC = 0, 1


def _launch_pd(model, act, control_action, joint_pos, joint_vel, vel_limit, sat, scene_update=None, joint_action=None):
    if joint_action is not None:
        return _launch_pd_joint_action(model, act, control_action, joint_pos, joint_vel, vel_limit, sat, joint_action)
    q_des = _abi.f32c(control_action.joint_positions)
    qd_des = None if control_action.joint_velocities is None else _abi.f32c(control_action.joint_velocities)
    tau_ff = None if control_action.joint_efforts is None else _abi.f32c(control_action.joint_efforts)
    q, qd = _abi.f32c(joint_pos), _abi.f32c(joint_vel)
    n, j = q.shape
    args = (model, _abi.ptr(q_des), _abi.ptr(qd_des), _abi.ptr(tau_ff), _abi.ptr(q), _abi.ptr(qd),
            _abi.ptr(act.stiffness), _abi.ptr(act.damping), _abi.ptr(act.effort_limit),
            None if vel_limit is None else _abi.ptr(vel_limit), 1, float(sat), _abi.ptr(act.computed_effort),
            _abi.ptr(act.applied_effort), n, j)
    if scene_update is not None:   # the scene update of the substep before travels with this launch
        _abi.check(_abi.actuator_pd_scene_update(*args, ctypes.byref(scene_update), _abi.stream()), "ilx_actuator_pd_scene_update")
    else:
        _abi.check(_abi.actuator_pd(*args, _abi.stream()), "ilx_actuator_pd")


def _launch_pd_joint_action(model, act, control_action, joint_pos, joint_vel, vel_limit, sat, ja):
    """The first substep's launch with the joint action term on it (``ilx_actuator_pd_joint_action``).  ``ja``: ``raw``,
    ``action``, ``prev_action`` (n, width), ``scale``, ``offset`` ((n, width) or None), ``target`` ("position" / "effort"),
    ``first_joint``, ``width`` and ``begin`` (``FusedMdp.begin_step_args`` or None).  The targets are written in place:
    they must be the articulation's own (n, j) buffers."""
    q_des, tau_ff = control_action.joint_positions, control_action.joint_efforts
    for t in (q_des, tau_ff):
        if t is not None and not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise _abi.IlxError("a joint action that rides on the actuator launch needs contiguous float32 GPU target buffers")
    qd_des = None if control_action.joint_velocities is None else _abi.f32c(control_action.joint_velocities)
    q, qd = _abi.f32c(joint_pos), _abi.f32c(joint_vel)
    n, j = q.shape
    begin = ja.get("begin") or (None, 0, None)
    _abi.check(_abi.actuator_pd_joint_action(
        model, _abi.ptr(_abi.f32c(ja["raw"])), _abi.ptr(ja["action"]), _abi.ptr(ja["prev_action"]), float(ja["scale"]),
        None if ja.get("offset") is None else _abi.ptr(ja["offset"]), {"position": 0, "effort": 1}[ja["target"]],
        int(ja["first_joint"]), int(ja["width"]), _abi.ptr(q_des), _abi.ptr(qd_des), _abi.ptr(tau_ff), _abi.ptr(q), _abi.ptr(qd),
        _abi.ptr(act.stiffness), _abi.ptr(act.damping), _abi.ptr(act.effort_limit),
        None if vel_limit is None else _abi.ptr(vel_limit), 1, float(sat), _abi.ptr(act.computed_effort),
        _abi.ptr(act.applied_effort), n, j, *begin, _abi.stream()), "ilx_actuator_pd_joint_action")


class ImplicitActuator(ActuatorBase):
    """actuator_pd.py:32-87: targets pass through; the PD torque is only estimated (for reward terms)."""

    def reset(self, *args, **kwargs):
        pass

    def compute(self, control_action, joint_pos, joint_vel, scene_update=None, joint_action=None):
        _launch_pd(_IDEAL, self, control_action, joint_pos, joint_vel, None, 0.0, scene_update, joint_action)
        return control_action


class IdealPDActuator(ActuatorBase):
    """actuator_pd.py:95-145: explicit PD torque with a symmetric effort clip."""

    def reset(self, env_ids: Sequence[int]):
        pass

    def compute(self, control_action, joint_pos, joint_vel, scene_update=None, joint_action=None):
        """``scene_update`` (no counterpart in the reference): the descriptor of the scene update of the substep before
        (``InteractiveScene.describe_update``), which then travels with this launch.  ``joint_action`` (likewise): the
        joint action term whose processing rides on the step's first launch (``_launch_pd_joint_action``)."""
        _launch_pd(_IDEAL, self, control_action, joint_pos, joint_vel, None, 0.0, scene_update, joint_action)
        control_action.joint_efforts = self.applied_effort
        control_action.joint_positions = None
        control_action.joint_velocities = None
        return control_action


class DCMotor(IdealPDActuator):
    """actuator_pd.py:148-233: PD torque clipped by the velocity dependent DC-motor envelope."""

    def __init__(self, cfg, *args, **kwargs):
        super().__init__(cfg, *args, **kwargs)
        self._saturation_effort = self.cfg.saturation_effort if self.cfg.saturation_effort is not None else torch.inf
        self._joint_vel = torch.zeros_like(self.computed_effort)
        if self.cfg.velocity_limit is None:
            raise ValueError("The velocity limit must be provided for the DC motor actuator model.")

    def compute(self, control_action, joint_pos, joint_vel, scene_update=None, joint_action=None):
        self._joint_vel = joint_vel  # kept for API compatibility; the kernel reads joint_vel directly
        _launch_pd(_DC, self, control_action, joint_pos, joint_vel, self.velocity_limit, self._saturation_effort, scene_update,
                   joint_action)
        control_action.joint_efforts = self.applied_effort
        control_action.joint_positions = None
        control_action.joint_velocities = None
        return control_action


class DelayBuffer:
    """State of one ``isaaclab.utils.buffers.DelayBuffer`` (delay_buffer.py:32-177) as the fused kernel
    ``ilx_actuator_delayed_pd`` reads and writes it: the ring ``(history_length + 1, batch, width)``, the per-env time
    lags and push counts.  The append + delayed read itself happens inside the actuator's launch."""

    def __init__(self, history_length: int, batch_size: int, device):
        self._history_length = max(0, history_length)
        self._batch_size, self._device = batch_size, torch.device(device)
        self._time_lags = torch.zeros(batch_size, dtype=torch.int, device=self._device)
        self._num_pushes = torch.zeros(batch_size, dtype=torch.long, device=self._device)
        self._min_time_lag = self._max_time_lag = 0
        self._buffer = None

    batch_size = property(lambda self: self._batch_size)
    device = property(lambda self: self._device)
    history_length = property(lambda self: self._history_length)
    min_time_lag = property(lambda self: self._min_time_lag)
    max_time_lag = property(lambda self: self._max_time_lag)
    time_lags = property(lambda self: self._time_lags)

    def ring(self, width: int) -> torch.Tensor:
        if self._buffer is None:
            self._buffer = torch.zeros(self._history_length + 1, self._batch_size, width, device=self._device)
        return self._buffer

    def set_time_lag(self, time_lag, batch_ids=None):
        """delay_buffer.py:105-147."""
        if batch_ids is None:
            batch_ids = slice(None)
        if isinstance(time_lag, int):
            self._time_lags[batch_ids] = time_lag
        elif isinstance(time_lag, torch.Tensor):
            if time_lag.dtype not in [torch.int, torch.long]:
                raise TypeError(f"Invalid dtype for time_lag: {time_lag.dtype}. Expected torch.int or torch.long.")
            self._time_lags[batch_ids] = time_lag.to(device=self._device, dtype=torch.int)
        else:
            raise TypeError(f"Invalid type for time_lag: {type(time_lag)}. Expected int or integer tensor.")
        self._min_time_lag = int(torch.min(self._time_lags).item())
        self._max_time_lag = int(torch.max(self._time_lags).item())
        if self._min_time_lag < 0:
            raise ValueError(f"The minimum time lag cannot be negative. Received: {self._min_time_lag}")
        if self._max_time_lag > self._history_length:
            raise ValueError(f"The maximum time lag cannot be larger than the history length. Received: {self._max_time_lag}")

    def reset(self, batch_ids=None):
        """delay_buffer.py:149-156 / circular_buffer.py:93-107."""
        if batch_ids is None:
            batch_ids = slice(None)
        self._num_pushes[batch_ids] = 0
        if self._buffer is not None:
            self._buffer[:, batch_ids] = 0.0


class DelayedPDActuator(IdealPDActuator):
    """actuator_pd.py:236-291: ideal PD on set-points that arrive a per-env number of physics steps late."""

    _lookup = None

    def __init__(self, cfg, *args, **kwargs):
        super().__init__(cfg, *args, **kwargs)
        self.positions_delay_buffer = DelayBuffer(cfg.max_delay, self._num_envs, device=self._device)
        self.velocities_delay_buffer = DelayBuffer(cfg.max_delay, self._num_envs, device=self._device)
        self.efforts_delay_buffer = DelayBuffer(cfg.max_delay, self._num_envs, device=self._device)
        self._buffers = (self.positions_delay_buffer, self.velocities_delay_buffer, self.efforts_delay_buffer)
        # the three buffers are appended to and reset together: one push count serves the kernel
        for b in self._buffers[1:]:
            b._num_pushes = self.positions_delay_buffer._num_pushes
        self._pointer = -1
        self._ALL_INDICES = torch.arange(self._num_envs, dtype=torch.long, device=self._device)

    def reset(self, env_ids: Sequence[int]):
        super().reset(env_ids)
        if env_ids is None or (isinstance(env_ids, slice) and env_ids == slice(None)):
            num_envs = self._num_envs
        else:
            num_envs = len(env_ids)
        time_lags = torch.randint(low=self.cfg.min_delay, high=self.cfg.max_delay + 1, size=(num_envs,), dtype=torch.int,
                                  device=self._device)
        for b in self._buffers:
            b.set_time_lag(time_lags, env_ids)
        for b in self._buffers:
            b.reset(env_ids)

    def compute(self, control_action, joint_pos, joint_vel):
        q_des = _abi.f32c(control_action.joint_positions)
        qd_des = None if control_action.joint_velocities is None else _abi.f32c(control_action.joint_velocities)
        tau_ff = None if control_action.joint_efforts is None else _abi.f32c(control_action.joint_efforts)
        q, qd = _abi.f32c(joint_pos), _abi.f32c(joint_vel)
        n, j = q.shape
        ring_len = self.cfg.max_delay + 1
        self._pointer = (self._pointer + 1) % ring_len
        pb, vb, eb = self._buffers
        lx, ly = (None, None) if self._lookup is None else self._lookup
        _abi.check(_abi.actuator_delayed_pd(
            _abi.ptr(q_des), _abi.ptr(qd_des), _abi.ptr(tau_ff), _abi.ptr(q), _abi.ptr(qd), _abi.ptr(self.stiffness),
            _abi.ptr(self.damping), _abi.ptr(self.effort_limit), 1, _abi.ptr(pb.ring(j)),
            None if qd_des is None else _abi.ptr(vb.ring(j)), None if tau_ff is None else _abi.ptr(eb.ring(j)), ring_len,
            self._pointer, pb._num_pushes.data_ptr(), pb.time_lags.data_ptr(), _abi.ptr(lx), _abi.ptr(ly),
            0 if lx is None else lx.numel(), _abi.ptr(self.computed_effort), _abi.ptr(self.applied_effort), n, j, _abi.stream()),
            "ilx_actuator_delayed_pd")
        control_action.joint_efforts = self.applied_effort
        control_action.joint_positions = None
        control_action.joint_velocities = None
        return control_action


class RemotizedPDActuator(DelayedPDActuator):
    """actuator_pd.py:294-363: delayed PD whose torque limit depends on the joint angle through a lookup table."""

    def __init__(self, cfg, joint_names, joint_ids, num_envs, device, stiffness=0.0, damping=0.0, armature=0.0, friction=0.0,
                 effort_limit=torch.inf, velocity_limit=torch.inf):
        cfg.effort_limit = torch.inf
        cfg.velocity_limit = torch.inf
        super().__init__(cfg, joint_names, joint_ids, num_envs, device, stiffness, damping, armature, friction, torch.inf, torch.inf)
        self._joint_parameter_lookup = torch.tensor(cfg.joint_parameter_lookup, device=device, dtype=torch.float32)
        x, y = self.angle_samples.contiguous(), self.max_torque_samples.contiguous()
        if x.numel() == 0:
            raise ValueError("Input tensor x is empty!")
        if torch.any(x[1:] < x[:-1]):
            raise ValueError("Input tensor x is not sorted in ascending order!")
        self._lookup = (x, y)

    @property
    def angle_samples(self) -> torch.Tensor:
        return self._joint_parameter_lookup[:, 0]

    @property
    def transmission_ratio_samples(self) -> torch.Tensor:
        return self._joint_parameter_lookup[:, 1]

    @property
    def max_torque_samples(self) -> torch.Tensor:
        return self._joint_parameter_lookup[:, 2]
