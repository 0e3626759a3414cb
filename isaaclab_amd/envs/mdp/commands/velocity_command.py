"""``UniformVelocityCommand`` and ``NormalVelocityCommand`` (interface of ``isaaclab/envs/mdp/commands/velocity_command.py``)."""

from __future__ import annotations

import ctypes
from collections.abc import Sequence
from ctypes import c_float, c_int64, c_uint64, c_void_p

import torch

from .... import _abi
from ....utils import rng as _rng
from ....managers.command_manager import CommandTerm


class UniformVelocityCommand(CommandTerm):
    """Base-frame velocity command (x, y, yaw rate) with optional heading control and standing envs.

    ``metrics["error_vel_xy"]`` / ``["error_vel_yaw"]`` accumulate as in velocity_command.py:111-121; resampling draws
    follow :123-138 (lin x, lin y, yaw rate, heading, heading-env coin, standing-env coin) and the post-processing
    :140-160 (heading error -> yaw rate, standing envs -> zero).
    """

    metric_names = ("error_vel_xy", "error_vel_yaw")

    def __init__(self, cfg, env):
        if cfg.heading_command and cfg.ranges.heading is None:  # velocity_command.py:62-66
            raise ValueError("The velocity command has heading commands active (heading_command=True) but the `ranges.heading`"
                             " parameter is set to None.")
        self.robot = env.scene[cfg.asset_name]
        super().__init__(cfg, env)
        m = self._mdp
        self.vel_command_b, self.heading_target = m.command, m.heading_target
        self.is_heading_env, self.is_standing_env = m.is_heading_env, m.is_standing_env

    def _kernel_cfg(self) -> dict:
        c, r = self.cfg, self.cfg.ranges
        heading = r.heading if r.heading is not None else (0.0, 0.0)
        return dict(kind="velocity", command_dim=3, resampling_time_range=tuple(c.resampling_time_range),
                    ranges=[r.lin_vel_x, r.lin_vel_y, r.ang_vel_z, heading],
                    rel_heading_envs=c.rel_heading_envs, rel_standing_envs=c.rel_standing_envs,
                    heading_control_stiffness=c.heading_control_stiffness, heading_command=c.heading_command)

    def _bind_state(self):
        data = getattr(self.robot, "data", self.robot)
        self._mdp.bind(root_state_w=data.root_state_w)

    @property
    def command(self) -> torch.Tensor:
        """The desired base velocity command in the base frame. Shape is (num_envs, 3)."""
        return self.vel_command_b


class NormalVelDesc(ctypes.Structure):
    """Mirror of ``ilx_normal_velocity_command`` (include/isaaclab_amd.h)."""

    _fields_ = [
        ("num_envs", c_int64), ("root_state_w", c_void_p), ("mean_vel", c_float * 3), ("std_vel", c_float * 3),
        ("zero_prob", c_float * 3), ("rel_standing_envs", c_float), ("max_command_step", c_float),
        ("resampling_time_range", c_float * 2), ("vel_command_b", c_void_p), ("is_zero_vel_env", c_void_p),
        ("is_standing_env", c_void_p), ("time_left", c_void_p), ("command_counter", c_void_p), ("error_vel_xy", c_void_p),
        ("error_vel_yaw", c_void_p), ("samples", c_void_p), ("reset_samples", c_void_p), ("seed", c_uint64), ("draw", c_uint64),
    ]


_normal_compute = _abi._sig("ilx_normal_velocity_command_compute", ctypes.POINTER(NormalVelDesc), c_float, _abi.P)
_normal_reset = _abi._sig("ilx_normal_velocity_command_reset", ctypes.POINTER(NormalVelDesc), _abi.P, _abi.P, _abi.P)


class NormalVelocityCommand:
    """velocity_command.py:218-287 under command_manager.py:27-222, one launch per ``compute(dt)`` / ``reset(env_ids)``.

    Each component is drawn from N(mean, std) and its sign flipped by a fair coin; ``is_zero_vel_{x,y,yaw}_env`` and
    ``is_standing_env`` zero components / the whole command after every compute.  ``_samples`` / ``_reset_samples`` (N, 11)
    inject the drawn values (time left, 3 normal values, 3 sign coins, 3 zero coins, standing coin) for replay; otherwise the
    kernel draws with Philox.
    """

    def __init__(self, cfg, env):
        self.cfg, self._env = cfg, env
        self.num_envs, self.device = env.num_envs, torch.device(env.device)
        self.robot = env.scene[cfg.asset_name]
        n, dev = self.num_envs, self.device
        self.vel_command_b = torch.zeros(n, 3, device=dev)
        self._is_zero = torch.zeros(n, 3, dtype=torch.bool, device=dev)
        self.is_zero_vel_x_env, self.is_zero_vel_y_env, self.is_zero_vel_yaw_env = self._is_zero.unbind(1)
        self.is_standing_env = torch.zeros(n, dtype=torch.bool, device=dev)
        self.is_heading_env = torch.zeros(n, dtype=torch.bool, device=dev)   # present on the reference's base class, unused here
        self.heading_target = torch.zeros(n, device=dev)
        self.time_left = torch.zeros(n, device=dev)
        self.command_counter = torch.zeros(n, dtype=torch.long, device=dev)
        self.metrics = {"error_vel_xy": torch.zeros(n, device=dev), "error_vel_yaw": torch.zeros(n, device=dev)}
        self._samples = self._reset_samples = None
        self._log = torch.zeros(3, device=dev)
        self._mask = torch.zeros(n, dtype=torch.bool, device=dev)

    def __str__(self) -> str:
        return ("NormalVelocityCommand:\n"
                f"\tCommand dimension: {tuple(self.command.shape[1:])}\n"
                f"\tResampling time range: {self.cfg.resampling_time_range}\n"
                f"\tStanding probability: {self.cfg.rel_standing_envs}")

    @property
    def command(self) -> torch.Tensor:
        """The desired base velocity command in the base frame. Shape is (num_envs, 3)."""
        return self.vel_command_b

    def _desc(self) -> NormalVelDesc:
        c, r = self.cfg, self.cfg.ranges
        data = getattr(self.robot, "data", self.robot)
        d = NormalVelDesc(num_envs=self.num_envs, root_state_w=_abi.ptr(data.root_state_w),
                          rel_standing_envs=float(c.rel_standing_envs),
                          max_command_step=float(c.resampling_time_range[1] / self._env.step_dt),
                          vel_command_b=_abi.ptr(self.vel_command_b), is_zero_vel_env=self._is_zero.data_ptr(),
                          is_standing_env=self.is_standing_env.data_ptr(), time_left=_abi.ptr(self.time_left),
                          command_counter=self.command_counter.data_ptr(), error_vel_xy=_abi.ptr(self.metrics["error_vel_xy"]),
                          error_vel_yaw=_abi.ptr(self.metrics["error_vel_yaw"]), samples=_abi.ptr(self._samples),
                          reset_samples=_abi.ptr(self._reset_samples), seed=_rng.seed(), draw=_rng.next_draw("velocity_command"))
        d.mean_vel, d.std_vel, d.zero_prob = (c_float * 3)(*r.mean_vel), (c_float * 3)(*r.std_vel), (c_float * 3)(*r.zero_prob)
        d.resampling_time_range = (c_float * 2)(*c.resampling_time_range)
        return d

    def compute(self, dt: float):
        d = self._desc()
        _abi.check(_normal_compute(ctypes.byref(d), float(dt), _abi.stream()), "ilx_normal_velocity_command_compute")

    def reset(self, env_ids: Sequence[int] | None = None) -> dict[str, float]:
        self._mask.zero_()
        if env_ids is None or (isinstance(env_ids, slice) and env_ids == slice(None)):
            self._mask.fill_(True)
        else:
            self._mask[torch.as_tensor(env_ids, device=self.device, dtype=torch.long)] = True
        self._log.zero_()
        d = self._desc()
        _abi.check(_normal_reset(ctypes.byref(d), self._mask.data_ptr(), _abi.ptr(self._log), _abi.stream()),
                   "ilx_normal_velocity_command_reset")
        sums = self._log.tolist()
        return {name: (sums[i] / sums[2] if sums[2] else float("nan")) for i, name in enumerate(("error_vel_xy", "error_vel_yaw"))}
