"""2-D pose command terms (interface of ``isaaclab/envs/mdp/commands/pose_2d_command.py``) on the HIP kernels
``ilx_pose2d_command_compute`` / ``ilx_pose2d_command_reset``: one launch per ``compute(dt)`` / ``reset(env_ids)``."""

from __future__ import annotations

import ctypes
from collections.abc import Sequence
from ctypes import c_float, c_int32, c_int64, c_uint64, c_void_p
from dataclasses import dataclass, field

import torch

from .... import _abi
from ....utils import rng as _rng
from ....managers.command_manager import CommandTermCfg


class Pose2dDesc(ctypes.Structure):
    """Mirror of ``ilx_pose2d_command`` (include/isaaclab_amd.h)."""

    _fields_ = [
        ("num_envs", c_int64), ("root_state_w", c_void_p), ("default_root_state", c_void_p), ("env_origins", c_void_p),
        ("valid_targets", c_void_p), ("terrain_levels", c_void_p), ("terrain_types", c_void_p), ("num_terrain_cols", c_int32),
        ("num_patches", c_int32), ("simple_heading", c_int32), ("resampling_time_range", c_float * 2), ("pos_x", c_float * 2),
        ("pos_y", c_float * 2), ("heading", c_float * 2), ("pos_command_w", c_void_p), ("heading_command_w", c_void_p),
        ("pos_command_b", c_void_p), ("heading_command_b", c_void_p), ("time_left", c_void_p), ("command_counter", c_void_p),
        ("error_pos", c_void_p), ("error_heading", c_void_p), ("error_pos_2d", c_void_p), ("samples", c_void_p),
        ("reset_samples", c_void_p), ("patch_ids", c_void_p), ("reset_patch_ids", c_void_p), ("seed", c_uint64), ("draw", c_uint64),
    ]


_compute = _abi._sig("ilx_pose2d_command_compute", ctypes.POINTER(Pose2dDesc), c_float, _abi.P)
_reset = _abi._sig("ilx_pose2d_command_reset", ctypes.POINTER(Pose2dDesc), _abi.P, _abi.P, _abi.P)


@dataclass
class UniformPose2dCommandCfg(CommandTermCfg):
    """commands_cfg.py:189-229."""

    @dataclass
    class Ranges:
        pos_x: tuple[float, float] = (0.0, 0.0)
        pos_y: tuple[float, float] = (0.0, 0.0)
        heading: tuple[float, float] = (0.0, 0.0)

    asset_name: str = "robot"
    simple_heading: bool = False
    ranges: Ranges = field(default_factory=Ranges)

    def __post_init__(self):
        self.class_type = self.class_type or UniformPose2dCommand


class UniformPose2dCommand:
    """pose_2d_command.py:26-158 under command_manager.py:27-222.  ``metrics`` holds ``error_pos`` (present but never
    updated, as in the reference), ``error_heading`` and ``error_pos_2d``.  ``_samples`` / ``_reset_samples`` (N, 4) inject
    the drawn values (time left, x, y, heading) for replay; otherwise the kernel draws with Philox."""

    def __init__(self, cfg: UniformPose2dCommandCfg, env):
        self.cfg, self._env = cfg, env
        self.num_envs, self.device = env.num_envs, torch.device(env.device)
        self.robot = env.scene[cfg.asset_name]
        n, dev = self.num_envs, self.device
        self.pos_command_w = torch.zeros(n, 3, device=dev)
        self.heading_command_w = torch.zeros(n, device=dev)
        self.pos_command_b = torch.zeros(n, 3, device=dev)
        self.heading_command_b = torch.zeros(n, device=dev)
        self.time_left = torch.zeros(n, device=dev)
        self.command_counter = torch.zeros(n, dtype=torch.long, device=dev)
        self.metrics = {"error_pos": torch.zeros(n, device=dev), "error_heading": torch.zeros(n, device=dev),
                        "error_pos_2d": torch.zeros(n, device=dev)}
        self._samples = self._reset_samples = self._patch_ids = self._reset_patch_ids = None
        self._log = torch.zeros(4, device=dev)
        self._mask = torch.zeros(n, dtype=torch.bool, device=dev)

    @property
    def command(self) -> torch.Tensor:
        """(x, y, z) of the target in the base frame and the heading error. Shape is (num_envs, 4)."""
        return torch.cat([self.pos_command_b, self.heading_command_b.unsqueeze(1)], dim=1)

    def _desc(self) -> Pose2dDesc:
        c, r = self.cfg, self.cfg.ranges
        data = getattr(self.robot, "data", self.robot)
        d = Pose2dDesc(num_envs=self.num_envs, root_state_w=_abi.ptr(data.root_state_w),
                       default_root_state=_abi.ptr(data.default_root_state), simple_heading=int(c.simple_heading),
                       pos_command_w=_abi.ptr(self.pos_command_w), heading_command_w=_abi.ptr(self.heading_command_w),
                       pos_command_b=_abi.ptr(self.pos_command_b), heading_command_b=_abi.ptr(self.heading_command_b),
                       time_left=_abi.ptr(self.time_left), command_counter=self.command_counter.data_ptr(),
                       error_pos=_abi.ptr(self.metrics["error_pos"]), error_heading=_abi.ptr(self.metrics["error_heading"]),
                       error_pos_2d=_abi.ptr(self.metrics["error_pos_2d"]), samples=_abi.ptr(self._samples),
                       reset_samples=_abi.ptr(self._reset_samples),
                       patch_ids=None if self._patch_ids is None else self._patch_ids.data_ptr(),
                       reset_patch_ids=None if self._reset_patch_ids is None else self._reset_patch_ids.data_ptr(),
                       seed=_rng.seed(), draw=_rng.next_draw("pose_2d_command"))
        d.resampling_time_range = (c_float * 2)(*c.resampling_time_range)
        d.pos_x, d.pos_y = (c_float * 2)(*r.pos_x), (c_float * 2)(*r.pos_y)
        d.heading = (c_float * 2)(*(r.heading if r.heading is not None else (0.0, 0.0)))
        self._fill_targets(d)
        return d

    def _fill_targets(self, d: Pose2dDesc):
        d.env_origins = _abi.ptr(_abi.f32c(self._env.scene.env_origins))

    def compute(self, dt: float):
        d = self._desc()
        _abi.check(_compute(ctypes.byref(d), float(dt), _abi.stream()), "ilx_pose2d_command_compute")

    def reset(self, env_ids: Sequence[int] | None = None) -> dict[str, float]:
        self._mask.zero_()
        if env_ids is None or (isinstance(env_ids, slice) and env_ids == slice(None)):
            self._mask.fill_(True)
        else:
            self._mask[torch.as_tensor(env_ids, device=self.device, dtype=torch.long)] = True
        self._log.zero_()
        d = self._desc()
        _abi.check(_reset(ctypes.byref(d), self._mask.data_ptr(), _abi.ptr(self._log), _abi.stream()), "ilx_pose2d_command_reset")
        sums = self._log.tolist()
        count = sums[3]
        return {name: (sums[i] / count if count else float("nan")) for i, name in enumerate(("error_pos", "error_heading", "error_pos_2d"))}


@dataclass
class TerrainBasedPose2dCommandCfg(UniformPose2dCommandCfg):
    """commands_cfg.py:232-248."""

    def __post_init__(self):
        self.class_type = self.class_type or TerrainBasedPose2dCommand


class TerrainBasedPose2dCommand(UniformPose2dCommand):
    """pose_2d_command.py:161-206: targets are flat patches of the robot's own sub-terrain
    (``terrain.flat_patches["target"]`` of shape (rows, cols, patches, 3))."""

    def __init__(self, cfg: TerrainBasedPose2dCommandCfg, env):
        super().__init__(cfg, env)
        self.terrain = env.scene["terrain"]
        if "target" not in self.terrain.flat_patches:
            raise RuntimeError("The terrain-based command generator requires a valid flat patch under 'target' in the terrain."
                               f" Found: {list(self.terrain.flat_patches.keys())}")
        self.valid_targets = _abi.f32c(self.terrain.flat_patches["target"])

    def _fill_targets(self, d: Pose2dDesc):
        d.valid_targets = _abi.ptr(self.valid_targets)
        d.terrain_levels, d.terrain_types = self.terrain.terrain_levels.data_ptr(), self.terrain.terrain_types.data_ptr()
        d.num_terrain_cols, d.num_patches = self.valid_targets.shape[1], self.valid_targets.shape[2]
