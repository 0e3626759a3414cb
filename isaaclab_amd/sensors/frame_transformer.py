"""FrameTransformer sensor on the HIP kernel ``ilx_frame_transformer_update`` (interface of
``isaaclab/sensors/frame_transformer/frame_transformer.py``, ``frame_transformer_cfg.py``, ``frame_transformer_data.py``).

The reference finds the tracked rigid bodies on the USD stage and reads their poses from a PhysX ``RigidBodyView``; here the
caller names the view's rows (``view_prim_paths``, in the view's own order) and hands in its transforms
(``set_physics_state``).  The resolution of frames to bodies and rows follows the reference; its chain of gathers is composed
once into one row index per (env, frame), and a refresh is one launch.
"""

from __future__ import annotations

import ctypes
import re
from collections.abc import Sequence
from ctypes import c_int32, c_int64, c_void_p
from dataclasses import dataclass, field

import torch

from .. import _abi
from ..utils.math import is_identity_pose


@dataclass
class OffsetCfg:
    """frame_transformer_cfg.py:15-22."""

    pos: tuple[float, float, float] = (0.0, 0.0, 0.0)
    rot: tuple[float, float, float, float] = (1.0, 0.0, 0.0, 0.0)


@dataclass
class FrameTransformerCfg:
    """frame_transformer_cfg.py:25-68."""

    @dataclass
    class FrameCfg:
        prim_path: str = ""
        name: str | None = None
        offset: OffsetCfg = field(default_factory=OffsetCfg)

    prim_path: str = ""
    update_period: float = 0.0
    history_length: int = 0
    debug_vis: bool = False
    source_frame_offset: OffsetCfg = field(default_factory=OffsetCfg)
    target_frames: list = field(default_factory=list)


@dataclass
class FrameTransformerData:
    """frame_transformer_data.py:12-59."""

    target_frame_names: list[str] | None = None
    target_pos_source: torch.Tensor | None = None
    target_quat_source: torch.Tensor | None = None
    target_pos_w: torch.Tensor | None = None
    target_quat_w: torch.Tensor | None = None
    source_pos_w: torch.Tensor | None = None
    source_quat_w: torch.Tensor | None = None


class FrameTransformerDesc(ctypes.Structure):
    """Mirror of ``ilx_frame_transformer`` (include/isaaclab_amd.h)."""

    _fields_ = [
        ("num_envs", c_int64), ("num_frames", c_int32), ("transforms", c_void_p), ("source_rows", c_void_p), ("target_rows", c_void_p),
        ("source_offset_pos", c_void_p), ("source_offset_quat", c_void_p), ("target_offset_pos", c_void_p),
        ("target_offset_quat", c_void_p), ("source_pos_w", c_void_p), ("source_quat_w", c_void_p), ("target_pos_w", c_void_p),
        ("target_quat_w", c_void_p), ("target_pos_source", c_void_p), ("target_quat_source", c_void_p),
    ]


_update = _abi._sig("ilx_frame_transformer_update", ctypes.POINTER(FrameTransformerDesc), _abi.P)


def _env_key(path: str):
    m = re.search(r"env_(\d+)(.*)", path)
    return (int(m.group(1)), m.group(2))


class FrameTransformer:
    def __init__(self, cfg: FrameTransformerCfg, num_envs: int, view_prim_paths: Sequence[str], device):
        self.cfg = cfg
        self._num_envs, self._device = num_envs, torch.device(device)
        n, dev = num_envs, self._device
        self._is_outdated = torch.ones(n, dtype=torch.bool, device=dev)
        self._timestamp = torch.zeros(n, device=dev)
        self._timestamp_last_update = torch.zeros(n, device=dev)
        self._data = FrameTransformerData()
        self._transforms = None
        all_prim_paths = list(view_prim_paths)

        # -- source offset (frame_transformer.py:104-118)
        src = cfg.source_frame_offset
        self._apply_source_frame_offset = not is_identity_pose(torch.tensor([src.pos], device=dev), torch.tensor([src.rot], device=dev))
        if self._apply_source_frame_offset:
            self._source_frame_offset_pos = torch.tensor(src.pos, device=dev).unsqueeze(0).repeat(n, 1)
            self._source_frame_offset_quat = torch.tensor(src.rot, device=dev).unsqueeze(0).repeat(n, 1)

        # -- frames -> bodies (:120-200); a body's frames keep their declaration order
        body_names_to_frames: dict[str, dict] = {}
        target_offsets: dict[str, OffsetCfg] = {}
        self._apply_target_frame_offset = False
        self._source_is_also_target_frame = False
        first_env = [p for p in all_prim_paths if "env_0/" in p or "env_" not in p]
        frames = [(None, cfg.prim_path, None, "source")] + [(t.name, t.prim_path, t.offset, "target") for t in cfg.target_frames]
        for frame, prim_path, offset, frame_type in frames:
            matching = [p for p in first_env if re.fullmatch(prim_path, p)]
            if len(matching) == 0:
                raise ValueError(f"Failed to create frame transformer for frame '{frame}' with path '{prim_path}'."
                                 " No matching prims were found.")
            for path in matching:
                body_name = path.rsplit("/", 1)[-1]
                frame_name = frame if frame is not None else body_name
                if body_name in body_names_to_frames:
                    body_names_to_frames[body_name]["frames"].setdefault(frame_name)
                    if body_names_to_frames[body_name]["type"] == "source" and frame_type == "target":
                        self._source_is_also_target_frame = True
                else:
                    body_names_to_frames[body_name] = {"frames": {frame_name: None}, "prim_path": path, "type": frame_type}
                if offset is not None:
                    if not is_identity_pose(torch.tensor([offset.pos], device=dev), torch.tensor([offset.rot], device=dev)):
                        self._apply_target_frame_offset = True
                    target_offsets[frame_name] = offset
        tracked_body_names = list(body_names_to_frames)

        # -- view order -> env order (:215-250)
        if "env_" in all_prim_paths[0]:
            per_env = [i for i, _ in sorted(enumerate(all_prim_paths), key=lambda x: _env_key(x[1]))]
            sorted_prim_paths = [all_prim_paths[i] for i in per_env if "env_0" in all_prim_paths[i]]
        else:
            per_env = [i for i, _ in sorted(enumerate(all_prim_paths), key=lambda x: x[1])]
            sorted_prim_paths = [all_prim_paths[i] for i in per_env]
        if len(all_prim_paths) != n * len(tracked_body_names):
            raise ValueError(f"the view has {len(all_prim_paths)} rows for {n} envs x {len(tracked_body_names)} tracked bodies")
        self._per_env_indices = torch.tensor(per_env, dtype=torch.long, device=dev)

        # -- bodies, frames, duplicates (:252-311)
        self._target_frame_body_names = [p.split("/")[-1] for p in sorted_prim_paths]
        self._source_frame_body_name = cfg.prim_path.split("/")[-1]
        source_frame_index = self._target_frame_body_names.index(self._source_frame_body_name)
        if not self._source_is_also_target_frame:
            self._target_frame_body_names.remove(self._source_frame_body_name)
        b = len(tracked_body_names)
        all_ids = torch.arange(n * b, device=dev)
        self._source_frame_body_ids = torch.arange(n, device=dev) * b + source_frame_index
        if self._source_is_also_target_frame:
            self._target_frame_body_ids = all_ids
        else:
            self._target_frame_body_ids = all_ids[~torch.isin(all_ids, self._source_frame_body_ids)]
        self._target_frame_names: list[str] = []
        offset_pos, offset_quat, duplicate = [], [], []
        for i, body_name in enumerate(self._target_frame_body_names):
            for frame in body_names_to_frames[body_name]["frames"]:
                if frame in target_offsets:
                    offset_pos.append(target_offsets[frame].pos), offset_quat.append(target_offsets[frame].rot)
                    self._target_frame_names.append(frame)
                    duplicate.append(i)
        f = len(duplicate)
        num_target_body_frames = b if self._source_is_also_target_frame else b - 1
        duplicate = torch.tensor(duplicate, dtype=torch.long, device=dev)
        self._duplicate_frame_indices = torch.cat([duplicate + num_target_body_frames * e for e in range(n)]) if n else duplicate
        if self._apply_target_frame_offset:
            self._target_frame_offset_pos = torch.tensor(offset_pos, device=dev).repeat(n, 1)
            self._target_frame_offset_quat = torch.tensor(offset_quat, device=dev).repeat(n, 1)
        # the gathers of a refresh (:333, 339, 353-355) composed into one row index per env / per (env, frame)
        self._source_rows = self._per_env_indices[self._source_frame_body_ids].contiguous()
        self._target_rows = self._per_env_indices[self._target_frame_body_ids[self._duplicate_frame_indices]].contiguous()

        d = self._data
        d.target_frame_names = self._target_frame_names
        d.source_pos_w, d.source_quat_w = torch.zeros(n, 3, device=dev), torch.zeros(n, 4, device=dev)
        d.target_pos_w, d.target_quat_w = torch.zeros(n, f, 3, device=dev), torch.zeros(n, f, 4, device=dev)
        d.target_pos_source, d.target_quat_source = torch.zeros(n, f, 3, device=dev), torch.zeros(n, f, 4, device=dev)

    def __str__(self) -> str:
        return (f"FrameTransformer @ '{self.cfg.prim_path}': \n\ttracked body frames: "
                f"{[self._source_frame_body_name] + self._target_frame_body_names} \n\tnumber of envs: {self._num_envs}\n"
                f"\tsource body frame: {self._source_frame_body_name}\n\ttarget frames (count: {self._target_frame_names}): "
                f"{len(self._target_frame_names)}\n")

    @property
    def data(self) -> FrameTransformerData:
        self._update_outdated_buffers()
        return self._data

    @property
    def num_instances(self) -> int:
        return self._num_envs

    @property
    def device(self):
        return self._device

    def set_physics_state(self, transforms: torch.Tensor):
        """What ``RigidBodyView.get_transforms()`` would return now: (rows, 7) position + quaternion x,y,z,w, rows in the order
        of ``view_prim_paths``."""
        self._transforms = _abi.f32c(transforms)
        if self._transforms.shape[0] != self._per_env_indices.numel():
            raise ValueError(f"{self._transforms.shape[0]} transforms for a view of {self._per_env_indices.numel()} rows")

    def reset(self, env_ids: Sequence[int] | None = None):
        """sensor_base.py:182-195 (the sensor keeps no other state, frame_transformer.py:90-95)."""
        if env_ids is None:
            env_ids = slice(None)
        self._timestamp[env_ids] = 0.0
        self._timestamp_last_update[env_ids] = 0.0
        self._is_outdated[env_ids] = True

    def update(self, dt: float, force_recompute: bool = False):
        """sensor_base.py:197-205."""
        self._timestamp += dt
        self._is_outdated |= self._timestamp - self._timestamp_last_update + 1e-6 >= self.cfg.update_period
        if force_recompute or self.cfg.history_length > 0:
            self._update_outdated_buffers()

    def _update_outdated_buffers(self):
        """sensor_base.py:284-293; a refresh rewrites every env (frame_transformer.py:378-385)."""
        if not bool(self._is_outdated.any()):
            return
        if self._transforms is None:
            raise RuntimeError("The FrameTransformer has no physics state: call set_physics_state() before reading data.")
        d = self._data
        desc = FrameTransformerDesc(
            num_envs=self._num_envs, num_frames=len(self._target_frame_names), transforms=_abi.ptr(self._transforms),
            source_rows=self._source_rows.data_ptr(), target_rows=self._target_rows.data_ptr(),
            source_offset_pos=_abi.ptr(self._source_frame_offset_pos) if self._apply_source_frame_offset else None,
            source_offset_quat=_abi.ptr(self._source_frame_offset_quat) if self._apply_source_frame_offset else None,
            target_offset_pos=_abi.ptr(self._target_frame_offset_pos) if self._apply_target_frame_offset else None,
            target_offset_quat=_abi.ptr(self._target_frame_offset_quat) if self._apply_target_frame_offset else None,
            source_pos_w=_abi.ptr(d.source_pos_w), source_quat_w=_abi.ptr(d.source_quat_w), target_pos_w=_abi.ptr(d.target_pos_w),
            target_quat_w=_abi.ptr(d.target_quat_w), target_pos_source=_abi.ptr(d.target_pos_source),
            target_quat_source=_abi.ptr(d.target_quat_source))
        _abi.check(_update(ctypes.byref(desc), _abi.stream()), "ilx_frame_transformer_update")
        self._timestamp_last_update.copy_(torch.where(self._is_outdated, self._timestamp, self._timestamp_last_update))
        self._is_outdated.zero_()
