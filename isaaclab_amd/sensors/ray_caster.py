"""RayCaster sensor on the fused HIP kernel ``ilx_raycaster_update``
(interface of ``isaaclab/sensors/ray_caster/ray_caster.py`` + ``ray_caster_data.py`` + ``ray_caster_cfg.py``).

The reference reads the sensor pose from a PhysX view and the mesh from the USD stage; here the owner passes the
pose (``set_pose`` / ``set_root_transforms``) and a :class:`isaaclab_amd.utils.mesh.Mesh`.  Pose -> world rays
-> BVH cast -> ``ray_hits_w`` is one launch (the reference: ~25 eager kernels over N*R rows, a Warp launch and
a device-wide synchronise).
"""

from __future__ import annotations

import ctypes
from collections.abc import Sequence
from dataclasses import dataclass, field

import torch

from .. import _abi
from ..utils import math as math_utils
from ..utils.mesh import Mesh
from .patterns import GridPatternCfg, PatternBaseCfg


@dataclass
class OffsetCfg:
    pos: tuple[float, float, float] = (0.0, 0.0, 0.0)
    rot: tuple[float, float, float, float] = (1.0, 0.0, 0.0, 0.0)


@dataclass
class RayCasterCfg:
    """ray_caster_cfg.py."""

    prim_path: str = ""
    update_period: float = 0.0
    history_length: int = 0
    debug_vis: bool = False
    mesh_prim_paths: list[str] = field(default_factory=lambda: ["/World/ground"])
    offset: OffsetCfg = field(default_factory=OffsetCfg)
    attach_yaw_only: bool = False
    pattern_cfg: PatternBaseCfg = field(default_factory=GridPatternCfg)
    max_distance: float = 1e6
    drift_range: tuple[float, float] = (0.0, 0.0)


class RayCasterData:
    """ray_caster_data.py:10-30.  ``ray_hits_w`` is a property: a sensor that refreshes with ``lean_hits`` leaves the hit
    records to the first reader (they are formed then, from ``pos_w`` / ``quat_w`` of the refresh, by the same device
    code -- the same bits as an eager refresh stores)."""

    def __init__(self, pos_w: torch.Tensor = None, quat_w: torch.Tensor = None, ray_hits_w: torch.Tensor = None):
        self.pos_w, self.quat_w = pos_w, quat_w
        self.hits_storage = ray_hits_w     # the (N, R, 3) buffer itself, whatever its state
        self._bring_up_to_date = None      # set by the owning sensor

    @property
    def ray_hits_w(self) -> torch.Tensor:
        if self._bring_up_to_date is not None:
            self._bring_up_to_date()
        return self.hits_storage

    @ray_hits_w.setter
    def ray_hits_w(self, value: torch.Tensor):
        self.hits_storage = value


class RayCaster:
    def __init__(self, cfg: RayCasterCfg, num_envs: int, mesh: Mesh, device):
        self.cfg = cfg
        self._num_envs = num_envs
        self._device = torch.device(device)
        if len(cfg.mesh_prim_paths) != 1:
            raise NotImplementedError(f"RayCaster currently only supports one mesh prim. Received: {len(cfg.mesh_prim_paths)}")
        self.meshes = {cfg.mesh_prim_paths[0]: mesh}
        n, dev = num_envs, self._device
        # sensor_base.py clocks
        self._is_outdated = torch.ones(n, dtype=torch.bool, device=dev)
        self._timestamp = torch.zeros(n, device=dev)
        self._timestamp_last_update = torch.zeros(n, device=dev)
        # ray_caster.py:201-218
        self.ray_starts, self.ray_directions = cfg.pattern_cfg.func(cfg.pattern_cfg, dev)
        self.num_rays = len(self.ray_directions)
        offset_pos = torch.tensor(list(cfg.offset.pos), device=dev)
        offset_quat = torch.tensor(list(cfg.offset.rot), device=dev)
        self.ray_directions = math_utils.quat_apply(offset_quat.repeat(self.num_rays, 1), self.ray_directions)
        self.ray_starts = self.ray_starts + offset_pos
        # z-parallel world rays (a yaw-only scanner looking straight down) take the column-grid kernel
        self._hints = 0
        if cfg.attach_yaw_only and bool((self.ray_directions[:, :2] == 0).all()):
            # ... and the check just made is the promise that lets the library drop the launch for other rays
            self._hints = _abi.RAYS_Z_PARALLEL | _abi.RAYS_Z_PARALLEL_CHECKED
        # the pattern tensors are never written in place after this point (a new pattern means new tensors): the library
        # may keep the packed pattern and its bounding box from one refresh to the next
        self._hints |= _abi.RAYS_PATTERN_STATIC
        self.drift = torch.zeros(n, 3, device=dev)
        self._has_drift = cfg.drift_range[0] != 0.0 or cfg.drift_range[1] != 0.0
        self._data = RayCasterData(pos_w=torch.zeros(n, 3, device=dev), quat_w=torch.zeros(n, 4, device=dev),
                                   ray_hits_w=torch.zeros(n, self.num_rays, 3, device=dev))
        self._pose_pos = torch.zeros(n, 3, device=dev)
        self._pose_quat = torch.zeros(n, 4, device=dev)
        self._pose_quat[:, 0] = 1.0
        # True: refresh through the entry point that allocates nothing (a buffer built into the library, shared by all
        # calls on the GPU, limited to 32,768 envs); for callers under graph capture or otherwise unable to allocate
        self.static_workspace = False
        # The scene update that ran last also issued this sensor's refresh preparation (InteractiveScene.update(dt,
        # prepare_scanner=True)): the clocks of the due envs are already advanced and their records wait in the
        # stream's workspace, so the next refresh -- whoever asks for it -- must be the cast alone.
        self._prepared = False
        # where every refresh also writes the height scan of the `height_scan` observation term (bind_height_scan)
        self._height_out, self._height_offset = None, 0.0
        # True: a refresh that writes a bound height scan writes nothing else per ray (ILX_RAYS_HEIGHT_ONLY, a quarter of
        # the bytes); the hit records follow when `data.ray_hits_w` is read.  For owners that read the scan every step and
        # the hit records rarely (the fused env steps).
        self.lean_hits = False
        self._hits_behind = False          # a lean refresh has run since the hit records were last formed
        self._data._bring_up_to_date = self._form_hits
        # a new pattern array may sit where a released one sat: the mesh's streams pack their pattern again; and the
        # per-stream workspaces of this sensor's refreshes are set aside now, so that no refresh allocates (`streams`
        # of them: the stream the owner steps on, one it may fork onto, and two capture streams)
        if self._device.type == "cuda":
            _abi.check(_abi.mesh_forget_patterns(mesh.handle), "ilx_mesh_forget_patterns")
            _abi.check(_abi.raycaster_reserve(mesh.handle, n, self.num_rays, 4), "ilx_raycaster_reserve")

    @property
    def num_instances(self) -> int:
        return self._num_envs

    @property
    def device(self):
        return self._device

    @property
    def data(self) -> RayCasterData:
        self._update_outdated_buffers()
        return self._data

    # -- physics hand-over
    def set_pose(self, pos_w: torch.Tensor, quat_w: torch.Tensor):
        """Sensor parent pose in the world frame, quaternion (w, x, y, z).  Column slices of a wider state tensor
        (``root_state_w[:, :3]``, ``root_state_w[:, 3:7]``) are taken as they are: the kernel reads strided rows."""
        self._pose_pos = pos_w if (pos_w.dtype == torch.float32 and pos_w.stride(-1) == 1) else _abi.f32c(pos_w)
        self._pose_quat = quat_w if (quat_w.dtype == torch.float32 and quat_w.stride(-1) == 1) else _abi.f32c(quat_w)

    def set_root_transforms(self, root_transforms_xyzw: torch.Tensor):
        """What ``ArticulationView.get_root_transforms`` returns: (N,7) with an xyzw quaternion (ray_caster.py:225-227)."""
        pos, quat = root_transforms_xyzw.split([3, 4], dim=-1)
        self.set_pose(pos.contiguous(), math_utils.convert_quat(quat, to="wxyz").contiguous())

    def bind_height_scan(self, out: torch.Tensor | None, offset: float = 0.0):
        """Every refresh from now on also writes ``pos_w.z - hit.z - offset`` per ray into ``out`` (N, R): the value of
        the ``height_scan`` observation term (observations.py:165-173)."""
        self._height_out, self._height_offset = out, float(offset)

    # -- operations
    def reset(self, env_ids: Sequence[int] | None = None):
        """sensor_base.py:182-195 + ray_caster.py:106-113 (drift resample)."""
        if env_ids is None:
            env_ids = slice(None)
        self._timestamp[env_ids] = 0.0
        self._timestamp_last_update[env_ids] = 0.0
        self._is_outdated[env_ids] = True
        lo, hi = self.cfg.drift_range
        if self._has_drift:
            self.drift[env_ids] = self.drift[env_ids].uniform_(lo, hi)

    def update(self, dt: float, force_recompute: bool = False):
        """sensor_base.py:197-205: lazy unless forced (the data is refreshed on the next ``data`` read)."""
        self._timestamp += dt
        self._is_outdated |= self._timestamp - self._timestamp_last_update + 1e-6 >= self.cfg.update_period
        if force_recompute or self.cfg.history_length > 0:
            self._update_outdated_buffers()

    def _descriptor(self, height_out: torch.Tensor | None = None, height_offset: float = 0.0, hints: int = 0):
        """The refresh's arguments at the C ABI (``ilx_raycaster``) for the pose set last."""
        d = self._data
        mesh = self.meshes[self.cfg.mesh_prim_paths[0]]
        pp, ps = _abi.row_view(self._pose_pos, 3)
        qp, qs = _abi.row_view(self._pose_quat, 4)
        return _abi.RayCasterDesc(
            mesh=mesh.handle, num_envs=self._num_envs, num_rays=self.num_rays, attach_yaw_only=int(self.cfg.attach_yaw_only),
            hints=self._hints | hints, max_distance=float(self.cfg.max_distance), pose_pos=pp, pose_quat=qp, pose_pos_stride=ps,
            pose_quat_stride=qs, drift=_abi.ptr(self.drift),
            ray_starts_local=_abi.ptr(self.ray_starts), ray_directions_local=_abi.ptr(self.ray_directions),
            pos_w=_abi.ptr(d.pos_w), quat_w=_abi.ptr(d.quat_w), ray_hits_w=_abi.ptr(d.hits_storage),
            height_scan=_abi.ptr(height_out), height_offset=float(height_offset), timestamp=_abi.ptr(self._timestamp),
            timestamp_last_update=_abi.ptr(self._timestamp_last_update), is_outdated=self._is_outdated.data_ptr())

    def _update_outdated_buffers(self, height_out: torch.Tensor | None = None, height_offset: float = 0.0, prepared: bool = False):
        """sensor_base.py:284-293 + ray_caster.py:220-260: the flagged envs take the new pose (+ drift), cast their
        rays and record the refresh time -- two launches, the env subset is the ``_is_outdated`` mask on the device.
        ``prepared``: the scene update just before issued the refresh's preparation with its own launch
        (``InteractiveScene.update(dt, prepare_scanner=True)``, same stream, same pose); only the cast follows.  The
        scene records that on the sensor (``_prepared``), so the argument is only needed by callers that issue the
        preparation themselves."""
        if height_out is None:
            height_out, height_offset = self._height_out, self._height_offset
        # the preparation is consumed by the first refresh after it, whether the step asks for it or a reader of `data`
        prepared, self._prepared = (prepared or self._prepared), False
        hints = _abi.RAYS_PREPARED if (prepared and not self.static_workspace) else 0
        if self.lean_hits and height_out is not None and (self._hints & _abi.RAYS_Z_PARALLEL_CHECKED):
            # the library takes the hint only where its cast is the refresh's last launch; elsewhere it writes the records
            # as always and forming them again later changes nothing
            hints |= _abi.RAYS_HEIGHT_ONLY
            self._hits_behind = True
        rc = self._descriptor(height_out, height_offset, hints)
        if self.static_workspace:
            _abi.check(_abi.raycaster_refresh_static(ctypes.byref(rc), _abi.stream()), "ilx_raycaster_refresh_static")
        else:
            _abi.check(_abi.raycaster_refresh(ctypes.byref(rc), _abi.stream()), "ilx_raycaster_refresh")

    def _form_hits(self):
        """The hit records of every env's last refresh, from the pose that refresh recorded (``pos_w`` holds pose + drift,
        ``quat_w`` the sensor orientation: ray_caster.py:233-260): the cast of ``ilx_raycaster_update`` on them is the cast
        the refresh ran.  Every env has been refreshed by the time ``data`` hands the container out."""
        if not self._hits_behind:
            return
        d = self._data
        mesh = self.meshes[self.cfg.mesh_prim_paths[0]]
        hints = self._hints & (_abi.RAYS_Z_PARALLEL | _abi.RAYS_Z_PARALLEL_CHECKED)
        _abi.check(_abi.raycaster_update(mesh.handle, _abi.ptr(d.pos_w), _abi.ptr(d.quat_w), _abi.ptr(self.ray_starts),
                                         _abi.ptr(self.ray_directions), None, int(self.cfg.attach_yaw_only), float(self.cfg.max_distance),
                                         _abi.ptr(d.hits_storage), None, 0.0, self._num_envs, self.num_rays, hints, _abi.stream()),
                   "ilx_raycaster_update")
        self._hits_behind = False
