"""RayCasterCamera on the fused HIP kernel ``ilx_raycaster_camera_refresh`` (interface of
``isaaclab/sensors/ray_caster/ray_caster_camera.py`` + ``ray_caster_camera_cfg.py`` + ``camera/camera_data.py``).

As with :class:`isaaclab_amd.sensors.ray_caster.RayCaster`, the owner supplies the pose of the prim the camera is
attached to (``set_view_pose``) and the collision mesh; camera pose, world rays, the cast, the two distance images with
their clipping rule and the normals are one launch, with no (N, H*W, 3) temporaries.
"""

from __future__ import annotations

import ctypes
from collections.abc import Sequence
from dataclasses import dataclass, field

import torch

from .. import _abi
from ..utils import camera_math, math as math_utils
from ..utils.mesh import Mesh
from .patterns import PinholeCameraPatternCfg


@dataclass
class CameraOffsetCfg:
    """ray_caster_camera_cfg.py:22-40."""

    pos: tuple[float, float, float] = (0.0, 0.0, 0.0)
    rot: tuple[float, float, float, float] = (1.0, 0.0, 0.0, 0.0)
    convention: str = "ros"


@dataclass
class RayCasterCameraCfg:
    """ray_caster_camera_cfg.py."""

    prim_path: str = ""
    update_period: float = 0.0
    history_length: int = 0
    debug_vis: bool = False
    mesh_prim_paths: list[str] = field(default_factory=lambda: ["/World/ground"])
    offset: CameraOffsetCfg = field(default_factory=CameraOffsetCfg)
    data_types: list[str] = field(default_factory=lambda: ["distance_to_image_plane"])
    depth_clipping_behavior: str = "none"
    pattern_cfg: PinholeCameraPatternCfg = field(default_factory=PinholeCameraPatternCfg)
    max_distance: float = 1e6


@dataclass
class CameraData:
    """camera/camera_data.py:17-91."""

    pos_w: torch.Tensor = None
    quat_w_world: torch.Tensor = None
    image_shape: tuple[int, int] = None
    intrinsic_matrices: torch.Tensor = None
    output: dict = None
    info: list = None

    @property
    def quat_w_ros(self) -> torch.Tensor:
        return camera_math.convert_camera_frame_orientation_convention(self.quat_w_world, origin="world", target="ros")

    @property
    def quat_w_opengl(self) -> torch.Tensor:
        return camera_math.convert_camera_frame_orientation_convention(self.quat_w_world, origin="world", target="opengl")


class RayCasterCamera:
    UNSUPPORTED_TYPES = {"rgb", "instance_id_segmentation", "instance_id_segmentation_fast", "instance_segmentation",
                         "instance_segmentation_fast", "semantic_segmentation", "skeleton_data", "motion_vectors",
                         "bounding_box_2d_tight", "bounding_box_2d_tight_fast", "bounding_box_2d_loose",
                         "bounding_box_2d_loose_fast", "bounding_box_3d", "bounding_box_3d_fast"}

    def __init__(self, cfg: RayCasterCameraCfg, num_envs: int, mesh: Mesh, device):
        common = set(cfg.data_types) & self.UNSUPPORTED_TYPES  # ray_caster_camera.py:331-342
        if common:
            raise ValueError(f"RayCasterCamera class does not support the following sensor types: {common}."
                             "\n\tThis is because these sensor types cannot be obtained in a fast way using ''warp''."
                             "\n\tHint: If you need to work with these sensor types, we recommend using the USD camera"
                             " interface from the isaaclab.sensors.camera module.")
        if len(cfg.mesh_prim_paths) != 1:
            raise NotImplementedError(f"RayCaster currently only supports one mesh prim. Received: {len(cfg.mesh_prim_paths)}")
        if cfg.depth_clipping_behavior not in _abi.DEPTH_CLIP:
            raise ValueError(f"Unknown depth_clipping_behavior '{cfg.depth_clipping_behavior}' (none, max, zero).")
        self.cfg, self._num_envs, self._device = cfg, num_envs, torch.device(device)
        self.meshes = {cfg.mesh_prim_paths[0]: mesh}
        n, dev, pc = num_envs, self._device, cfg.pattern_cfg
        self._is_outdated = torch.ones(n, dtype=torch.bool, device=dev)
        self._timestamp = torch.zeros(n, device=dev)
        self._timestamp_last_update = torch.zeros(n, device=dev)
        self._ALL_INDICES = torch.arange(n, device=dev, dtype=torch.long)
        self._frame = torch.zeros(n, device=dev, dtype=torch.long)
        # -- buffers (ray_caster_camera.py:344-368)
        self.drift = torch.zeros(n, 3, device=dev)
        self._data = CameraData(pos_w=torch.zeros(n, 3, device=dev), quat_w_world=torch.zeros(n, 4, device=dev),
                                intrinsic_matrices=torch.zeros(n, 3, 3, device=dev), image_shape=self.image_shape, output={},
                                info=[{name: None for name in cfg.data_types}] * n)
        self._data.intrinsic_matrices[:, 2, 2] = 1.0
        for name in cfg.data_types:
            if name in ("distance_to_image_plane", "distance_to_camera"):
                shape = (pc.height, pc.width, 1)
            elif name == "normals":
                shape = (pc.height, pc.width, 3)
            else:
                raise ValueError(f"Received unknown data type: {name}. Please check the configuration.")
            self._data.output[name] = torch.zeros((n, *shape), device=dev)
        # -- intrinsics (:370-392)
        if pc.vertical_aperture is None:
            pc.vertical_aperture = pc.horizontal_aperture * pc.height / pc.width
        f_x = pc.width * pc.focal_length / pc.horizontal_aperture
        f_y = pc.height * pc.focal_length / pc.vertical_aperture
        k = self._data.intrinsic_matrices
        k[:, 0, 0], k[:, 0, 2] = f_x, pc.horizontal_aperture_offset * f_x + pc.width / 2
        k[:, 1, 1], k[:, 1, 2] = f_y, pc.vertical_aperture_offset * f_y + pc.height / 2
        self._focal_length = pc.focal_length
        # -- rays and offsets (:243-255)
        self.ray_starts, self.ray_directions = pc.func(pc, k, dev)
        self.num_rays = self.ray_directions.shape[1]
        self.ray_hits_w = torch.zeros(n, self.num_rays, 3, device=dev)
        quat_w = camera_math.convert_camera_frame_orientation_convention(torch.tensor([cfg.offset.rot], device=dev),
                                                                         origin=cfg.offset.convention, target="world")
        self._offset_quat = quat_w.repeat(n, 1).contiguous()
        self._offset_pos = torch.tensor(list(cfg.offset.pos), device=dev).repeat(n, 1).contiguous()
        self._view_pos = torch.zeros(n, 3, device=dev)
        self._view_quat = torch.zeros(n, 4, device=dev)
        self._view_quat[:, 0] = 1.0

    # -- properties
    @property
    def num_instances(self) -> int:
        return self._num_envs

    @property
    def device(self):
        return self._device

    @property
    def data(self) -> CameraData:
        self._update_outdated_buffers()
        return self._data

    @property
    def image_shape(self) -> tuple[int, int]:
        return (self.cfg.pattern_cfg.height, self.cfg.pattern_cfg.width)

    @property
    def frame(self) -> torch.Tensor:
        return self._frame

    # -- physics hand-over
    def set_view_pose(self, pos_w: torch.Tensor, quat_w: torch.Tensor):
        """World pose of the prim the camera is attached to, quaternion (w, x, y, z); row-strided views are read in place."""
        self._view_pos = pos_w if (pos_w.dtype == torch.float32 and pos_w.stride(-1) == 1) else _abi.f32c(pos_w)
        self._view_quat = quat_w if (quat_w.dtype == torch.float32 and quat_w.stride(-1) == 1) else _abi.f32c(quat_w)

    # -- operations
    def set_intrinsic_matrices(self, matrices: torch.Tensor, focal_length: float = 1.0, env_ids: Sequence[int] | None = None):
        """ray_caster_camera.py:121-140."""
        if env_ids is None:
            env_ids = slice(None)
        self._data.intrinsic_matrices[env_ids] = matrices.to(self._device)
        self._focal_length = focal_length
        self.ray_starts[env_ids], self.ray_directions[env_ids] = self.cfg.pattern_cfg.func(
            self.cfg.pattern_cfg, self._data.intrinsic_matrices[env_ids], self._device)

    def reset(self, env_ids: Sequence[int] | None = None):
        """ray_caster_camera.py:142-154."""
        if env_ids is None:
            env_ids = slice(None)
        self._timestamp[env_ids] = 0.0
        self._timestamp_last_update[env_ids] = 0.0
        self._is_outdated[env_ids] = True
        pos_w, quat_w = self._compute_camera_world_poses(env_ids)
        self._data.pos_w[env_ids] = pos_w
        self._data.quat_w_world[env_ids] = quat_w
        self._frame[env_ids] = 0

    def set_world_poses(self, positions=None, orientations=None, env_ids=None, convention: str = "ros"):
        """ray_caster_camera.py:156-206: re-derives the offsets such that the camera sits at the given world pose."""
        if env_ids is None:
            env_ids = self._ALL_INDICES
        pos_w, quat_w = self._view_pos[env_ids].clone(), self._view_quat[env_ids].clone().contiguous()
        if positions is not None:
            self._offset_pos[env_ids] = math_utils.quat_apply(math_utils.quat_inv(quat_w), (positions - pos_w).contiguous())
        if orientations is not None:
            quat_w_set = camera_math.convert_camera_frame_orientation_convention(orientations, origin=convention, target="world")
            self._offset_quat[env_ids] = math_utils.quat_mul(math_utils.quat_inv(quat_w), quat_w_set.contiguous())
        pos_w, quat_w = self._compute_camera_world_poses(env_ids)
        self._data.pos_w[env_ids] = pos_w
        self._data.quat_w_world[env_ids] = quat_w

    def set_world_poses_from_view(self, eyes: torch.Tensor, targets: torch.Tensor, env_ids=None, up_axis: str = "Z"):
        """ray_caster_camera.py:208-228 (the stage's up axis is an argument here: there is no stage)."""
        orientations = camera_math.quat_from_matrix(
            camera_math.create_rotation_matrix_from_view(eyes, targets, up_axis=up_axis, device=self._device))
        self.set_world_poses(eyes, orientations, env_ids, convention="opengl")

    def update(self, dt: float, force_recompute: bool = False):
        """sensor_base.py:197-205."""
        self._timestamp += dt
        self._is_outdated |= self._timestamp - self._timestamp_last_update + 1e-6 >= self.cfg.update_period
        if force_recompute or self.cfg.history_length > 0:
            self._update_outdated_buffers()

    def _compute_camera_world_poses(self, env_ids):
        """ray_caster_camera.py:415-430."""
        pos_w, quat_w = self._view_pos[env_ids].clone(), self._view_quat[env_ids].clone().contiguous()
        pos_w = pos_w + math_utils.quat_apply(quat_w, self._offset_pos[env_ids].contiguous())
        return pos_w, math_utils.quat_mul(quat_w, self._offset_quat[env_ids].contiguous())

    def _update_outdated_buffers(self):
        """sensor_base.py:284-293 + ray_caster_camera.py:257-318 for the flagged cameras: two launches."""
        d, out = self._data, self._data.output
        pp, ps = _abi.row_view(self._view_pos, 3)
        qp, qs = _abi.row_view(self._view_quat, 4)
        rc = _abi.RayCasterCameraDesc(
            mesh=self.meshes[self.cfg.mesh_prim_paths[0]].handle, num_envs=self._num_envs, num_rays=self.num_rays,
            depth_clipping=_abi.DEPTH_CLIP[self.cfg.depth_clipping_behavior], max_distance=float(self.cfg.max_distance),
            view_pos=pp, view_quat=qp, view_pos_stride=ps, view_quat_stride=qs, offset_pos=_abi.ptr(self._offset_pos),
            offset_quat=_abi.ptr(self._offset_quat), ray_starts=_abi.ptr(self.ray_starts), ray_directions=_abi.ptr(self.ray_directions),
            pos_w=_abi.ptr(d.pos_w), quat_w_world=_abi.ptr(d.quat_w_world), frame=self._frame.data_ptr(),
            ray_hits_w=_abi.ptr(self.ray_hits_w), distance_to_image_plane=_abi.ptr(out.get("distance_to_image_plane")),
            distance_to_camera=_abi.ptr(out.get("distance_to_camera")), normals=_abi.ptr(out.get("normals")),
            timestamp=_abi.ptr(self._timestamp), timestamp_last_update=_abi.ptr(self._timestamp_last_update),
            is_outdated=self._is_outdated.data_ptr())
        _abi.check(_abi.raycaster_camera_refresh(ctypes.byref(rc), _abi.stream()), "ilx_raycaster_camera_refresh")
