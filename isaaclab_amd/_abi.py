"""ctypes binding of ``libisaaclab_amd.so`` (the C ABI declared in ``include/isaaclab_amd.h``).

There is no fallback: if the shared library is missing, importing this module raises, and every wrapper
refuses tensors that do not live on a GPU.  ``torch`` is imported first so that the HIP runtime the
library binds to is the one PyTorch already loaded (same process-wide streams and allocations).
"""

from __future__ import annotations

import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libisaaclab_amd.so")


class IlxError(RuntimeError):
    """Raised when a C-ABI call returns a negative status."""


def _load() -> ctypes.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP kernels have not been built. Run `make -C isaaclab_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback."
        )
    return ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)


lib = _load()
lib.ilx_last_error.restype = ctypes.c_char_p
lib.ilx_version.restype = ctypes.c_int

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_int64 = ctypes.c_int64
c_float = ctypes.c_float


def _sig(name, *argtypes):
    fn = getattr(lib, name)
    fn.argtypes = list(argtypes)
    fn.restype = ctypes.c_int
    return fn


def check(status: int, what: str = ""):
    if status != 0:
        msg = lib.ilx_last_error()
        raise IlxError(f"{what} failed ({status}): {msg.decode() if msg else ''}")


def ptr(t: torch.Tensor | None) -> int | None:
    """Device pointer of a dense GPU tensor (None passes NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise IlxError("isaaclab_amd kernels need GPU tensors (got a CPU tensor); there is no CPU fallback")
    if not t.is_contiguous():
        raise IlxError("isaaclab_amd kernels need contiguous tensors")
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream() -> int:
    """The current torch stream (of the current device) as a raw hipStream_t."""
    if _raw_stream is not None:   # the same handle without building a torch.cuda.Stream object per launch (~3 us each)
        return _raw_stream(-1)
    return torch.cuda.current_stream().cuda_stream


def f32c(t: torch.Tensor) -> torch.Tensor:
    """Dense float32 view/copy of ``t``."""
    if t.dtype != torch.float32:
        t = t.to(torch.float32)
    return t if t.is_contiguous() else t.contiguous()


# ---- signatures --------------------------------------------------------------------------------
P = c_void_p
math_op = _sig("ilx_math_op", c_int, P, P, P, P, P, P, c_int64, P)
sample = _sig("ilx_sample", c_int, P, c_int64, c_int64, c_float, P, c_int64, c_int64, c_float, P, P, ctypes.c_uint64, ctypes.c_uint64, P,
              c_int64, c_int, P)
math_range_op = _sig("ilx_math_range_op", c_int, P, P, P, c_int, P, c_int64, c_int, P)
root_state_pack = _sig("ilx_root_state_pack", P, P, P, c_int64, P)
root_frame_derive = _sig("ilx_root_frame_derive", P, c_float, c_float, c_float, P, P, P, P, c_int64, P)
joint_acc = _sig("ilx_joint_acc", P, P, P, c_float, c_int64, P)
actuator_pd = _sig("ilx_actuator_pd", c_int, P, P, P, P, P, P, P, P, P, c_int, c_float, P, P, c_int64, c_int, P)
actuator_delayed_pd = _sig("ilx_actuator_delayed_pd", P, P, P, P, P, P, P, P, c_int, P, P, P, c_int, c_int, P, P, P, P, c_int, P, P,
                           c_int64, c_int, P)
lstm_create = _sig("ilx_lstm_create", c_int, c_int, c_int, P, P, P, P, c_int, P, c_int, P, P, P)
lib.ilx_lstm_destroy.argtypes = [P]
lib.ilx_lstm_destroy.restype = None
actuator_net_lstm = _sig("ilx_actuator_net_lstm", P, P, P, P, P, P, P, P, c_int, c_float, P, P, c_int64, c_int, P)
mlp_create = _sig("ilx_mlp_create", c_int, P, c_int, P, P, P)
lib.ilx_mlp_destroy.argtypes = [P]
lib.ilx_mlp_destroy.restype = None
actuator_net_mlp = _sig("ilx_actuator_net_mlp", P, P, P, P, P, P, c_int, P, c_int, c_int, c_float, c_float, c_float,
                        P, P, c_int, c_float, P, P, c_int64, c_int, c_int, P)
contact_update = _sig("ilx_contact_update", P, P, P, P, P, P, P, P, P, P, P, c_int, c_float, c_float, c_float, c_int64, c_int,
                      c_int, P)
mesh_create = _sig("ilx_mesh_create", P, c_int64, P, c_int64, P)
lib.ilx_mesh_destroy.argtypes = [P]
lib.ilx_mesh_destroy.restype = None
mesh_info = _sig("ilx_mesh_info", P, P, P, P)
mesh_grid_stats = _sig("ilx_mesh_grid_stats", P, P)
mesh_lattice_stats = _sig("ilx_mesh_lattice_stats", P, P)
mesh_lattice_counter = _sig("ilx_mesh_lattice_counter", P, c_int, P)
mesh_lattice_probe = _sig("ilx_mesh_lattice_probe", P, c_int64, P, c_int64, P)
MLP_FORCE_GENERIC, MLP_EXACT_F32 = 1, 2   # ilx_actuator_net_mlp flags (ILX_MLP_*)
RAYS_Z_PARALLEL = 1
RAYS_Z_PARALLEL_CHECKED = 2
RAYS_PATTERN_STATIC = 4
RAYS_PREPARED = 8
RAYS_HEIGHT_ONLY = 16
raycast_mesh = _sig("ilx_raycast_mesh", P, P, P, c_int64, c_float, P, P, P, P, c_int, P)
raycaster_update = _sig("ilx_raycaster_update", P, P, P, P, P, P, c_int, c_float, P, P, c_float, c_int64, c_int, c_int, P)
joint_action_process = _sig("ilx_joint_action_process", P, P, P, P, c_float, P, c_float, P, P, c_int64, c_int, P)
joint_action_process_begin = _sig("ilx_joint_action_process_begin", P, P, P, P, c_float, P, c_float, P, P, c_int64, c_int, P,
                                  ctypes.c_int32, P, P)
joint_action_to_limits = _sig("ilx_joint_action_to_limits", P, P, c_float, P, c_int, P, P, c_int, P, c_float, P, P, c_int64, c_int, P)
binary_joint_action = _sig("ilx_binary_joint_action", P, P, P, P, P, P, c_int64, c_int, P)
non_holonomic_action_apply = _sig("ilx_non_holonomic_action_apply", P, c_int64, P, P, c_int64, P)
raycast_mesh_stats = _sig("ilx_raycast_mesh_stats", P, P, P, c_int64, c_float, P, P, P)


c_uint64 = ctypes.c_uint64
event_reset_root_state_from_terrain = _sig("ilx_event_reset_root_state_from_terrain", P, c_int64, P, P, P, P, c_int, c_int, P, P, P, P,
                                          ctypes.c_uint64, ctypes.c_uint64, P, P)
event_reset_root_state_uniform = _sig("ilx_event_reset_root_state_uniform", P, c_int64, P, P, P, P, P, c_uint64, c_uint64, P, P)
event_reset_joints = _sig("ilx_event_reset_joints", c_int, P, c_int64, P, P, P, P, c_float, c_float, c_float, c_float, P, c_uint64,
                          c_uint64, P, P, c_int, P)
event_push_by_setting_velocity = _sig("ilx_event_push_by_setting_velocity", P, c_int64, P, P, c_uint64, c_uint64, P, P)
event_reset_root_state_random_orientation = _sig("ilx_event_reset_root_state_random_orientation", P, c_int64, P, P, P, P, P, P, c_uint64,
                                                 c_uint64, P, P)
event_apply_external_force_torque = _sig("ilx_event_apply_external_force_torque", P, c_int64, P, c_int, c_int, c_float, c_float, c_float,
                                         c_float, P, c_uint64, c_uint64, P, P, P)
event_reset_scene_to_default = _sig("ilx_event_reset_scene_to_default", P, c_int64, P, P, P, P, P, P, P, c_int, P)
curriculum_terrain_levels = _sig("ilx_curriculum_terrain_levels", P, c_int64, P, P, c_int, c_float, c_float, P, P, P, P, P, c_int, c_int,
                                 P, c_uint64, c_uint64, P, P)


class Scene(ctypes.Structure):
    """Mirror of ``ilx_scene`` (include/isaaclab_amd.h)."""

    _fields_ = [
        ("num_envs", c_int64), ("dt", c_float),
        ("joint_vel", c_void_p), ("previous_joint_vel", c_void_p), ("joint_acc", c_void_p),
        ("num_joints", ctypes.c_int32), ("joint_acc_elapsed", c_float),
        ("net_forces_in", c_void_p), ("net_forces_w", c_void_p), ("net_forces_w_history", c_void_p),
        ("current_air_time", c_void_p), ("last_air_time", c_void_p), ("current_contact_time", c_void_p),
        ("last_contact_time", c_void_p), ("contact_timestamp", c_void_p), ("contact_timestamp_last_update", c_void_p),
        ("contact_is_outdated", c_void_p), ("contact_update_period", c_float), ("force_threshold", c_float),
        ("num_bodies", ctypes.c_int32), ("history_length", ctypes.c_int32), ("contact_lazy", ctypes.c_int32),
        ("ray_timestamp", c_void_p), ("ray_timestamp_last_update", c_void_p), ("ray_is_outdated", c_void_p),
        ("ray_update_period", c_float), ("contact_history_head", c_void_p),
    ]


class RayCasterDesc(ctypes.Structure):
    """Mirror of ``ilx_raycaster`` (include/isaaclab_amd.h)."""

    _fields_ = [
        ("mesh", c_void_p), ("num_envs", c_int64), ("num_rays", ctypes.c_int32), ("attach_yaw_only", ctypes.c_int32),
        ("hints", ctypes.c_int32), ("max_distance", c_float),
        ("pose_pos", c_void_p), ("pose_quat", c_void_p), ("pose_pos_stride", c_int64), ("pose_quat_stride", c_int64),
        ("drift", c_void_p), ("ray_starts_local", c_void_p), ("ray_directions_local", c_void_p),
        ("pos_w", c_void_p), ("quat_w", c_void_p), ("ray_hits_w", c_void_p), ("height_scan", c_void_p),
        ("height_offset", c_float), ("timestamp", c_void_p), ("timestamp_last_update", c_void_p), ("is_outdated", c_void_p),
    ]


class RayCasterCameraDesc(ctypes.Structure):
    """Mirror of ``ilx_raycaster_camera`` (include/isaaclab_amd.h)."""

    _fields_ = [
        ("mesh", c_void_p), ("num_envs", c_int64), ("num_rays", ctypes.c_int32), ("depth_clipping", ctypes.c_int32),
        ("max_distance", c_float), ("view_pos", c_void_p), ("view_quat", c_void_p), ("view_pos_stride", c_int64),
        ("view_quat_stride", c_int64), ("offset_pos", c_void_p), ("offset_quat", c_void_p), ("ray_starts", c_void_p),
        ("ray_directions", c_void_p), ("pos_w", c_void_p), ("quat_w_world", c_void_p), ("frame", c_void_p),
        ("ray_hits_w", c_void_p), ("distance_to_image_plane", c_void_p), ("distance_to_camera", c_void_p), ("normals", c_void_p),
        ("timestamp", c_void_p), ("timestamp_last_update", c_void_p), ("is_outdated", c_void_p),
    ]


DEPTH_CLIP = {"none": 0, "max": 1, "zero": 2}
raycaster_camera_refresh = _sig("ilx_raycaster_camera_refresh", ctypes.POINTER(RayCasterCameraDesc), P)
scene_update = _sig("ilx_scene_update", ctypes.POINTER(Scene), P)
commands_begin = _sig("ilx_commands_begin", ctypes.POINTER(c_void_p))
commands_end = _sig("ilx_commands_end")
commands_count = _sig("ilx_commands_count", P, ctypes.POINTER(c_int))
commands_replay = _sig("ilx_commands_replay", P, P)
commands_destroy = _sig("ilx_commands_destroy", P)
raycaster_refresh = _sig("ilx_raycaster_refresh", ctypes.POINTER(RayCasterDesc), P)
raycaster_refresh_static = _sig("ilx_raycaster_refresh_static", ctypes.POINTER(RayCasterDesc), P)
raycaster_reserve = _sig("ilx_raycaster_reserve", P, c_int64, ctypes.c_int32, ctypes.c_int32)
mesh_forget_patterns = _sig("ilx_mesh_forget_patterns", P)
actuator_net_mlp_joint_action = _sig("ilx_actuator_net_mlp_joint_action", P, P, P, P, c_float, P, P, P, P, P, P, c_int, P, c_int, c_int,
                                     c_float, c_float, c_float, P, P, c_int, c_float, P, P, c_int64, c_int, c_int, P, ctypes.c_int32, P, P)
actuator_pd_joint_action = _sig("ilx_actuator_pd_joint_action", c_int, P, P, P, c_float, P, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                P, P, P, P, P, P, P, P, P, c_int, c_float, P, P, c_int64, c_int, P, ctypes.c_int32, P, P)
actuator_pd_scene_update = _sig("ilx_actuator_pd_scene_update", c_int, P, P, P, P, P, P, P, P, P, c_int, c_float, P, P, c_int64, c_int,
                                ctypes.POINTER(Scene), P)
actuator_net_mlp_scene_update = _sig("ilx_actuator_net_mlp_scene_update", P, P, P, P, P, P, c_int, P, c_int, c_int, c_float, c_float, c_float,
                                     P, P, c_int, c_float, P, P, c_int64, c_int, c_int, ctypes.POINTER(Scene), P)
scene_update_prepare_raycaster = _sig("ilx_scene_update_prepare_raycaster", ctypes.POINTER(Scene), ctypes.POINTER(RayCasterDesc), P)


def row_view(t: torch.Tensor, width: int):
    """(pointer, row stride in floats) of an (n, width) float32 GPU view whose rows are contiguous (e.g. a column
    slice of ``root_state_w``)."""
    if not t.is_cuda:
        raise IlxError("isaaclab_amd kernels need GPU tensors (got a CPU tensor); there is no CPU fallback")
    if t.dtype != torch.float32 or t.dim() != 2 or t.shape[1] != width or t.stride(1) != 1:
        raise IlxError(f"expected an (n, {width}) float32 view with contiguous rows, got {tuple(t.shape)} {t.dtype} "
                       f"strides {t.stride()}")
    return t.data_ptr(), t.stride(0)


class IkFrame(ctypes.Structure):
    """Mirror of ``ilx_ik_frame`` (include/isaaclab_amd.h)."""

    _fields_ = [
        ("ee_pos", c_void_p), ("ee_quat", c_void_p), ("ee_pos_stride", c_int64), ("ee_quat_stride", c_int64),
        ("root_pos", c_void_p), ("root_quat", c_void_p), ("root_pos_stride", c_int64), ("root_quat_stride", c_int64),
        ("has_offset", ctypes.c_int32), ("offset_pos", c_float * 3), ("offset_rot", c_float * 4),
    ]


IK_POSITION, IK_POSE = 0, 1
IK_METHODS = {"pinv": 0, "svd": 1, "trans": 2, "dls": 3}
diff_ik_set_command = _sig("ilx_diff_ik_set_command", ctypes.POINTER(IkFrame), c_int, c_int, P, P, P, P, P, P, P, P, c_int64, P)
diff_ik_compute = _sig("ilx_diff_ik_compute", ctypes.POINTER(IkFrame), c_int, c_int, c_float, c_float, c_float, P, c_int64,
                       c_int64, P, P, c_int64, P, P, P, P, c_int64, c_int64, c_int, P)


def ik_frame(ee_pos, ee_quat, root_pos=None, root_quat=None, offset_pos=None, offset_rot=None) -> IkFrame:
    """Describe the controlled frame from (possibly strided) row views."""
    f = IkFrame()
    f.ee_pos, f.ee_pos_stride = row_view(ee_pos, 3)
    f.ee_quat, f.ee_quat_stride = row_view(ee_quat, 4)
    if root_pos is not None:
        f.root_pos, f.root_pos_stride = row_view(root_pos, 3)
        f.root_quat, f.root_quat_stride = row_view(root_quat, 4)
        if offset_pos is not None:
            f.has_offset = 1
            f.offset_pos = (c_float * 3)(*[float(v) for v in offset_pos])
            f.offset_rot = (c_float * 4)(*[float(v) for v in offset_rot])
    return f
