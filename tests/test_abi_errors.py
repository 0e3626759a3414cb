"""C ABI error behaviour: every entry point validates its arguments on the host before touching the GPU, returns a
negative code and leaves a message that names the function (no GPU needed: nothing here gets as far as a launch)."""

import ctypes

import numpy as np
import pytest

from isaaclab_amd import _abi

EINVAL, ESHAPE = -1, -2
NULL = None
FAKE = 0x1000  # a non-NULL value for pointers that are never dereferenced on the paths below


def fails(status, code, needle):
    assert status == code, f"status {status}, expected {code}: {_abi.lib.ilx_last_error().decode()}"
    msg = _abi.lib.ilx_last_error().decode()
    assert needle in msg, msg


def test_version_and_check_helper():
    assert _abi.lib.ilx_version() >= 1
    with pytest.raises(_abi.IlxError, match="ilx_math_op"):
        _abi.check(_abi.math_op(9999, FAKE, NULL, NULL, NULL, FAKE, NULL, 4, NULL), "ilx_math_op")


def test_math_and_articulation_arguments():
    fails(_abi.math_op(9999, FAKE, NULL, NULL, NULL, FAKE, NULL, 4, NULL), EINVAL, "ilx_math_op")
    fails(_abi.math_op(0, NULL, NULL, NULL, NULL, NULL, NULL, 4, NULL), EINVAL, "ilx_math_op")
    fails(_abi.joint_acc(NULL, FAKE, FAKE, 0.005, 8, NULL), EINVAL, "ilx_joint_acc")
    fails(_abi.root_frame_derive(NULL, 0.0, 0.0, -1.0, FAKE, FAKE, FAKE, FAKE, 4, NULL), EINVAL, "ilx_root_frame_derive")
    fails(_abi.joint_action_process(NULL, FAKE, FAKE, NULL, 1.0, NULL, 0.0, NULL, FAKE, 4, 2, NULL), EINVAL,
          "ilx_joint_action_process")
    fails(_abi.joint_action_process_begin(NULL, FAKE, FAKE, NULL, 1.0, NULL, 0.0, NULL, FAKE, 4, 2, NULL, 0, NULL, NULL), EINVAL,
          "ilx_joint_action_process_begin")
    fails(_abi.joint_action_process_begin(FAKE, FAKE, FAKE, NULL, 1.0, NULL, 0.0, NULL, FAKE, 4, 2, NULL, 3, NULL, NULL), EINVAL,
          "NULL pointer")
    fails(_abi.joint_action_process_begin(FAKE, FAKE, FAKE, NULL, 1.0, NULL, 0.0, NULL, FAKE, 0, 2, NULL, 0, NULL, NULL), EINVAL,
          "bad shape")
    # the PD model with the joint action on its launch: an effort target without tau_ff; joints 2..3 of 3; no envs
    pdja = lambda target, first, width, tau, n: _abi.actuator_pd_joint_action(
        0, FAKE, FAKE, FAKE, 1.0, NULL, target, first, width, FAKE, NULL, tau, FAKE, FAKE, FAKE, FAKE, FAKE, NULL, 1, 0.0, FAKE, FAKE,
        n, 3, NULL, 0, NULL, NULL)
    fails(pdja(1, 0, 2, NULL, 4), EINVAL, "ilx_actuator_pd_joint_action")
    fails(pdja(0, 2, 2, NULL, 4), ESHAPE, "joints 2 .. 3 of 3")
    fails(pdja(0, 0, 2, NULL, 0), EINVAL, "bad shape")


def test_actuator_arguments():
    fails(_abi.actuator_pd(7, FAKE, NULL, NULL, FAKE, FAKE, FAKE, FAKE, FAKE, NULL, 0, 0.0, FAKE, FAKE, 4, 2, NULL), EINVAL,
          "unknown model")
    # the DC motor needs velocity limits
    fails(_abi.actuator_pd(1, FAKE, NULL, NULL, FAKE, FAKE, FAKE, FAKE, FAKE, NULL, 0, 0.0, FAKE, FAKE, 4, 2, NULL), EINVAL,
          "velocity_limit")
    # delayed PD: the write pointer must lie inside the ring; a set-point and its ring come together
    fails(_abi.actuator_delayed_pd(FAKE, NULL, NULL, FAKE, FAKE, FAKE, FAKE, FAKE, 1, FAKE, NULL, NULL, 3, 3, FAKE, FAKE, NULL, NULL, 0,
                                   FAKE, FAKE, 4, 2, NULL), EINVAL, "pointer 3 outside a ring of 3")
    fails(_abi.actuator_delayed_pd(FAKE, FAKE, NULL, FAKE, FAKE, FAKE, FAKE, FAKE, 1, FAKE, NULL, NULL, 3, 0, FAKE, FAKE, NULL, NULL, 0,
                                   FAKE, FAKE, 4, 2, NULL), EINVAL, "together")
    fails(_abi.actuator_delayed_pd(FAKE, NULL, NULL, FAKE, FAKE, FAKE, FAKE, FAKE, 1, FAKE, NULL, NULL, 3, 0, FAKE, FAKE, NULL, NULL, 0,
                                   FAKE, FAKE, 4, 300, NULL), ESHAPE, "300 joints")
    w = [np.zeros((65, 6), np.float32), np.zeros((1, 65), np.float32)]
    b = [np.zeros(65, np.float32), np.zeros(1, np.float32)]
    arr = lambda xs: (ctypes.c_void_p * len(xs))(*[x.ctypes.data for x in xs])
    out = ctypes.c_void_p()
    fails(_abi.mlp_create(2, (ctypes.c_int32 * 3)(6, 65, 1), 0, arr(w), arr(b), ctypes.byref(out)), ESHAPE, "layer width 65")
    fails(_abi.mlp_create(2, (ctypes.c_int32 * 3)(6, 32, 2), 0, arr(w), arr(b), ctypes.byref(out)), ESHAPE, "output width")
    fails(_abi.mlp_create(1, (ctypes.c_int32 * 2)(6, 1), 99, arr(w), arr(b), ctypes.byref(out)), EINVAL, "activation")
    fails(_abi.actuator_net_mlp(NULL, FAKE, FAKE, FAKE, FAKE, FAKE, 3, (ctypes.c_int32 * 3)(0, 1, 2), 3, 0, 1.0, 1.0, 1.0, FAKE, FAKE,
                                1, 1.0, FAKE, FAKE, 4, 2, 0, NULL), EINVAL, "ilx_actuator_net_mlp")
    z = [np.zeros(4, np.float32)] * 4
    fails(_abi.lstm_create(2, 2, 17, arr(z), arr(z), arr(z), arr(z), 1, (ctypes.c_int32 * 2)(17, 1), 0, arr(z), arr(z),
                           ctypes.byref(out)), ESHAPE, "hidden size 17")
    fails(_abi.lstm_create(2, 2, 8, arr(z), arr(z), arr(z), arr(z), 1, (ctypes.c_int32 * 2)(6, 1), 0, arr(z), arr(z),
                           ctypes.byref(out)), ESHAPE, "head input width")
    fails(_abi.lstm_create(5, 2, 8, arr(z), arr(z), arr(z), arr(z), 1, (ctypes.c_int32 * 2)(8, 1), 0, arr(z), arr(z),
                           ctypes.byref(out)), ESHAPE, "5 layers")
    fails(_abi.actuator_net_lstm(NULL, FAKE, FAKE, FAKE, FAKE, FAKE, FAKE, FAKE, 1, 1.0, FAKE, FAKE, 4, 2, NULL), EINVAL,
          "ilx_actuator_net_lstm")


def test_sensor_arguments():
    fails(_abi.contact_update(NULL, FAKE, NULL, NULL, NULL, NULL, NULL, NULL, FAKE, FAKE, FAKE, 1, 0.005, 0.0, 1.0, 4, 2, 0, NULL), EINVAL,
          "ilx_contact_update")
    fails(_abi.contact_update(FAKE, NULL, NULL, NULL, NULL, NULL, NULL, NULL, FAKE, FAKE, FAKE, 1, 0.005, 0.0, 1.0, 4, 2, 0, NULL), EINVAL,
          "only with a history")
    fails(_abi.contact_update(FAKE, FAKE, NULL, FAKE, NULL, NULL, NULL, NULL, FAKE, FAKE, FAKE, 1, 0.005, 0.0, 1.0, 4, 2, 0, NULL), EINVAL,
          "history_head without a history")
    sc = _abi.Scene()
    sc.num_envs = -1
    fails(_abi.scene_update(ctypes.byref(sc), NULL), EINVAL, "ilx_scene_update")
    verts = np.zeros((3, 3), np.float32)
    tris = np.array([[0, 1, 5]], np.int32)  # vertex 5 does not exist
    out = ctypes.c_void_p()
    fails(_abi.mesh_create(verts.ctypes.data, 3, tris.ctypes.data, 1, ctypes.byref(out)), EINVAL, "outside")
    fails(_abi.mesh_create(verts.ctypes.data, 3, tris.ctypes.data, 0, ctypes.byref(out)), EINVAL, "empty mesh")
    bad = verts.copy()
    bad[1, 2] = np.nan
    ok_tris = np.array([[0, 1, 2]], np.int32)
    fails(_abi.mesh_create(bad.ctypes.data, 3, ok_tris.ctypes.data, 1, ctypes.byref(out)), EINVAL, "non-finite")
    fails(_abi.raycast_mesh(NULL, FAKE, FAKE, 4, 1e6, FAKE, NULL, NULL, NULL, 0, NULL), EINVAL, "mesh is NULL")
    fails(_abi.raycaster_update(NULL, FAKE, FAKE, FAKE, FAKE, NULL, 1, 1e6, FAKE, NULL, 0.0, 4, 8, 0, NULL), EINVAL, "mesh is NULL")
    rc = _abi.RayCasterDesc()
    fails(_abi.raycaster_refresh(ctypes.byref(rc), NULL), EINVAL, "ilx_raycaster_refresh")
    fails(_abi.mesh_grid_stats(NULL, NULL), EINVAL, "ilx_mesh_grid_stats")


def test_event_arguments():
    r = (ctypes.c_float * 12)()
    fails(_abi.event_reset_root_state_uniform(NULL, 4, FAKE, NULL, r, r, NULL, 0, 0, FAKE, NULL), EINVAL,
          "ilx_event_reset_root_state_uniform")
    fails(_abi.event_reset_joints(1, FAKE, 4, FAKE, FAKE, FAKE, FAKE, 0.0, 1.0, 0.0, 1.0, NULL, 0, 0, FAKE, FAKE, 0, NULL), EINVAL,
          "ilx_event_reset_joints")
    fails(_abi.event_push_by_setting_velocity(FAKE, -1, r, NULL, 0, 0, FAKE, NULL), EINVAL, "ilx_event_push_by_setting_velocity")
    # an empty env list is a no-op, as the reference's functions are for an empty env_ids tensor
    assert _abi.event_push_by_setting_velocity(NULL, 0, r, NULL, 0, 0, NULL, NULL) == 0


def test_ik_arguments():
    f = _abi.IkFrame()
    f.ee_pos = f.ee_quat = FAKE
    f.ee_pos_stride, f.ee_quat_stride = 13, 13
    fails(_abi.diff_ik_compute(ctypes.byref(f), 1, 9, 1.0, 0.01, 1e-5, FAKE, 42, 7, 7, FAKE, 7, NULL, FAKE, FAKE, FAKE, 7, 4, 7, NULL),
          EINVAL, "unknown method")
    # a Jacobian row shorter than the joint count
    fails(_abi.diff_ik_compute(ctypes.byref(f), 1, 3, 1.0, 0.01, 1e-5, FAKE, 42, 5, 7, FAKE, 7, NULL, FAKE, FAKE, FAKE, 7, 4, 7, NULL),
          EINVAL, "ilx_diff_ik_compute")
    fails(_abi.diff_ik_set_command(ctypes.byref(f), 5, 0, NULL, NULL, NULL, NULL, NULL, FAKE, FAKE, FAKE, 4, NULL), EINVAL,
          "ilx_diff_ik_set_command")


def test_mdp_plan_arguments():
    from isaaclab_amd.envs import fused_mdp as FM

    out = ctypes.c_void_p()
    fails(_abi.lib.ilx_mdp_plan_create(None, ctypes.byref(out)), EINVAL, "ilx_mdp_plan_create")
    d = FM.MdpDesc()
    d.num_envs, d.num_ids = 8, 5000   # more index-list entries than a plan can hold
    d.ids = (ctypes.c_int32 * 1)(0)
    fails(_abi.lib.ilx_mdp_plan_create(ctypes.byref(d), ctypes.byref(out)), ESHAPE, "index-list entries")
    fails(_abi.lib.ilx_mdp_step(None, ctypes.byref(d), None), EINVAL, "ilx_mdp_step")
    # an unknown term name never reaches the library
    with pytest.raises(KeyError):
        FM.FusedMdp(8, "cpu", num_joints=2, num_bodies=1, num_sensor_bodies=1, history_len=1, num_rays=0, action_dim=1,
                    command_dim=0, step_dt=0.02, max_episode_length=10, rewards=[FM.TermSpec("no_such_term", weight=1.0)],
                    terminations=[], observations=[])


def test_cpu_tensors_are_refused():
    import torch

    with pytest.raises(_abi.IlxError, match="no CPU fallback"):
        _abi.ptr(torch.zeros(4))
    with pytest.raises(_abi.IlxError, match="no CPU fallback"):
        _abi.row_view(torch.zeros(4, 3), 3)


def test_curriculum_arguments():
    f = _abi.curriculum_terrain_levels
    fails(f(FAKE, 4, FAKE, FAKE, 3, 4.0, 20.0, NULL, NULL, FAKE, FAKE, FAKE, 0, 6, NULL, 0, 1, FAKE, NULL), EINVAL, "bad shape")
    fails(f(FAKE, 4, FAKE, FAKE, 3, 4.0, 20.0, FAKE, NULL, FAKE, FAKE, FAKE, 5, 6, NULL, 0, 1, FAKE, NULL), EINVAL, "come together")
    fails(f(FAKE, 4, NULL, NULL, 0, 4.0, 20.0, NULL, NULL, FAKE, FAKE, FAKE, 5, 6, NULL, 0, 1, FAKE, NULL), EINVAL, "without move flags")
    fails(f(FAKE, 4, FAKE, FAKE, 1, 4.0, 20.0, NULL, NULL, FAKE, FAKE, FAKE, 5, 6, NULL, 0, 1, FAKE, NULL), EINVAL, "width >= 2")
    assert f(NULL, 0, NULL, NULL, 0, 4.0, 20.0, NULL, NULL, NULL, NULL, NULL, 5, 6, NULL, 0, 1, NULL, NULL) == 0  # nothing listed


def test_camera_arguments():
    d = _abi.RayCasterCameraDesc(mesh=FAKE, num_envs=4, num_rays=0)
    fails(_abi.raycaster_camera_refresh(ctypes.byref(d), NULL), EINVAL, "bad shape")
    d = _abi.RayCasterCameraDesc(mesh=FAKE, num_envs=4, num_rays=8, depth_clipping=7)
    fails(_abi.raycaster_camera_refresh(ctypes.byref(d), NULL), EINVAL, "depth clipping")
    d = _abi.RayCasterCameraDesc(mesh=FAKE, num_envs=4, num_rays=8)
    fails(_abi.raycaster_camera_refresh(ctypes.byref(d), NULL), EINVAL, "NULL operand")
    d = _abi.RayCasterCameraDesc(mesh=NULL, num_envs=4, num_rays=8)
    fails(_abi.raycaster_camera_refresh(ctypes.byref(d), NULL), EINVAL, "mesh is NULL")


def test_action_term_arguments():
    fails(_abi.joint_action_to_limits(FAKE, NULL, 1.0, NULL, 1, NULL, NULL, 8, NULL, 1.0, NULL, FAKE, 4, 3, NULL), EINVAL,
          "soft_joint_pos_limits is needed")
    fails(_abi.joint_action_to_limits(FAKE, NULL, 1.0, NULL, 0, FAKE, NULL, 8, NULL, 1.5, FAKE, FAKE, 4, 3, NULL), EINVAL,
          "Moving average weight")
    fails(_abi.joint_action_to_limits(FAKE, NULL, 1.0, NULL, 0, FAKE, NULL, 2, NULL, 1.0, NULL, FAKE, 4, 3, NULL), EINVAL, "bad shape")
    fails(_abi.binary_joint_action(FAKE, FAKE, FAKE, FAKE, NULL, FAKE, 4, 2, NULL), EINVAL, "exactly one")
    fails(_abi.non_holonomic_action_apply(FAKE, 3, FAKE, FAKE, 4, NULL), EINVAL, "quaternion stride")


def test_pose2d_command_arguments():
    from isaaclab_amd.envs.mdp.commands.pose_2d_command import Pose2dDesc, _compute, _reset

    d = Pose2dDesc(num_envs=4)
    fails(_compute(ctypes.byref(d), 0.02, NULL), EINVAL, "NULL operand")
    full = {name: FAKE for name in ("root_state_w", "default_root_state", "pos_command_w", "heading_command_w", "pos_command_b",
                                    "heading_command_b", "time_left", "command_counter", "error_pos", "error_heading", "error_pos_2d")}
    d = Pose2dDesc(num_envs=4, **full)
    fails(_compute(ctypes.byref(d), 0.02, NULL), EINVAL, "env_origins is NULL")
    d = Pose2dDesc(num_envs=4, valid_targets=FAKE, **full)
    fails(_compute(ctypes.byref(d), 0.02, NULL), EINVAL, "terrain-based targets need")
    d = Pose2dDesc(num_envs=4, env_origins=FAKE, **full)
    fails(_reset(ctypes.byref(d), NULL, NULL, NULL), EINVAL, "reset_mask is NULL")


def test_normal_velocity_command_arguments():
    from isaaclab_amd.envs.mdp.commands.velocity_command import NormalVelDesc, _normal_compute, _normal_reset

    fails(_normal_compute(ctypes.byref(NormalVelDesc(num_envs=4)), 0.02, NULL), EINVAL, "NULL operand")
    full = {name: FAKE for name in ("root_state_w", "vel_command_b", "is_zero_vel_env", "is_standing_env", "time_left", "command_counter",
                                    "error_vel_xy", "error_vel_yaw")}
    fails(_normal_compute(ctypes.byref(NormalVelDesc(num_envs=4, **full)), 0.02, NULL), EINVAL, "max_command_step must be positive")
    fails(_normal_reset(ctypes.byref(NormalVelDesc(num_envs=4, max_command_step=4.5, **full)), NULL, NULL, NULL), EINVAL, "reset_mask is NULL")
    assert _normal_compute(ctypes.byref(NormalVelDesc(num_envs=0)), 0.02, NULL) == 0


def test_modifier_arguments():
    from isaaclab_amd.utils.modifiers.modifier import ModifierDesc, _apply, _reset

    fails(_apply(9, ctypes.byref(ModifierDesc(n=4, d=3)), NULL), EINVAL, "unknown kind")
    fails(_apply(1, ctypes.byref(ModifierDesc(n=4, d=3)), NULL), EINVAL, "in or out is NULL")
    fails(_apply(1, ctypes.byref(ModifierDesc(inp=FAKE, out=FAKE, n=4, d=3, in_stride=2, out_stride=3)), NULL), ESHAPE, "row strides")
    ok = dict(inp=FAKE, out=FAKE, n=4, d=3, in_stride=3, out_stride=3)
    fails(_apply(4, ctypes.byref(ModifierDesc(na=1, nb=17, **ok)), NULL), ESHAPE, "filter lengths")
    fails(_apply(4, ctypes.byref(ModifierDesc(na=1, nb=2, **ok)), NULL), EINVAL, "histories")
    fails(_apply(5, ctypes.byref(ModifierDesc(**ok)), NULL), EINVAL, "integral and previous-value")
    assert _apply(1, ctypes.byref(ModifierDesc(n=0, d=3)), NULL) == 0
    fails(_reset(NULL, 3, NULL, 0, NULL, 4, NULL), EINVAL, "NULL state buffer")
    assert _reset(NULL, 0, NULL, 0, NULL, 4, NULL) == 0


def test_imu_arguments():
    from isaaclab_amd.sensors.imu import ImuDesc, _update

    fails(_update(ctypes.byref(ImuDesc(num_sensors=4)), 1, 0.02, 0.02, NULL), EINVAL, "physics views' buffers")
    views = dict(transforms=FAKE, velocities=FAKE, coms=FAKE)
    fails(_update(ctypes.byref(ImuDesc(num_sensors=4, **views)), 1, 0.02, 0.02, NULL), EINVAL, "NULL data buffer")
    data = {k: FAKE for k in ("pos_w", "quat_w", "lin_vel_b", "ang_vel_b", "lin_acc_b", "ang_acc_b", "prev_lin_vel_w", "prev_ang_vel_w")}
    fails(_update(ctypes.byref(ImuDesc(num_sensors=4, **views, **data)), 1, 0.02, 0.02, NULL), EINVAL, "NULL clock buffer")
    clocks = dict(timestamp=FAKE, timestamp_last_update=FAKE, is_outdated=FAKE)
    fails(_update(ctypes.byref(ImuDesc(num_sensors=4, **views, **data, **clocks)), 1, 0.02, 0.0, NULL), EINVAL, "must be positive")
    assert _update(ctypes.byref(ImuDesc(num_sensors=0)), 1, 0.02, 0.02, NULL) == 0


def test_noise_arguments():
    from isaaclab_amd.utils.noise.noise_model import NoiseDesc, _apply

    fails(_apply(3, 0, ctypes.byref(NoiseDesc(n=4, d=3)), NULL), EINVAL, "unknown kind")
    fails(_apply(1, 3, ctypes.byref(NoiseDesc(n=4, d=3)), NULL), EINVAL, "Unknown operation in noise")
    fails(_apply(1, 0, ctypes.byref(NoiseDesc(n=4, d=3)), NULL), EINVAL, "data or out is NULL")
    fails(_apply(1, 0, ctypes.byref(NoiseDesc(data=FAKE, out=FAKE, n=4, d=3, data_stride=3, out_stride=2)), NULL), ESHAPE, "row strides")
    assert _apply(1, 0, ctypes.byref(NoiseDesc(n=0, d=3)), NULL) == 0


def test_circular_buffer_arguments():
    from isaaclab_amd.utils.buffers.circular_buffer import RingDesc, _append, _reset

    fails(_append(ctypes.byref(RingDesc(n=4, d=3, max_len=0, num_pushes=FAKE)), NULL, 0, NULL, 0, NULL, NULL, 0, NULL), EINVAL, "greater than zero")
    fails(_append(ctypes.byref(RingDesc(n=4, d=3, max_len=2)), NULL, 0, NULL, 0, NULL, NULL, 0, NULL), EINVAL, "num_pushes is NULL")
    fails(_append(ctypes.byref(RingDesc(n=4, d=3, max_len=2, num_pushes=FAKE)), NULL, 0, NULL, 0, NULL, NULL, 0, NULL), EINVAL, "buffer is NULL")
    ring = dict(n=4, d=3, max_len=2, num_pushes=FAKE, buffer=FAKE)
    fails(_append(ctypes.byref(RingDesc(pointer=2, **ring)), NULL, 0, NULL, 0, NULL, NULL, 0, NULL), EINVAL, "outside the 2 slots")
    fails(_append(ctypes.byref(RingDesc(**ring)), FAKE, 2, NULL, 0, NULL, NULL, 0, NULL), ESHAPE, "data row stride")
    fails(_append(ctypes.byref(RingDesc(**ring)), FAKE, 3, FAKE, 5, NULL, NULL, 0, NULL), ESHAPE, "ordered row stride")
    fails(_reset(ctypes.byref(RingDesc(**ring)), NULL, -1, NULL), EINVAL, "negative count")
    assert _append(ctypes.byref(RingDesc(n=0, d=3, max_len=2)), NULL, 0, NULL, 0, NULL, NULL, 0, NULL) == 0


def test_pose_and_camera_op_arguments():
    from isaaclab_amd.utils.pose_math import _camera_op, _cylinder, _pose_op

    fails(_pose_op(7, FAKE, 4, NULL, 1, 0.0, FAKE, NULL, 4, NULL), EINVAL, "unknown op")
    fails(_pose_op(0, FAKE, 4, NULL, 4, 0.0, FAKE, NULL, 4, NULL), EINVAL, "NULL operand")
    fails(_pose_op(1, FAKE, 4, NULL, 1, 0.0, NULL, NULL, 4, NULL), EINVAL, "NULL output")
    fails(_pose_op(2, FAKE, 3, FAKE, 4, 0.0, FAKE, NULL, 4, NULL), ESHAPE, "neither 1 nor 4")
    fails(_pose_op(4, FAKE, 4, FAKE, 4, 0.0, NULL, NULL, 4, NULL), EINVAL, "NULL output")
    assert _pose_op(1, NULL, 0, NULL, 0, 0.0, NULL, NULL, 0, NULL) == 0
    fails(_camera_op(3, FAKE, FAKE, 1, FAKE, 2, 4, 4, NULL), EINVAL, "unknown op")
    fails(_camera_op(0, FAKE, NULL, 1, FAKE, 2, 4, 4, NULL), EINVAL, "NULL operand")
    fails(_camera_op(1, FAKE, FAKE, 3, FAKE, 2, 4, 4, NULL), ESHAPE, "calibration matrices")
    fails(_cylinder(1.0, 0.0, 1.0, NULL, 0, 0, NULL, 4, NULL), EINVAL, "out is NULL")


def test_frame_transformer_arguments():
    from isaaclab_amd.sensors.frame_transformer import FrameTransformerDesc, _update

    fails(_update(ctypes.byref(FrameTransformerDesc(num_envs=4, num_frames=2)), NULL), EINVAL, "row indices are NULL")
    rows = dict(transforms=FAKE, source_rows=FAKE, target_rows=FAKE)
    fails(_update(ctypes.byref(FrameTransformerDesc(num_envs=4, num_frames=2, source_offset_pos=FAKE, **rows)), NULL), EINVAL,
          "both its position and its orientation")
    fails(_update(ctypes.byref(FrameTransformerDesc(num_envs=4, num_frames=2, **rows)), NULL), EINVAL, "NULL data buffer")
    assert _update(ctypes.byref(FrameTransformerDesc(num_envs=4, num_frames=0)), NULL) == 0


def test_reset_from_terrain_arguments():
    r12 = (ctypes.c_float * 12)()
    fails(_abi.event_reset_root_state_from_terrain(FAKE, 4, FAKE, NULL, FAKE, FAKE, 3, 5, NULL, r12, r12, NULL, 0, 0, FAKE, NULL), EINVAL,
          "flat patches need their table")
    fails(_abi.event_reset_root_state_from_terrain(FAKE, 4, NULL, FAKE, FAKE, FAKE, 3, 5, NULL, r12, r12, NULL, 0, 0, FAKE, NULL), EINVAL,
          "NULL operand")
    assert _abi.event_reset_root_state_from_terrain(NULL, 0, NULL, NULL, NULL, NULL, 0, 0, NULL, NULL, NULL, NULL, 0, 0, NULL, NULL) == 0


def test_flat_patch_arguments():
    from isaaclab_amd.terrains.utils import _check, _rays

    fails(_rays(FAKE, FAKE, FAKE, FAKE, 4, 0, FAKE, FAKE, FAKE, NULL), EINVAL, "bad shape")
    fails(_rays(FAKE, NULL, FAKE, FAKE, 4, 10, FAKE, FAKE, FAKE, NULL), EINVAL, "NULL operand")
    fails(_check(NULL, FAKE, 4, 10, 0.0, 1.0, 0.1, FAKE, FAKE, NULL), EINVAL, "NULL operand")
    assert _check(NULL, NULL, 0, 10, 0.0, 1.0, 0.1, NULL, NULL, NULL) == 0


def test_osc_action_state_arguments():
    from isaaclab_amd.envs.mdp.actions.task_space_actions import OscActionStateDesc, _osc_state

    fails(_osc_state(ctypes.byref(OscActionStateDesc(num_envs=4)), NULL), EINVAL, "root / body state or ee_pose_b is NULL")
    pose = dict(num_envs=4, root_state_w=FAKE, body_state_w=FAKE, ee_pose_b=FAKE, num_bodies=5)
    fails(_osc_state(ctypes.byref(OscActionStateDesc(ee_body_idx=5, **pose)), NULL), EINVAL, "outside the 5 bodies")
    fails(_osc_state(ctypes.byref(OscActionStateDesc(ee_body_idx=4, num_dof=13, **pose)), NULL), ESHAPE, "controlled joints outside")
    fails(_osc_state(ctypes.byref(OscActionStateDesc(ee_body_idx=4, num_dof=7, **pose)), NULL), EINVAL, "index lists are NULL")
    assert _osc_state(ctypes.byref(OscActionStateDesc(num_envs=0)), NULL) == 0


def test_sampler_arguments():
    fails(_abi.sample(9, NULL, 0, 0, 0.0, NULL, 0, 0, 1.0, NULL, NULL, 0, 1, FAKE, 4, 3, NULL), EINVAL, "unknown kind")
    fails(_abi.sample(4, NULL, 0, 0, 0.0, NULL, 0, 0, 1.0, NULL, NULL, 0, 1, FAKE, 4, 3, NULL), ESHAPE, "4 wide")
    fails(_abi.sample(0, NULL, 0, 0, 0.0, NULL, 0, 0, 1.0, NULL, NULL, 0, 1, NULL, 4, 3, NULL), EINVAL, "out is NULL")
