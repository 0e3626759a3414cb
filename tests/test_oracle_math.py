"""CPU: the oracle's math restatement against fixtures produced by the reference's utils/math.py."""

import math

import torch

from conftest import assert_close, assert_exact, load_golden
from oracle import mathx as M

# functions whose reference evaluation contains no transcendental call: the oracle must be bit-identical
EXACT = True


def _g():
    return load_golden("math_fns")


def test_rotations_bit_exact():
    g = _g()
    q1, q2, qu, v = g["q1"], g["q2"], g["qu"], g["v"]
    assert_exact(M.quat_rotate(q1, v), g["quat_rotate"], "quat_rotate")
    assert_exact(M.quat_rotate_inverse(q1, v), g["quat_rotate_inverse"], "quat_rotate_inverse")
    assert_exact(M.quat_rotate_inverse(qu, v), g["quat_rotate_inverse_unnorm"], "quat_rotate_inverse(unnorm)")
    assert_exact(M.quat_apply(q1, v), g["quat_apply"], "quat_apply")
    assert_exact(M.quat_mul(q1, q2), g["quat_mul"], "quat_mul")
    assert_exact(M.quat_mul(qu, q2), g["quat_mul_unnorm"], "quat_mul(unnorm)")
    assert_exact(M.quat_conjugate(q1), g["quat_conjugate"], "quat_conjugate")
    assert_exact(M.quat_inv(qu), g["quat_inv"], "quat_inv")
    assert_exact(M.quat_unique(q1), g["quat_unique"], "quat_unique")
    assert_exact(M.normalize(v), g["normalize3"], "normalize3")
    assert_exact(M.normalize(qu), g["normalize4"], "normalize4")
    assert_exact(M.matrix_from_quat(qu), g["matrix_from_quat"], "matrix_from_quat")
    assert_exact(M.skew_symmetric_matrix(v), g["skew"], "skew")


def test_rotations_with_transcendentals():
    g = _g()
    q1, q2, v, v2 = g["q1"], g["q2"], g["v"], g["v2"]
    assert_close(M.yaw_quat(q1), g["yaw_quat"], what="yaw_quat")
    assert_close(M.quat_apply_yaw(q1, v), g["quat_apply_yaw"], what="quat_apply_yaw")
    assert_close(M.quat_box_minus(q1, q2), g["quat_box_minus"], what="quat_box_minus")
    assert_close(M.axis_angle_from_quat(q1), g["axis_angle_from_quat"], what="axis_angle_from_quat")
    assert_close(M.quat_error_magnitude(q1, q2), g["quat_error_magnitude"], what="quat_error_magnitude")
    assert_close(M.quat_from_angle_axis(g["angle"], v2), g["quat_from_angle_axis"], what="quat_from_angle_axis")
    rpy = g["rpy"]
    assert_close(M.quat_from_euler_xyz(rpy[:, 0], rpy[:, 1], rpy[:, 2]), g["quat_from_euler_xyz"], what="from_euler")
    assert_close(torch.stack(M.euler_xyz_from_quat(q1), -1), g["euler_xyz_from_quat"], what="euler_xyz_from_quat")


def test_frames():
    g = _g()
    q1, q2, qu, v, v2 = g["q1"], g["q2"], g["qu"], g["v"], g["v2"]
    p, q = M.combine_frame_transforms(v, q1, g["t12"], q2)
    assert_exact(p, g["combine_pos"], "combine pos")
    assert_exact(q, g["combine_quat"], "combine quat")
    p, q = M.subtract_frame_transforms(v, qu, v2, q2)
    assert_exact(p, g["subtract_pos"], "subtract pos")
    assert_exact(q, g["subtract_quat"], "subtract quat")
    p, q = M.subtract_frame_transforms(v, q1)
    assert_exact(p, g["subtract_pos_none"], "subtract pos (t02=None)")
    assert_exact(q, g["subtract_quat_none"], "subtract quat (q02=None)")
    pe, re = M.compute_pose_error(v, q1, v2, q2)
    assert_exact(pe, g["pose_err_pos"], "pose error pos")
    assert_close(re, g["pose_err_aa"], what="pose error axis-angle")
    _, rq = M.compute_pose_error(v, qu, v2, q2, "quat")
    assert_exact(rq, g["pose_err_quat"], "pose error quat")
    p, q = M.apply_delta_pose(v, q1, g["delta"])
    assert_exact(p, g["delta_pos"], "delta pos")
    assert_close(q, g["delta_quat"], what="delta quat")
    assert not torch.isnan(q).any(), "zero-rotation rows must be masked, not NaN"
    assert_exact(M.transform_points(g["tp_points"], v[:16], qu[:16]), g["transform_points"], "transform_points")


def test_general():
    g = _g()
    assert_exact(M.wrap_to_pi(g["wrap_in"]), g["wrap_to_pi"], "wrap_to_pi")
    x, lo, hi = g["st_x"], g["st_lo"], g["st_hi"]
    assert_exact(M.scale_transform(x, lo, hi), g["scale_transform"], "scale_transform")
    assert_exact(M.unscale_transform(x, lo, hi), g["unscale_transform"], "unscale_transform")
    assert_exact(M.saturate(x, lo, hi), g["saturate"], "saturate")


def test_wrap_to_pi_table():
    # the reference's MATLAB table (test_math.py:246-278): +-pi stay, odd multiples map to +-pi
    a = torch.tensor([0.0, math.pi, -math.pi, 3 * math.pi, -3 * math.pi, 4 * math.pi, -4 * math.pi, math.pi / 2,
                      -math.pi / 2, 1.5 * math.pi, 7 * math.pi / 4, -1.5 * math.pi])
    e = torch.tensor([0.0, math.pi, -math.pi, math.pi, -math.pi, 0.0, 0.0, math.pi / 2, -math.pi / 2, -math.pi / 2,
                      -math.pi / 4, math.pi / 2])
    assert_close(M.wrap_to_pi(a), e, rtol=1e-5, atol=1e-5, what="wrap_to_pi table")


def test_known_answer_tables():
    k = load_golden("math_kats")
    assert_close(M.axis_angle_from_quat(k["aa_q"]), k["aa_out"], what="axis_angle table")
    assert_close(M.axis_angle_from_quat(k["aa_small_q"]), k["aa_small_out"], what="axis_angle small-angle")
    assert_close(M.quat_error_magnitude(k["qem_a"], k["qem_b"]), k["qem_out"], atol=1e-5, what="quat_error_magnitude")
    # analytic answers for the table (test_math.py:66-85 style): rotation of pi/2 about x, pi about x
    aa = M.axis_angle_from_quat(torch.tensor([[0.7071068, 0.7071068, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0]]))
    assert_close(aa, torch.tensor([[math.pi / 2, 0, 0], [math.pi, 0, 0]]), atol=1e-5, what="axis_angle analytic")


def test_sign_invariance():
    # test_math.py:157-207: q and -q are the same rotation
    torch.manual_seed(3)
    q1 = M.normalize(torch.randn(64, 4))
    q2 = M.normalize(torch.randn(64, 4))
    assert_close(M.quat_error_magnitude(q1, q2), M.quat_error_magnitude(-q1, q2), atol=2e-5, what="error magnitude sign")
    assert_close(M.quat_unique(M.quat_mul(q1, q2)), M.quat_unique(M.quat_mul(-q1, q2)), what="quat_mul sign")
