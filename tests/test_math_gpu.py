"""GPU: isaaclab_amd.utils.math (HIP kernels through the C ABI) against the oracle and the golden fixtures."""

import math

import pytest
import torch

from conftest import assert_close, assert_exact, load_golden
from oracle import mathx as M

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hm():
    from isaaclab_amd.utils import math as hm

    return hm


@pytest.fixture(scope="module")
def g(dev):
    return load_golden("math_fns")


def d(t, dev):
    return t.to(dev)


def test_exact_ops_match_oracle_and_reference(hm, g, dev):
    q1, q2, qu, v, v2 = (g[k].to(dev) for k in ("q1", "q2", "qu", "v", "v2"))
    cases = [
        ("quat_rotate", hm.quat_rotate(q1, v)),
        ("quat_rotate_inverse", hm.quat_rotate_inverse(q1, v)),
        ("quat_rotate_inverse_unnorm", hm.quat_rotate_inverse(qu, v)),
        ("quat_apply", hm.quat_apply(q1, v)),
        ("quat_mul", hm.quat_mul(q1, q2)),
        ("quat_mul_unnorm", hm.quat_mul(qu, q2)),
        ("quat_conjugate", hm.quat_conjugate(q1)),
        ("quat_inv", hm.quat_inv(qu)),
        ("quat_unique", hm.quat_unique(q1)),
        ("normalize3", hm.normalize(v)),
        ("normalize4", hm.normalize(qu)),
        ("matrix_from_quat", hm.matrix_from_quat(qu)),
        ("skew", hm.skew_symmetric_matrix(v)),
    ]
    for name, got in cases:
        assert_exact(got, g[name], f"{name} vs reference fixture")
    p, q = hm.combine_frame_transforms(v, q1, g["t12"].to(dev), q2)
    assert_exact(p, g["combine_pos"], "combine pos")
    assert_exact(q, g["combine_quat"], "combine quat")
    p, q = hm.subtract_frame_transforms(v, qu, v2, q2)
    assert_exact(p, g["subtract_pos"], "subtract pos")
    assert_exact(q, g["subtract_quat"], "subtract quat")
    p, q = hm.subtract_frame_transforms(v, q1)
    assert_exact(p, g["subtract_pos_none"], "subtract pos none")
    assert_exact(q, g["subtract_quat_none"], "subtract quat none")
    pe, rq = hm.compute_pose_error(v, qu, v2, q2, "quat")
    assert_exact(pe, (g["v2"] - g["v"]), "pose err pos")
    assert_exact(rq, g["pose_err_quat"], "pose err quat")
    assert_exact(hm.wrap_to_pi(g["wrap_in"].to(dev)), g["wrap_to_pi"], "wrap_to_pi")
    x, lo, hi = (g[k].to(dev) for k in ("st_x", "st_lo", "st_hi"))
    assert_exact(hm.scale_transform(x, lo, hi), g["scale_transform"], "scale_transform")
    assert_exact(hm.unscale_transform(x, lo, hi), g["unscale_transform"], "unscale_transform")
    assert_exact(hm.saturate(x, lo, hi), g["saturate"], "saturate")


def test_transcendental_ops_within_tolerance(hm, g, dev):
    q1, q2, v, v2 = (g[k].to(dev) for k in ("q1", "q2", "v", "v2"))
    assert_close(hm.yaw_quat(q1), g["yaw_quat"], what="yaw_quat")
    assert_close(hm.quat_apply_yaw(q1, v), g["quat_apply_yaw"], what="quat_apply_yaw")
    assert_close(hm.quat_box_minus(q1, q2), g["quat_box_minus"], what="quat_box_minus")
    assert_close(hm.axis_angle_from_quat(q1), g["axis_angle_from_quat"], what="axis_angle_from_quat")
    assert_close(hm.quat_error_magnitude(q1, q2), g["quat_error_magnitude"], what="quat_error_magnitude")
    assert_close(hm.quat_from_angle_axis(g["angle"].to(dev), v2), g["quat_from_angle_axis"], what="from_angle_axis")
    rpy = g["rpy"].to(dev)
    assert_close(hm.quat_from_euler_xyz(rpy[:, 0].contiguous(), rpy[:, 1].contiguous(), rpy[:, 2].contiguous()),
                 g["quat_from_euler_xyz"], what="from_euler")
    # angles are returned modulo 2*pi: compare on the circle
    e = torch.stack(hm.euler_xyz_from_quat(q1), -1).cpu()
    diff = M.wrap_to_pi(e - g["euler_xyz_from_quat"])
    assert diff.abs().max() < 2e-5, f"euler_xyz_from_quat: {diff.abs().max()}"
    pe, re = hm.compute_pose_error(v, q1, v2, q2)
    assert_close(re, g["pose_err_aa"], what="pose err axis-angle")
    p, q = hm.apply_delta_pose(v, q1, g["delta"].to(dev))
    assert_exact(p, g["delta_pos"], "delta pos")
    assert_close(q, g["delta_quat"], what="delta quat")
    assert_close(hm.transform_points(g["tp_points"].to(dev), v[:16], g["qu"][:16].to(dev)), g["transform_points"],
                 what="transform_points")


def test_known_answers(hm, dev):
    k = load_golden("math_kats")
    assert_close(hm.axis_angle_from_quat(k["aa_q"].to(dev)), k["aa_out"], what="axis_angle table")
    assert_close(hm.axis_angle_from_quat(k["aa_small_q"].to(dev)), k["aa_small_out"], what="axis_angle small")
    assert_close(hm.quat_error_magnitude(k["qem_a"].to(dev), k["qem_b"].to(dev)), k["qem_out"], atol=1e-5, what="qem")
    a = torch.tensor([0.0, math.pi, -math.pi, 3 * math.pi, -3 * math.pi, 4 * math.pi, math.pi / 2, 1.5 * math.pi])
    e = torch.tensor([0.0, math.pi, -math.pi, math.pi, -math.pi, 0.0, math.pi / 2, -math.pi / 2])
    assert_close(hm.wrap_to_pi(a.to(dev)), e, atol=1e-5, what="wrap_to_pi table")


def test_large_batch_and_leading_dims(hm, dev):
    # (1024, 2, 5, .) as the reference's quat_rotate tests use (test_math.py:280-425), vs the oracle
    torch.manual_seed(11)
    q = M.normalize(torch.randn(1024, 2, 5, 4))
    v = torch.randn(1024, 2, 5, 3)
    assert_exact(hm.quat_rotate(q.to(dev), v.to(dev)), M.quat_rotate(q, v), "quat_rotate 4-D")
    assert_exact(hm.quat_rotate_inverse(q.to(dev), v.to(dev)), M.quat_rotate_inverse(q, v), "quat_rotate_inverse 4-D")
    # rotate then inverse-rotate is the identity (size-independent property), 1M rows
    n = 1 << 20
    q = M.normalize(torch.randn(n, 4)).to(dev)
    v = torch.randn(n, 3).to(dev)
    back = hm.quat_rotate_inverse(q, hm.quat_rotate(q, v))
    assert (back - v).abs().max().item() < 1e-5


def test_empty_input(hm, dev):
    out = hm.quat_mul(torch.zeros(0, 4, device=dev), torch.zeros(0, 4, device=dev))
    assert out.shape == (0, 4)
