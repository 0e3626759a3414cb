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


def test_samplers_replay_reference_draws(dev):
    """The device samplers fed with the draws the reference consumed (fixture from the reference functions)."""
    from isaaclab_amd.utils import math as M

    g = load_golden("sampling")
    d = str(dev)
    assert_exact(M.sample_uniform(-0.5, 2.0, (500, 3), d, _u=g["uniform_scalar_u"]), g["uniform_scalar"], "uniform, scalar bounds")
    assert_exact(M.sample_uniform(g["uniform_cols_lo"], g["uniform_cols_hi"], (500, 6), d, _u=g["uniform_cols_u"]), g["uniform_cols"],
                 "uniform, per-column bounds")
    assert_exact(M.sample_uniform(0.1, 0.2, 500, d, _u=g["uniform_1d_u"]), g["uniform_1d"], "uniform, 1-D")
    assert_close(M.sample_log_uniform(1e-3, 10.0, (500, 2), d, _u=g["log_uniform_u"]), g["log_uniform"], rtol=2e-6, atol=0, what="log-uniform")
    # five roundings in a row (two of them after a square root): a last-place difference on a few elements
    assert_close(M.sample_triangle(-2.0, 3.0, (500, 2), d, _u=g["triangle_u"]), g["triangle"], rtol=3e-7, atol=3e-7, what="triangle")
    assert_close(M.sample_gaussian(0.5, 2.0, (500, 3), d, _z=g["gaussian_z"]), g["gaussian"], rtol=1e-6, atol=1e-6, what="gaussian")
    assert_close(M.sample_gaussian(g["gaussian_t_mean"], g["gaussian_t_std"], (500, 3), d, _z=g["gaussian_t_z"]), g["gaussian_t"],
                 rtol=1e-5, atol=1e-6, what="gaussian, tensor parameters")
    assert_close(M.random_orientation(500, d, _z=g["random_orientation_z"]), g["random_orientation"], atol=1e-6, what="random orientation")
    assert_close(M.random_yaw_orientation(500, d, _u=g["random_yaw_u"]), g["random_yaw"], atol=1e-6, what="random yaw orientation")
    assert_exact(M.default_orientation(7, d), g["default_orientation"], "default orientation")


def test_samplers_generate_the_right_distributions(dev):
    """Generated (Philox) draws: ranges, moments, reproducibility per seed, fresh values per call."""
    from isaaclab_amd.utils import math as M

    d, n = str(dev), 200000
    M.set_sampling_seed(5)
    a = M.sample_uniform(-1.0, 3.0, (n, 2), d)
    b = M.sample_uniform(-1.0, 3.0, (n, 2), d)
    M.set_sampling_seed(5)
    a2 = M.sample_uniform(-1.0, 3.0, (n, 2), d)
    assert torch.equal(a, a2) and not torch.equal(a, b)
    assert float(a.min()) >= -1.0 and float(a.max()) < 3.0 and abs(float(a.mean()) - 1.0) < 0.01
    assert abs(float(a.std()) - 4.0 / 12 ** 0.5) < 0.01 and abs(float(torch.corrcoef(a.T)[0, 1])) < 0.01
    lg = M.sample_log_uniform(1e-2, 1e2, n, d)
    assert float(lg.min()) >= 1e-2 * (1 - 1e-6) and float(lg.max()) <= 1e2 * (1 + 1e-6) and abs(float(lg.log10().mean())) < 0.02
    tri = M.sample_triangle(0.0, 2.0, n, d)
    assert abs(float(tri.mean()) - 1.0) < 0.01 and float(tri.min()) >= 0.0 and float(tri.max()) <= 2.0
    z = M.sample_gaussian(1.5, 0.5, (n, 3), d)
    assert abs(float(z.mean()) - 1.5) < 0.01 and abs(float(z.std()) - 0.5) < 0.01
    k = ((z - 1.5) / 0.5) ** 4
    assert abs(float(k.mean()) - 3.0) < 0.1, "normal kurtosis"
    q = M.random_orientation(n, d)
    assert_close(q.norm(dim=-1), torch.ones(n), atol=1e-6, what="unit quaternions")
    assert float(q.mean(0).abs().max()) < 0.01 and abs(float((q[:, 0] ** 2).mean()) - 0.25) < 0.01
    y = M.random_yaw_orientation(n, d)
    assert float(y[:, 1:3].abs().max()) == 0.0
    yaw = 2 * torch.atan2(y[:, 3], y[:, 0])
    assert abs(float(yaw.mean()) - torch.pi) < 0.02 and float(yaw.min()) >= 0.0
