"""CPU: the sampler restatements against outputs of the reference functions, fed with the draws the reference consumed."""

import torch

from conftest import assert_close, assert_exact, load_golden
from oracle import mathx as M


def test_samplers_match_reference():
    g = load_golden("sampling")
    assert_exact(M.sample_uniform(g["uniform_scalar_u"], -0.5, 2.0), g["uniform_scalar"], "uniform, scalar bounds")
    assert_exact(M.sample_uniform(g["uniform_cols_u"], g["uniform_cols_lo"], g["uniform_cols_hi"]), g["uniform_cols"], "uniform, per column")
    assert_exact(M.sample_uniform(g["uniform_1d_u"], 0.1, 0.2), g["uniform_1d"], "uniform, 1-D")
    assert_exact(M.sample_log_uniform(g["log_uniform_u"], 1e-3, 10.0), g["log_uniform"], "log-uniform")
    assert_exact(M.sample_triangle(g["triangle_u"], -2.0, 3.0), g["triangle"], "triangle")
    # the stored normal draws were recovered from torch.normal's output: one rounding each way
    assert_close(M.sample_gaussian(g["gaussian_z"], 0.5, 2.0), g["gaussian"], rtol=1e-6, atol=1e-6, what="gaussian")
    assert_close(M.sample_gaussian(g["gaussian_t_z"], g["gaussian_t_mean"], g["gaussian_t_std"]), g["gaussian_t"], rtol=1e-5, atol=1e-6,
                 what="gaussian, tensor parameters")
    assert_exact(M.random_orientation(g["random_orientation_z"]), g["random_orientation"], "random orientation")
    assert_close(M.random_yaw_orientation(g["random_yaw_u"]), g["random_yaw"], atol=1e-6, what="random yaw orientation")
