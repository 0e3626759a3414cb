"""CPU: the noise oracle against the fixture recorded from the reference's functions and models."""

import pytest

from conftest import assert_exact, load_golden
from oracle import noise as O


@pytest.mark.parametrize("op", ["add", "scale", "abs"])
def test_noise_functions(op):
    g = load_golden("noise")
    x, ca, cb = g["x"], g["cols_a"], g["cols_b"]
    assert_exact(O.constant(x, 0.37, op), g[f"constant_{op}"], "constant")
    assert_exact(O.constant(x, cb, op), g[f"constant_cols_{op}"], "constant, per column")
    assert_exact(O.uniform(x, -0.15, 0.45, op, g[f"uniform_{op}_draws"]), g[f"uniform_{op}"], "uniform")
    assert_exact(O.uniform(x, ca, cb, op, g[f"uniform_cols_{op}_draws"]), g[f"uniform_cols_{op}"], "uniform, per column")
    assert_exact(O.gaussian(x, -0.15, 0.45, op, g[f"gaussian_{op}_draws"]), g[f"gaussian_{op}"], "gaussian")
    assert_exact(O.gaussian(x, ca, cb, op, g[f"gaussian_cols_{op}_draws"]), g[f"gaussian_cols_{op}"], "gaussian, per column")


def test_additive_bias_model():
    g = load_golden("noise")
    model = O.AdditiveBiasModel(g["x"].shape[0])
    for k in range(3):
        model.reset(g[f"model_reset{k}_ids"], 0.0, 0.3, g[f"model_reset{k}_draws"])
        assert_exact(model.bias, g[f"model_bias{k}"], f"bias after reset {k}")
        assert_exact(model.apply(g["x"], 0.0, 0.05, g[f"model_apply{k}_draws"]), g[f"model_apply{k}"], f"apply {k}")
