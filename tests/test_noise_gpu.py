"""GPU: noise functions and models (HIP ``ilx_noise_apply``) against the fixture recorded from the reference."""

import pytest
import torch

from conftest import assert_close, assert_exact, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.mark.parametrize("op", ["add", "scale", "abs"])
def test_noise_functions(dev, op):
    from isaaclab_amd.utils import noise as N

    g = load_golden("noise")
    x, ca, cb = g["x"].to(dev), g["cols_a"].to(dev), g["cols_b"].to(dev)
    assert_exact(N.constant_noise(x, N.ConstantNoiseCfg(bias=0.37, operation=op)), g[f"constant_{op}"], "constant")
    assert_exact(N.constant_noise(x, N.ConstantNoiseCfg(bias=cb, operation=op)), g[f"constant_cols_{op}"], "constant, per column")
    u = lambda tag: g[f"{tag}_{op}_draws"].to(dev)  # noqa: E731
    assert_exact(N.uniform_noise(x, N.UniformNoiseCfg(-0.15, 0.45, operation=op), u("uniform")), g[f"uniform_{op}"], "uniform")
    assert_exact(N.uniform_noise(x, N.UniformNoiseCfg(ca, cb, operation=op), u("uniform_cols")), g[f"uniform_cols_{op}"],
                 "uniform, per column")
    assert_exact(N.gaussian_noise(x, N.GaussianNoiseCfg(mean=-0.15, std=0.45, operation=op), u("gaussian")), g[f"gaussian_{op}"],
                 "gaussian")
    assert_exact(N.gaussian_noise(x, N.GaussianNoiseCfg(mean=ca, std=cb, operation=op), u("gaussian_cols")), g[f"gaussian_cols_{op}"],
                 "gaussian, per column")


def test_generated_draws_and_errors(dev):
    from isaaclab_amd.utils import noise as N

    x = torch.zeros(4096, 8, device=dev)
    N.set_noise_seed(7)
    y = N.uniform_noise(x, N.UniformNoiseCfg(-0.5, 1.5))
    assert float(y.min()) >= -0.5 and float(y.max()) < 1.5 and abs(float(y.mean()) - 0.5) < 0.02
    z = N.gaussian_noise(x, N.GaussianNoiseCfg(mean=1.0, std=2.0))
    assert abs(float(z.mean()) - 1.0) < 0.05 and abs(float(z.std()) - 2.0) < 0.05
    assert not torch.equal(z, N.gaussian_noise(x, N.GaussianNoiseCfg(mean=1.0, std=2.0)))   # a new call draws anew
    with pytest.raises(ValueError, match="Unknown operation in noise"):
        N.constant_noise(x, N.ConstantNoiseCfg(bias=1.0, operation="mul"))
    with pytest.raises(Exception, match="GPU tensors"):
        N.constant_noise(x.cpu(), N.ConstantNoiseCfg(bias=1.0))
    assert N.uniform_noise(torch.zeros(0, 8, device=dev), N.UniformNoiseCfg()).shape == (0, 8)
    y3 = N.constant_noise(torch.ones(5, 3, 2, device=dev), N.ConstantNoiseCfg(bias=torch.tensor([1.0, 2.0])))
    assert_close(y3[2, 1], torch.tensor([2.0, 3.0]), what="per-column bias on 3-D data")


def test_additive_bias_model(dev):
    from isaaclab_amd.utils import noise as N

    g = load_golden("noise")
    x = g["x"].to(dev)
    cfg = N.NoiseModelWithAdditiveBiasCfg(noise_cfg=N.GaussianNoiseCfg(mean=0.0, std=0.05), bias_noise_cfg=N.GaussianNoiseCfg(mean=0.0, std=0.3))
    model = cfg.class_type(cfg, num_envs=x.shape[0], device=dev)
    for k in range(3):
        ids = None if k == 0 else g[f"model_reset{k}_ids"].tolist()
        model.reset(ids, _draws=g[f"model_reset{k}_draws"])
        assert_exact(model._bias, g[f"model_bias{k}"], f"bias after reset {k}")
        assert_exact(model.apply(x, _draws=g[f"model_apply{k}_draws"]), g[f"model_apply{k}"], f"apply {k}")
    plain = N.NoiseModelCfg(noise_cfg=N.ConstantNoiseCfg(bias=0.37))
    assert_exact(plain.class_type(plain, x.shape[0], dev).apply(x), g["constant_add"], "NoiseModel.apply")
