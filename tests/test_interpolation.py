"""LinearInterpolation: the oracle (CPU) and the HIP kernel (GPU) against the fixture recorded from the reference class."""

import pytest
import torch

from conftest import assert_close, load_golden
from oracle import actuators as A


def _same_nan(a, b, what):
    # a repeated sample makes the reference divide 0 by 0 for queries on it: NaN there, and here
    assert torch.equal(torch.isnan(a.cpu()), torch.isnan(b)), f"{what}: NaN pattern differs"
    assert_close(torch.nan_to_num(a.cpu()), torch.nan_to_num(b), atol=1e-6, what=what)


def test_oracle_linear_interpolation():
    g = load_golden("interpolation")
    _same_nan(A.linear_interpolation(g["x"], g["y"], g["q"]), g["f"], "oracle")


@pytest.mark.gpu
def test_linear_interpolation_gpu():
    from isaaclab_amd.utils.interpolation import LinearInterpolation

    dev = torch.device("cuda:0")
    g = load_golden("interpolation")
    f = LinearInterpolation(g["x"], g["y"], dev)
    _same_nan(f.compute(g["q"].to(dev)), g["f"], "kernel")
    _same_nan(LinearInterpolation(torch.tensor([0.3]), torch.tensor([2.0]), dev).compute(g["q"].to(dev)), g["one_sample"], "one sample")
    with pytest.raises(ValueError, match="not sorted"):
        LinearInterpolation(torch.tensor([1.0, 0.0]), torch.tensor([0.0, 1.0]), dev)
    with pytest.raises(ValueError, match="different sizes"):
        LinearInterpolation(torch.tensor([1.0, 2.0]), torch.tensor([0.0]), dev)
